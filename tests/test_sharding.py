"""world_size-2 gloo coverage of the N>1 path (scan sharding + scalar gather +
the bench's max-over-ranks timing reduction).  Runs on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from genpc_amd import sharding
    r, lr, w = sharding.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = sharding.shard_indices(n_items, r, w)
    # per-scan "metrics": (scan id, 10*scan id, rank) -- checks order and ownership
    local = torch.tensor([[s, 10.0 * s, float(r)] for s in mine], dtype=torch.float32).reshape(-1, 3)
    table = sharding.gather_scan_metrics(local, n_items, r, w)
    t = sharding.max_over_ranks(1.0 + rank)
    s = sharding.sum_over_ranks(len(mine))
    sharding.barrier()
    q.put((rank, table.tolist(), t, s))
    sharding.shutdown()


@pytest.mark.parametrize("n_items", [13, 4, 1])
def test_two_rank_gather(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, table, t, s in res:
        assert t == 2.0 and s == n_items
        assert len(table) == n_items
        for sidx, row in enumerate(table):
            assert row == [float(sidx), 10.0 * sidx, float(sidx % world)]


def test_shard_indices_partition():
    from genpc_amd.sharding import shard_indices
    for n, w in [(64, 8), (13, 4), (59, 4), (3, 8)]:
        owned = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert owned == list(range(n))
        sizes = [len(shard_indices(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
