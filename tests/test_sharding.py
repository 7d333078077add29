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


def _stub_metric(p, g):
    """Stand-in for evaluate_scans on CPU tensors: per scan (first x of pred, sum of gt, #points)."""
    return torch.stack([p[:, 0, 0], g.sum(dim=(1, 2)), torch.full((p.shape[0],), float(p.shape[1]))], dim=1)


def _worker_eval(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from genpc_amd import sharding
    from genpc_amd.metric import evaluate_sharded
    rng = np.random.default_rng(5)
    pred = rng.random((13, 32, 3), dtype=np.float32)
    gt = rng.random((13, 32, 3), dtype=np.float32)
    # evaluate_sharded initialises the process group itself (gloo here, nccl = RCCL on GPUs)
    table = evaluate_sharded(pred, gt, device=torch.device("cpu"), max_batch=3, metric_fn=_stub_metric, backend="gloo")
    seen = sharding.all_ranks()
    # more ranks than scans, custom metric with 5 columns: the rank that owns nothing learns K from the others
    # (it used to guess 3 from an environment variable and hang the all_gather)
    t1 = evaluate_sharded(pred[:1], gt[:1], device=torch.device("cpu"), metric_fn=lambda p, g: torch.cat(
        [_stub_metric(p, g), _stub_metric(p, g)[:, :2] * 2], dim=1), backend="gloo")
    assert tuple(t1.shape) == (1, 5) and float(t1[0, 3]) == 2 * float(t1[0, 0])
    # a caller that skips init() must be told, not handed a misaligned table
    sharding.shutdown()
    err = None
    try:
        sharding.gather_scan_metrics(torch.zeros(len(sharding.shard_indices(13, rank, world)), 3), 13, rank, world)
    except RuntimeError as e:
        err = str(e)
    q.put((rank, table.numpy().tolist(), seen, err))


def test_evaluate_sharded_uneven_two_ranks():
    """13 scans over 2 ranks (7 + 6), batches of 3, stub metric on CPU tensors: the gathered
    table equals the single-process one, in scan order, on both ranks."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_eval, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    pred = rng.random((13, 32, 3), dtype=np.float32)
    gt = rng.random((13, 32, 3), dtype=np.float32)
    exp = _stub_metric(torch.from_numpy(pred), torch.from_numpy(gt)).numpy()
    for rank, table, seen, err in res:
        np.testing.assert_array_equal(np.asarray(table, np.float32), exp)
        assert seen == [0, 1]
        assert err is not None and "not initialised" in err


def test_gather_rejects_wrong_row_count():
    from genpc_amd.sharding import gather_scan_metrics
    with pytest.raises(ValueError):
        gather_scan_metrics(torch.zeros(5, 3), 13, 0, 1)


def test_bench_spawns_the_torchrun_form(monkeypatch):
    """`python bench.py --gpus 4` outside torchrun must start `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 ... bench.py --gpus 4 ...` as a child
    and exit with its code -- before touching a GPU; inside torchrun (WORLD_SIZE set) or with
    --gpus 1 it must not."""
    import subprocess
    import types
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.maybe_spawn(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd and "4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.maybe_spawn(types.SimpleNamespace(gpus=4))          # returns: already under a launcher
    monkeypatch.delenv("WORLD_SIZE")
    bench.maybe_spawn(types.SimpleNamespace(gpus=1))
    assert len(calls) == 1


def test_bench_scan_workload_two_ranks_gloo():
    """`bench.py --gpus 2 --workload c4` end to end on two CPU ranks over gloo: the self-spawned
    torchrun form, 59 scans dealt unevenly over 2 ranks (30 + 29), barriers, MAX-over-ranks timing,
    gather_scan_metrics and ranks_seen all execute; registration + metric are replaced by a stub of the same shapes
    (GENPC_BENCH_STUB=1 -- no kernel runs on CPU, nothing is measured).  The gathered table must equal the
    one-process run's."""
    import json
    import subprocess
    env = dict(os.environ)
    env["GENPC_BENCH_STUB"] = "1"
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    bench = os.path.join(ROOT, "bench.py")

    def run(gpus, workload="c4"):
        p = subprocess.run([sys.executable, bench, "--gpus", str(gpus), "--workload", workload, "--steps", "2", "--warmup", "1",
                            "--no-extra"], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        return json.loads(lines[0])

    two, one = run(2), run(1)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1] and two["scaling"] == "strong"
    assert two["config"]["scans"] == 59 and two["steps"] == 2 and two["value"] > 0 and two["extra"]["stub"] is True
    assert one["n_gpus"] == 1 and one["ranks_seen"] == [0]
    assert two["extra"]["scan_table_checksum"] == one["extra"]["scan_table_checksum"]
    assert two["extra"]["mean_cd_l1"] == one["extra"]["mean_cd_l1"]
    # every rank's own scan count and elapsed time travel with the line (load imbalance: 30 + 29)
    assert [r["scans"] for r in two["per_rank"]] == [30, 29] and all(r["elapsed_s"] >= 0 for r in two["per_rank"])
    assert [r["scans"] for r in one["per_rank"]] == [59]
    # config 3 (13 bundled scans, metric only) through the same path: 7 + 6
    three = run(2, "c3")
    assert three["config"]["scans"] == 13 and [r["scans"] for r in three["per_rank"]] == [7, 6]
    assert three["extra"]["scan_table_checksum"] == run(1, "c3")["extra"]["scan_table_checksum"]
