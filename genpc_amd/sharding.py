"""Per-scan sharding over the GPUs of one node.

Every scan (batch element) of the hot path is independent, so the data path has
NO collective: rank r owns scans r, r+W, r+2W, ...  The only exchange is the
epilogue gather of per-scan scalars (CD-L1, CD-L2, EMD: a few hundred bytes), done
with one all_gather over RCCL (backend "nccl" on ROCm) -- or gloo on CPU, which
is how tests/test_sharding.py covers this module without GPUs.  The reference's
only multi-device mechanism, DataParallel around EMD (utils/loss_util.py:12), is a
pass-through at its batch-of-one call sites and is not reproduced.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns
    (rank, local_rank, world_size).  Rendezvous always on 127.0.0.1 unless the
    launcher says otherwise."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_indices(n_items, rank, world):
    """Round-robin ownership: scan s belongs to rank s % world."""
    return list(range(rank, n_items, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_scan_metrics(local, n_items, rank, world):
    """local: [n_local, K] tensor of per-scan scalars for the scans of
    shard_indices(n_items, rank, world), in that order.  Returns the [n_items, K]
    table in scan order on every rank.  Ranks may own different numbers of scans
    (13 scans over 4 ranks), so rows are padded to the maximum before the gather."""
    k = local.shape[1] if local.dim() == 2 else 1
    local = local.reshape(-1, k)
    n_mine = len(shard_indices(n_items, rank, world))
    if local.shape[0] != n_mine:
        raise ValueError("gather_scan_metrics: rank %d of %d owns %d of %d scans but was handed %d rows"
                         % (rank, world, n_mine, n_items, local.shape[0]))
    if world == 1:
        return local.clone()
    if not dist.is_initialized():
        # a [n_local, K] table passed off as the full one would misalign every scan id
        raise RuntimeError("gather_scan_metrics: WORLD_SIZE is %d but torch.distributed is not initialised "
                           "(call genpc_amd.sharding.init() first)" % world)
    per = (n_items + world - 1) // world
    pad = torch.full((per, k), float("nan"), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    table = torch.empty((n_items, k), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        if idx:
            table[torch.tensor(idx, device=local.device)] = out[r][: len(idx)]
    assert table.shape[0] == n_items
    return table


def all_ranks(device="cpu"):
    """The ranks that answer an all_gather, in order (bench.py's `ranks_seen`)."""
    if not dist.is_initialized():
        return [0]
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t.item()) for t in out]
