"""Evaluation harness with the arithmetic of the reference's ``main.metric``
(main.py:11-36): CD-L1 and EMD(eps 0.005, 50 rounds) between a completed cloud and
its ground truth, printed x100.  The reference does this one scan at a time after
reading PLY files with open3d and subsampling with fpsample (random start, not
reproducible); here the clouds arrive as tensors (fixtures are deterministic-FPS
subsamples, tests/golden/make_golden.py), scans are batched per call and sharded
over ranks (genpc_amd.sharding) with one all_gather of the per-scan scalars.

    python -m genpc_amd.metric [--npz tests/golden/scans13_fps16384.npz]
    torchrun --nproc-per-node 4 -m genpc_amd.metric --npz ...
"""
import argparse
import os

import numpy as np
import torch

from . import sharding
from .loss_functions import chamfer_3DDist, emdModule


def evaluate_scans(pred, gt, eps=0.005, iters=50):
    """pred, gt: [S,N,3] GPU tensors (N % 256 == 0).  Returns [S,3] = CD-L1, CD-L2,
    EMD per scan, with the reductions of utils/loss_util.py:25-49 applied per scan."""
    d1, d2, _, _ = chamfer_3DDist()(pred, gt)
    cd_l1 = (torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1)) / 2
    cd_l2 = d1.mean(1) + d2.mean(1)
    de, _ = emdModule()(pred, gt, eps, iters)
    emd = torch.sqrt(de).mean(1)
    return torch.stack([cd_l1, cd_l2, emd], dim=1)


def cd_l1_cpu_plumbing(pred, gt, chunk=512):
    """BASELINE config 1: CD-L1 of CPU tensors in plain torch -- the metric's plumbing without a GPU.  NOT a
    fallback: nothing in the library routes here (CPU tensors handed to chamfer_3DDist / emdModule raise); a caller that
    wants the CPU number asks for it by name (``python -m genpc_amd.metric --cpu``).  Direct form in the reference's
    operation order, fp32: d = ((x2-x1)^2 + (y2-y1)^2) + (z2-z1)^2, min over the other cloud, then
    (mean sqrt d1 + mean sqrt d2) / 2 (utils/loss_util.py:25-29).  pred, gt: [S,N,3] / [S,M,3] float32 CPU."""
    if pred.is_cuda or gt.is_cuda:
        raise ValueError("cd_l1_cpu_plumbing is the CPU plumbing path; GPU tensors go through evaluate_scans")
    pred, gt = pred.float(), gt.float()

    def one_way(a, b):                      # [N,3] x [M,3] -> [N] squared NN distances
        out = []
        for i in range(0, a.shape[0], chunk):
            d = b[None, :, :] - a[i:i + chunk, None, :]
            out.append(((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).min(1).values)
        return torch.cat(out)
    rows = [(torch.sqrt(one_way(p, g)).mean() + torch.sqrt(one_way(g, p)).mean()) / 2 for p, g in zip(pred, gt)]
    return torch.stack(rows)


def evaluate_sharded(pred_np, gt_np, device=None, max_batch=16, metric_fn=None, backend=None):
    """Round-robin shard of the S scans over the ranks of the default process group
    (initialised here if the launcher's WORLD_SIZE > 1 and nobody did yet); every rank
    returns the full [S,K] table (scan order).  metric_fn(pred[S',N,3], gt[S',N,3]) ->
    [S',K] defaults to evaluate_scans (K = 3: CD-L1, CD-L2, EMD)."""
    rank, local_rank, world = sharding.init(backend)
    if metric_fn is None:
        metric_fn = evaluate_scans
    if device is None:
        device = torch.device("cuda", local_rank)
    s_total = pred_np.shape[0]
    mine = sharding.shard_indices(s_total, rank, world)
    rows = []
    for i in range(0, len(mine), max_batch):
        sel = mine[i:i + max_batch]
        p = torch.from_numpy(np.ascontiguousarray(pred_np[sel])).to(device)
        g = torch.from_numpy(np.ascontiguousarray(gt_np[sel])).to(device)
        rows.append(metric_fn(p, g))
    # K (columns per scan) is agreed on collectively: a rank that owns no scan (more ranks than scans) cannot know
    # what a custom metric_fn returns, and ranks entering the all_gather with different shapes would hang
    k_local = rows[0].shape[1] if rows else 0
    k = int(sharding.max_over_ranks(float(k_local), device=device if (world > 1 and torch.device(device).type == "cuda") else "cpu"))
    if k <= 0:
        raise ValueError("evaluate_sharded: no rank owns a scan")
    if rows and any(r.shape[1] != k for r in rows):
        raise ValueError("evaluate_sharded: metric_fn returned %d columns here, %d elsewhere" % (k_local, k))
    local = torch.cat(rows) if rows else torch.empty(0, k, device=device)
    return sharding.gather_scan_metrics(local, s_total, rank, world)


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--npz", default=os.path.join(here, "..", "tests", "golden", "scans13_fps16384.npz"))
    ap.add_argument("--cpu", action="store_true", help="BASELINE config 1: CD-L1 only, plain torch on the CPU (plumbing, no GPU)")
    args = ap.parse_args()
    if args.cpu:
        z = np.load(args.npz)
        cd = cd_l1_cpu_plumbing(torch.from_numpy(z["partial"]), torch.from_numpy(z["gt"]))
        ids = z["ids"] if "ids" in z.files else [str(i) for i in range(len(cd))]
        for flag, v in zip(ids, cd):
            print(f"Flag: {flag}, CD: {float(v) * 100:.3f}")
        return
    rank, local_rank, world = sharding.init()
    torch.cuda.set_device(local_rank)
    z = np.load(args.npz)
    table = evaluate_sharded(z["partial"], z["gt"]).cpu().numpy()
    if rank == 0:
        for flag, (cd, _, emd) in zip(z["ids"], table):
            print(f"Flag: {flag}, CD: {cd * 100:.3f}, EMD: {emd * 100:.3f}")      # main.py:35
        ok = [i for i, f in enumerate(z["ids"]) if f != "06830"]                  # GT mis-framed
        print(f"mean over {len(ok)} well-framed scans: CD {table[ok, 0].mean() * 100:.6f} "
              f"EMD {table[ok, 2].mean() * 100:.6f}")
    sharding.shutdown()


if __name__ == "__main__":
    main()
