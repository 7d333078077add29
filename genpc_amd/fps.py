"""Farthest point sampling on the gfx950 library: the deterministic counterpart of
``fpsample.fps_sampling(points, k)`` as the reference calls it (main.py:21-24,
reg_xyz.py:215, DepthPrompting.py:88).  Start index 0 (fpsample starts at a random
index, so the reference's own subsamples are not reproducible)."""
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr


def fps_sampling(points, k):
    """points: [N,3] or [C,N,3] GPU tensor -> int32 indices [k] or [C,k].  The first
    index is always 0; the kernel writes -1 there if its inter-workgroup hand-off timed
    out (another kernel kept the cloud's workgroups from being co-resident)."""
    single = points.dim() == 2
    pts = (points[None] if single else points).contiguous().float()
    _lib.check_tensors((("points", pts),))
    c, n, _ = pts.shape
    out = torch.empty(c, k, device=pts.device, dtype=torch.int32)
    rc = _lib.on_device_of(pts, _L.genpc_fps, c, n, _p(pts), int(k), _p(out))
    if rc == -1:
        raise ValueError("fps_sampling: need 0 < k <= N <= 262144 (k=%d, N=%d)" % (k, n))
    if rc != 1:
        raise RuntimeError("genpc_fps failed: " + _lib.last_error())
    # index 0 is the start point of every cloud; the kernel writes -1 there when a hand-off
    # between its workgroups timed out (the samples after it would be garbage)
    if bool((out[:, 0] != 0).any()):
        raise RuntimeError("genpc_fps: inter-workgroup hand-off timed out (workgroups of a cloud were not "
                           "co-resident); no samples returned")
    return out[0] if single else out


def fps_subsample(points, k):
    """The helper metric.py calls but never defines: [B,N,3] -> [B,k,3]."""
    idx = fps_sampling(points, k).long()
    if points.dim() == 2:
        return points[idx]
    return torch.gather(points, 1, idx[..., None].expand(-1, -1, 3))
