"""The HBM-bound rows of SURVEY.md section 8(a) at a size where the launch is not what is measured (VERDICT r4 item 5:
`chamfer_grad_kernel` appeared in no profile; CalcDist, the EMD gradient, the splat and the colour gather only at 4-9 us).

    rooflines(device, stream, reps) -> {row: {"bound": "hbm", "achieved": GB/s, "peak": 8000, "frac": ..., ...}}

Every entry: ALGORITHMIC bytes (the model is SURVEY 8d's, stated per entry) / average duration over `reps` launches
(HIP events on the stream the kernels run on) / 8 TB/s.  64 clouds x 32768 points (2.1 M points) unless stated.
Used by bench.py (`extra.streaming_rooflines`) and by tools/collect_profiles.sh (family `streaming_64x32768`), which
commits the rocprofv3 kernel trace + PMC passes of the same calls under profiles/."""
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr
PEAK_HBM_GBS = 8000.0
B, N = 64, 32768


def _time(fn, reps, stream):
    fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _entry(row, kernel, alg_bytes, ms, model, points):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"row": row, "kernel": kernel, "bound": "hbm", "unit": "GB/s", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
            "frac": round(gbs / PEAK_HBM_GBS, 4), "ms_per_launch": round(ms, 4), "algorithmic_bytes": int(alg_bytes),
            "points": int(points), "bytes_model": model}


def rooflines(dev, stream=None, reps=20, only=None):
    stream = stream or torch.cuda.current_stream(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(20250101)
    out = {}
    P = B * N
    X1 = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    X2 = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    idx = torch.randint(0, N, (B, N), device=dev, generator=g, dtype=torch.int32)
    idx2 = torch.randint(0, N, (B, N), device=dev, generator=g, dtype=torch.int32)
    gd1 = torch.rand(B, N, device=dev, generator=g)
    gd2 = torch.rand(B, N, device=dev, generator=g)

    def want(k):
        return only is None or k in only

    if want("a3"):
        # a3 NmDistanceGradKernel x2 (chamfer3D.cu:155-195): per query point of either direction: own xyz 12 + target
        # xyz (gather) 12 + graddist 4 + idx 4 read, 12 RMW on its own gradient row + 12 RMW scattered = 56 B / point
        g1 = torch.zeros(B, N, 3, device=dev)
        g2 = torch.zeros(B, N, 3, device=dev)
        f = lambda: _lib.on_device_of(X1, _L.genpc_chamfer_backward, B, N, _p(X1), N, _p(X2), _p(gd1), _p(idx), _p(gd2), _p(idx2), _p(g1), _p(g2))  # noqa: E731
        ms = _time(f, reps, stream)
        out["a3_chamfer_backward"] = _entry("a3", "chamfer_grad_kernel", 56.0 * 2 * P, ms,
                                            "56 B per point of either cloud (SURVEY 8d): xyz 12 + gathered target 12 + graddist 4 + idx 4 + two 12-B RMW", 2 * P)
    if want("a8"):
        # a8 CalcDist (emd_cuda.cu:217-226): xyz1 12 + assignment 4 + gathered xyz2 12 read, dist 4 written (+4 of slack in
        # SURVEY's 36: the index of the batch) -> 36 B / point
        dist = torch.empty(B, N, device=dev)
        f = lambda: _lib.on_device_of(X1, _L.genpc_emd_calc_dist, B, N, _p(X1), _p(X2), _p(idx), _p(dist))  # noqa: E731
        ms = _time(f, reps, stream)
        out["a8_emd_calc_dist"] = _entry("a8", "emd_calc_dist_kernel", 36.0 * P, ms, "36 B per point (SURVEY 8d): xyz1 12 + assignment 4 + gathered xyz2 12 + dist 4 (+4)", P)
    if want("a10"):
        # a10 EMD NmDistanceGradKernel (emd_cuda.cu:284-300): xyz1 12 + idx 4 + graddist 4 + gathered xyz2 12 read, gradxyz 12
        # read + 12 written (+=) = 56 B / point
        gx = torch.zeros(B, N, 3, device=dev)
        f = lambda: _lib.on_device_of(X1, _L.genpc_emd_backward, B, N, _p(X1), _p(X2), _p(gx), _p(gd1), _p(idx))  # noqa: E731
        ms = _time(f, reps, stream)
        out["a10_emd_backward"] = _entry("a10", "emd_grad_kernel", 56.0 * P, ms, "56 B per point: xyz1 12 + idx 4 + graddist 4 + gathered xyz2 12 + gradxyz 12 RMW (24)", P)
    if want("pose"):
        # ObjectPoseOptim.forward's point map (diff_obj_pose.py:419-423): 12 read + 12 written per point
        V = X1.reshape(-1, 3).contiguous()
        pts = torch.empty_like(V)
        center = V.mean(0).contiguous()
        params = torch.tensor([1, 0, 0, 0, 1, 0, 0.01, -0.02, 0.03, -0.1], device=dev, dtype=torch.float32)
        f = lambda: _lib.on_device_of(V, _L.genpc_pose_transform, V.shape[0], _p(V), _p(center), _p(params), _p(pts))  # noqa: E731
        ms = _time(f, reps, stream)
        out["a16_pose_transform"] = _entry("a16 (point map)", "pose_transform_kernel", 24.0 * P, ms, "24 B per point: 12 read + 12 written", P)
    if want("a15"):
        # a15 colorPoint's gather (ScaleAdapter.py:57-66): pix 8 read + 3 gathered colour channels 12 + colours written 12 = 32 B
        # per point; image 1024 x 1024 x 3 (the reference's img_resource size)
        H = W = 1024
        img = torch.rand(3, H, W, device=dev, generator=g)
        pix = torch.stack([torch.randint(0, H, (P,), device=dev, generator=g, dtype=torch.int32),
                           torch.randint(0, W, (P,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
        col = torch.empty(P, 3, device=dev)
        f = lambda: _lib.on_device_of(img, _L.genpc_gather_colors, P, _p(pix), _p(img), 3, H, W, _p(col))  # noqa: E731
        ms = _time(f, reps, stream)
        out["a15_gather_colors"] = _entry("a15", "gather_colors_kernel", 32.0 * P, ms, "32 B per point (SURVEY 8d): pix 8 + gathered colour 12 + written colour 12", P)
    if want("a14"):
        # a14 paintPixels (DepthPrompting.py:292-339), point_size 1, res 1024 (lidar configs use 2-3: more stamps per point):
        # pass 1 (owner election): pix 8 read + one 4-B atomic per point; pass 2: one thread per pixel, owner 4 read, the winner's
        # colour 12 gathered for covered pixels, img RMW + flipped out written 2 x 12 per pixel.
        # algorithmic bytes: 12 B per point (pix + election) + 12 B per point whose colour lands (<= covered pixels) + 28 B / pixel
        res = 1024
        pix = torch.stack([torch.randint(0, res, (P,), device=dev, generator=g, dtype=torch.int32),
                           torch.randint(0, res, (P,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
        col = torch.rand(P, 3, device=dev, generator=g)
        img = torch.zeros(3, res, res, device=dev)
        outi = torch.empty(3, res, res, device=dev)
        owner = torch.empty(res * res, device=dev, dtype=torch.int32)
        f = lambda: _lib.on_device_of(img, _L.genpc_paint_pixels, res, P, _p(pix), _p(col), 3, 1, _p(img), _p(outi), _p(owner))  # noqa: E731
        ms = _time(f, reps, stream)
        covered = min(P, res * res)
        alg = 12.0 * P + 12.0 * covered + 28.0 * res * res
        out["a14_paint_pixels"] = _entry("a14", "splat_owner_kernel + splat_write_kernel", alg, ms,
                                         "12 B per point (pix 8 + one 4-B election) + 12 B per covered pixel (winner's colour) + 28 B per pixel (owner 4, img 12, flipped out 12)", P)
    return out


if __name__ == "__main__":
    import json
    import sys
    dev = torch.device("cuda:0")
    only = set(sys.argv[1:]) or None
    print(json.dumps(rooflines(dev, only=only), indent=1))
