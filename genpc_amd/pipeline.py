"""The geometric stages of one completed scan, chained on the GPU (BASELINE config 2 without
the two generators): what ``main.py`` does around the diffusion model and
the image-to-3D model of the reference, on tensors instead of files.

    stage 1  DepthPrompting (DepthPrompting.py:100-237): viewpoint selection, projection of the
             partial cloud into the chosen camera, sparse colour / depth images and hole masks
             -- the inputs of the depth-conditioned diffusion model (a stock torch module, out of
             scope: its output image is an argument here)
    stage 2  ScaleAdapter (ScaleAdapter.py:15-86): colours of the partial points from the
             generated image (colorPoint), then reg(): pose initialisation, coarse scale sweep,
             anisotropic scale search (reg_xyz.py:99-205) against the generated shape (the
             image-to-3D model's output, an argument here), then the fusion tail (:207-219)
    metric   main.metric (main.py:11-36): FPS to 16384 points, CD-L1 / CD-L2 / EMD against the
             ground truth

``complete_scan`` returns every intermediate product so that tests can check each stage
against the oracle on the stage's actual input.

Stage 1 and the rest read the same inputs and nothing of each other (the registration works on coordinates; the
colours only travel into the fused file), so ``complete_scan`` runs stage 1 on a second HIP stream, driven by a second
host thread, under the scan's TAIL (fusion + metric: two farthest-point samplings that occupy one compute unit each and
leave the chip to the viewpoint selection's chip-wide kernels; rounds 3-5 ran it under the alignment loop, whose small
dependent launches it delayed).  The library's scratch is keyed by stream, its entry points take the stream as an
argument, ctypes releases the interpreter lock during a call: the two calls share nothing but the inputs.
``overlap=False`` runs the stages one after the other (same results).
"""
import threading
from types import SimpleNamespace

import torch

from .DepthPrompting import DepthPrompting
from .ScaleAdapter import ScaleAdapter
from .fps import fps_sampling, FpsCombiner
from .metric import evaluate_scans
from . import reg_xyz
from . import _lib


def default_cfg(device="cuda", view_num=1024):
    """configs/config.yaml values of the reference that the geometric stages read."""
    return SimpleNamespace(device=str(device), fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                           mask_pixel_rate=3, view_num=view_num, distance=1.6, downsample_num=10000, removal_radius=10000,
                           generative_model="trellis", dataset="redwood")


def fps_to(xyz, k):
    """main.py:21-24: farthest point sampling to k points (pad-repeat when the cloud is smaller)."""
    n = xyz.shape[0]
    if n >= k:
        return xyz[fps_sampling(xyz.contiguous().float(), k).long()]
    rep = xyz[torch.arange(k - n, device=xyz.device) % n]
    return torch.cat([xyz, rep], dim=0)


_SIDE = {}
_LANE = {}
_NO_OVERLAP = __import__("os").environ.get("GENPC_C2_NO_OVERLAP", "0") == "1"      # A/B switch
_FPS_DEFER = __import__("os").environ.get("GENPC_FPS_DEFER", "1") != "0"              # A/B switch: the samplings' check beside the tail (complete_scan)
_NO_COMBINER = __import__("os").environ.get("GENPC_FPS_COMBINER", "0") != "1"        # A/B switch (off: see run_in_lanes)


_SIDE_LOCK = threading.Lock()
_TLS = threading.local()        # .lanes: how many scans the calling lane thread's run_in_lanes keeps in flight


def _fresh_stream(dev, avoid=(), priority=0):
    """A stream that is none of ours and none of `avoid` (torch deals streams from a pool of 32 per device, round-robin: a
    new Stream() can BE an old one; two host threads driving one stream would share the library's per-stream scratch)."""
    taken = {s.cuda_stream for s in list(_SIDE.values()) + list(_LANE.values())} | {int(a) for a in avoid}
    for _ in range(64):
        st = torch.cuda.Stream(device=dev, priority=priority)
        if st.cuda_stream not in taken:
            return st
    raise RuntimeError("genpc_amd.pipeline: no unused stream left in torch's pool")


def _side_stream(device, main):
    """Stage 1's stream for the scan whose other stages run on `main` (one per main stream: several scans in flight,
    complete_scans below, must not share it)."""
    # (measured next to it: a high-priority stream for the loop -- no difference, 19.5 scans/s either way; stage 1's stream
    #  confined to three quarters of the CUs with hipExtStreamCreateWithCUMask -- 15.4)
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, main.cuda_stream)
    with _SIDE_LOCK:
        if key not in _SIDE:
            # A stream of ANOTHER priority class: the runtime deals a class's streams onto a handful of hardware queues (four by
            # default), and two streams that land on one queue do not overlap at all -- in a process that has made many streams
            # (bench.py by the time it reaches this line) stage 1 then simply ran behind the tail: 20.4 scans/s where a fresh
            # process measured 24.6.  The high-priority class has queues of its own.  (GENPC_C2_SIDE_PRIORITY=0: the old choice.)
            # Several scans in flight (run_in_lanes) keep the caller's class: six high-priority stage-1 streams crowd the few queues
            # of that class (six lanes, three runs each: 46.8 scans/s with the caller's class, 44.8 with the high one).
            prio = -1 if __import__("os").environ.get("GENPC_C2_SIDE_PRIORITY", "-1") != "0" and getattr(_TLS, "lanes", 1) <= 1 else 0
            _SIDE[key] = _fresh_stream(dev, avoid=(main.cuda_stream,), priority=prio)
        return _SIDE[key]


def prepare_streams(device):
    """The side streams of the single-scan path for the CURRENT stream of `device` (stage 1's, the alignment loop's, the
    sampling check's), made and used once NOW.  Which hardware queue a stream gets is decided at its first use, and a
    completed scan whose side streams came after six lane streams ran at 26.7 scans/s against 28.9 -- call this (or complete a
    scan) before making other streams when single-scan latency matters.  run_in_lanes does."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        prev = getattr(_TLS, "lanes", 1)
        _TLS.lanes = 1
        try:
            side = _side_stream(dev, main)
        finally:
            _TLS.lanes = prev
        if side.cuda_stream != main.cuda_stream:
            with torch.cuda.stream(side):
                torch.zeros(1, device=dev)
            side.synchronize()
        _lib.lib.genpc_streams_prepare(_lib._vp(main.cuda_stream))


def complete_scan(partial_xyz, generated_xyz, generated_img, gt_xyz=None, cfg=None, dp=None, metric_points=16384,
                  fused_points=20000, cd_only_pose=False, overlap=True):
    """partial_xyz [Np,3]: the observed scan; generated_xyz [Ng,3]: points of the generated shape in
    the generator's frame; generated_img [3,1024,1024]: the image the partial points take their
    colours from.  Returns a dict of every stage's outputs."""
    cfg = cfg or default_cfg(partial_xyz.device)
    dp = dp or DepthPrompting(cfg)
    sa = ScaleAdapter(cfg)
    out = {}
    def stage1():
        # ---- stage 1: DepthPrompting.getImage (:100-170) ----
        g = dp.getDepth(partial_xyz)
        # ---- stage 2, first half: ScaleAdapter.colorPoint (:15-50) ----
        return g, sa.colorPoint(g["uv"], generated_img)

    def stage2():
        return reg_xyz.reg(partial_xyz, generated_xyz, generative_model=cfg.generative_model, dataset=cfg.dataset,
                           cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True, cd_only_pose=cd_only_pose)

    def tail(res):
        # the ground truth's metric subsampling (main.py:21) depends on nothing above: it rides along in the fused
        # cloud's FPS launch
        side_fps = [(gt_xyz, metric_points)] if gt_xyz is not None and gt_xyz.shape[0] >= metric_points else None
        fused = reg_xyz.fuse(res["source"], res["target"], num_points=fused_points, side_fps=side_fps)
        gt_idx = None
        if side_fps:
            fused, (gt_idx,) = fused
        t = {"fused": fused}
        # ---- metric: main.metric (main.py:11-36) ----
        if gt_xyz is not None:
            pred = fps_to(fused, metric_points)
            gt = gt_xyz[gt_idx.long()] if gt_idx is not None else fps_to(gt_xyz, metric_points)
            t.update(pred=pred, gt=gt, metric=evaluate_scans(pred[None].contiguous(), gt[None].contiguous())[0])
        return t

    tail_inline = tail

    def tail(res):          # noqa: F811
        # The samplings' device-side check runs BESIDE what follows them (csrc/fps.hip, genpc_fps_defer: ~1 ms of chip-wide kernels
        # behind each of the tail's two samplings otherwise); its verdict is collected here, before anything is returned, and a
        # failed check means the tail again with the check in line.
        # (with several scans in flight the chip is shared already and the check's side stream only adds to the queues: six lanes
        #  42.8 scans/s with it, 47.5 without -- in line there)
        if not (partial_xyz.is_cuda and _FPS_DEFER) or getattr(_TLS, "lanes", 1) > 1:
            return tail_inline(res)
        prev = _lib.lib.genpc_fps_defer(1)
        try:
            t = tail_inline(res)
        finally:
            _lib.lib.genpc_fps_defer(prev)
        bad = _lib.on_device_of(partial_xyz, _lib.lib.genpc_fps_deferred_check)
        if bad < 0:
            raise RuntimeError("genpc_fps_deferred_check failed: " + _lib.last_error())
        if bad:
            from . import fps as _fps
            with _fps._stats_lock:
                _fps.stats["failed_check"] += int(bad)
            t = tail_inline(res)
        return t

    main = side = None
    if overlap and partial_xyz.is_cuda and not _NO_OVERLAP:
        main = torch.cuda.current_stream(partial_xyz.device)
        side = _side_stream(partial_xyz.device, main)
        if side.cuda_stream == main.cuda_stream:        # (torch's stream pool wrapped around onto the caller's stream)
            side = None
    if side is not None:
        # Stage 1 runs under the TAIL of the scan, not under the alignment loop (round 6): the two farthest-point samplings of
        # the tail are ~9 ms on ONE compute unit each (csrc/fps_grid.hip) and leave the chip to the viewpoint selection's
        # chip-wide kernels, while the alignment loop's small dependent launches were delayed by them (stage 1 under the loop
        # hid 2 of its 10 ms; under the tail all of it).
        res = stage2()
        side.wait_stream(main)                      # the inputs are the main stream's
        box = {}
        state = _lib.thread_state()                 # (thread-local library modes and grad mode: the side thread gets the caller's)

        def run():
            try:
                _lib.apply_thread_state(state)
                with torch.cuda.device(partial_xyz.device), torch.cuda.stream(side):
                    box["out"] = stage1()
            except BaseException as e:              # re-raised on the caller's thread
                box["err"] = e

        th = threading.Thread(target=run, name="genpc-stage1")
        th.start()
        try:
            t = tail(res)
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        main.wait_stream(side)                      # stage 1's products are read on the main stream from here on
        g, colors = box["out"]
    else:
        g, colors = stage1()
        res = stage2()
        t = tail(res)
    view, uv, depth = g["view_index"], g["uv"][None], g["depth"][None]
    out.update(view=view, used_opposite=g["used_opposite"], visible=g["visible"], uv=g["uv"], depth=g["depth"], pixels=g["pixels"],
               sparse_img=g["sparse_img"], sparse_depth=g["raw_depth"], hole_mask1=g["hole_mask1"], hole_mask2=g["hole_mask2"])
    out["point_colors"] = colors
    out["reg"] = res
    out["fused"] = t["fused"]
    if gt_xyz is not None:
        out["pred_metric_points"] = t["pred"]
        out["gt_metric_points"] = t["gt"]
        out["metric"] = t["metric"]
    return out


def _lane_stream(device, li):
    """Lane li's stream, made once: torch hands out streams from a pool of 32 per device and wraps around, so a stream made
    per call would sooner or later BE another lane's stage-1 stream -- two host threads on one stream share the library's
    scratch (that is what the memory access fault of a six-lane run after a sweep of lane counts was)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    # (lanes on streams of alternating priority classes, to bring more hardware queues into play: 41.4 scans/s against 46.8)
    with _SIDE_LOCK:
        if (dev, li) not in _LANE:
            _LANE[(dev, li)] = _fresh_stream(dev)
        return _LANE[(dev, li)]


def run_in_lanes(fn, items, lanes, device):
    """fn(lane_index, item) for every item, `lanes` of them in flight: a host thread and a stream of its own per lane (the
    caller's stream is waited for first; every lane's stream is synchronised before this returns).  -> results in order."""
    items = list(items)
    if not items:
        return []
    dev = torch.device(device)
    lanes = max(1, min(int(lanes), len(items), 8))
    caller = torch.cuda.current_stream(dev)
    if __import__("os").environ.get("GENPC_PREPARE_STREAMS", "1") != "0":
        prepare_streams(dev)             # (the single-scan path's side streams get their queues before the lanes' streams exist)
    results, errors = [None] * len(items), []
    nxt = [0]
    lock = threading.Lock()

    state = _lib.thread_state()          # the caller's per-thread modes (arithmetic, kernel choices, grad mode) travel with the work

    def lane(li):
        try:
            _lib.apply_thread_state(state)
            _TLS.lanes = lanes
            if lanes > 1 and __import__("os").environ.get("GENPC_LANES_DUAL", "0") != "1":
                # several scans in flight share the chip already: the alignment loop's second stream (csrc/pose.hip) costs
                # throughput there (six lanes 30 scans/s with it, 40 without) where it saves a scan alone 6 % of its time
                _lib.lib.genpc_pose_dual(0)
            with torch.cuda.device(dev):
                st = _lane_stream(dev, li)
                st.wait_stream(caller)                  # the inputs are the caller's
                with torch.cuda.stream(st):
                    while True:
                        with lock:
                            k = nxt[0]
                            nxt[0] += 1
                        if k >= len(items) or errors:
                            break
                        results[k] = fn(li, items[k])
                st.synchronize()                        # (the lane's products are complete when its thread ends)
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=lane, args=(i,), name="genpc-lane-%d" % i) for i in range(lanes)]
    # GENPC_FPS_COMBINER=1: the lanes' farthest-point samplings leave in shared launches (fps.FpsCombiner).  Measured on
    # config 2's chain, scans/s with / without: four lanes 29.4 / 34.4, six 31.1 / 35.5, eight 35.7 / 36.1 -- the samplings of
    # different lanes are not what the lanes wait for, and a launch of many clouds gives each fewer workgroups: off.
    import contextlib
    with (FpsCombiner.installed(dev) if lanes > 1 and not _NO_COMBINER else contextlib.nullcontext()):
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return results


def complete_scans(jobs, lanes=2, cfg=None, dps=None, **kw):
    """Completed scans per second is a throughput: the stages of ONE scan are chains of small dependent launches, two
    sequential farthest-point samplings and a few chip-wide kernels with host round trips in between -- most of the chip
    idles most of the time -- and independent scans share nothing (BASELINE north_star: "independent scans shard
    embarrassingly").  This runs `lanes` scans at a time on one GPU: a host thread, a stream pair and a DepthPrompting
    object per lane (the library's scratch is keyed by stream; ctypes releases the interpreter lock during a call).

    jobs: sequence of (partial_xyz, generated_xyz, generated_img, gt_xyz) -> list of complete_scan's dicts, in order; every
    scan's products are the bits a call of complete_scan gives (tests/test_gpu_pipeline.py).  dps: one DepthPrompting per
    lane to reuse between calls (built here otherwise)."""
    jobs = list(jobs)
    if not jobs:
        return []
    dev = jobs[0][0].device
    cfg = cfg or default_cfg(dev)
    lanes = max(1, min(int(lanes), len(jobs), 8))      # (throughput peaks at four to six)
    own = {}

    def one(li, job):
        if li not in own:
            own[li] = dps[li] if dps else DepthPrompting(cfg)
        return complete_scan(*job, cfg=cfg, dp=own[li], **kw)

    return run_in_lanes(one, jobs, lanes, dev)
