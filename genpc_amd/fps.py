"""Farthest point sampling on the gfx950 library: the deterministic counterpart of
``fpsample.fps_sampling(points, k)`` as the reference calls it (main.py:21-24,
reg_xyz.py:215, DepthPrompting.py:88).  Start index 0 (fpsample starts at a random
index, so the reference's own subsamples are not reproducible)."""
import torch

from . import _lib

_L = _lib.lib
_p = _lib.ptr
# A/B switch (tools/time_c2_lanes.py): one sampling launch at a time in the process, whatever threads / streams call
import os as _os
import threading as _threading
_ONE_AT_A_TIME = _threading.Lock() if _os.environ.get("GENPC_FPS_LOCK", "0") == "1" else None
# what the device-side verification has seen in this process (tools/soak_lanes.py prints it): clouds sampled, clouds whose
# hand-off timed out (first index -1) or whose sequence failed the check -- a step's sample was not the first arg-max
# (first index -2) -- and were sampled again
stats = {"clouds": 0, "timed_out": 0, "failed_check": 0}
_stats_lock = _threading.Lock()


class FpsCombiner:
    """Samplings of several host threads in ONE launch.  A sampling is k sequential steps per cloud, its launch needs its
    workgroups co-resident (csrc/fps.hip) and takes a large share of the admission budget: samplings of different streams
    run one after the other however many scans are in flight -- 2 x 8.7 ms per completed scan of config 2, the largest
    serial item of pipeline.complete_scans.  Independent clouds side by side in one launch cost the longest chain, not the
    sum: while a launch is in flight the requests of the other threads queue up here and leave together in the next one
    (a leader among the waiting threads runs it on the combiner's stream; every requester's stream waits for the launch's
    event).  Results are the bits a call of its own gives: clouds never interact.  (Opt-in, GENPC_FPS_COMBINER=1 in
    pipeline.run_in_lanes: measured, it does not raise the scans in flight -- the numbers are there.)

        with FpsCombiner.installed(device): ...      # fps_sampling / fps_sampling_multi of ANY thread go through it
    """
    _current = None

    def __init__(self, device):
        d = torch.device(device)
        self.device = torch.device("cuda", torch.cuda.current_device() if d.index is None else d.index)
        self.stream = torch.cuda.Stream(device=self.device)
        self.cond = _threading.Condition()
        self.pending = []
        self.busy = False
        self.launches = 0
        self.clouds = 0

    class _Req:
        __slots__ = ("clouds", "ks", "ready", "outs", "done", "err", "finished")

    def submit(self, clouds, ks):
        lane = torch.cuda.current_stream(self.device)
        r = FpsCombiner._Req()
        r.clouds, r.ks, r.outs, r.err, r.finished = clouds, ks, None, None, False
        r.ready = torch.cuda.Event()
        r.ready.record(lane)                     # the clouds are the lane's products
        batch = None
        with self.cond:
            self.pending.append(r)
            while not r.finished:
                if not self.busy:
                    batch, self.pending, self.busy = self.pending, [], True
                    break
                self.cond.wait()
        if batch is not None:                    # this thread leads: one launch for everything that queued up
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                    for q in batch:
                        self.stream.wait_event(q.ready)
                    flat_c = [c for q in batch for c in q.clouds]
                    flat_k = [k for q in batch for k in q.ks]
                    outs = _fps_multi_direct(flat_c, flat_k)         # (reads the first indices on the host: returns when the launch is done)
                    done = torch.cuda.Event()
                    done.record(self.stream)
                at = 0
                for q in batch:
                    q.outs, q.done = outs[at:at + len(q.clouds)], done
                    at += len(q.clouds)
                self.launches += 1
                self.clouds += len(flat_c)
            except BaseException as e:
                for q in batch:
                    q.err = e
            finally:
                with self.cond:
                    self.busy = False
                    for q in batch:
                        q.finished = True
                    self.cond.notify_all()
        if r.err is not None:
            raise r.err
        lane.wait_event(r.done)
        for o in r.outs:
            o.record_stream(lane)                # (allocated on the combiner's stream, read on the lane's)
        return r.outs

    @classmethod
    def installed(cls, device):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev = cls._current
            cls._current = cls(device)
            try:
                yield cls._current
            finally:
                cls._current = prev
        return ctx()


def fps_sampling(points, k):
    """points: [N,3] or [C,N,3] GPU tensor -> int32 indices [k] or [C,k].  The first
    index is always 0; the kernel writes -1 there if its inter-workgroup hand-off timed
    out (another kernel kept the cloud's workgroups from being co-resident)."""
    single = points.dim() == 2
    if FpsCombiner._current is not None and points.is_cuda:
        pts = (points[None] if single else points).contiguous().float()
        outs = fps_sampling_multi([pts[j] for j in range(pts.shape[0])], [int(k)] * pts.shape[0])
        return outs[0] if single else torch.stack(outs)
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
    # between its workgroups timed out (the samples after it would be garbage): those clouds go again, one at a time
    bad = torch.nonzero(out[:, 0] != 0).flatten().tolist()
    for j in bad:
        out[j] = _fps_multi_direct([pts[j]], [k])[0]
    return out[0] if single else out


def fps_sampling_multi(clouds, ks):
    """Several clouds of different sizes / sample counts in ONE pass (FPS is latency-bound: k sequential
    steps per cloud, so independent clouds side by side cost the longest one, not the sum).
    clouds: list of [N_j,3] GPU tensors, ks: list of ints -> list of int32 index tensors [k_j]."""
    comb = FpsCombiner._current
    if comb is not None and len(clouds) > 0 and clouds[0].is_cuda and clouds[0].device == comb.device:
        pts = [c.contiguous().float() for c in clouds]
        if len(ks) != len(pts) or any(p.dim() != 2 or p.shape[1] != 3 for p in pts):
            raise ValueError("fps_sampling_multi: need one k per [N,3] cloud")
        if any(not (0 < int(k) <= p.shape[0] <= 262144) for p, k in zip(pts, ks)):
            raise ValueError("fps_sampling_multi: need 0 < k <= N <= 262144 for every cloud")
        return comb.submit(pts, [int(k) for k in ks])
    return _fps_multi_direct(clouds, ks)


def _fps_multi_direct(clouds, ks, _attempt=0):
    import ctypes
    pts = [c.contiguous().float() for c in clouds]
    _lib.check_tensors(tuple(("clouds[%d]" % j, p) for j, p in enumerate(pts)))
    c = len(pts)
    if c == 0:
        return []
    if len(ks) != c or any(p.dim() != 2 or p.shape[1] != 3 for p in pts):
        raise ValueError("fps_sampling_multi: need one k per [N,3] cloud")
    dev = pts[0].device
    outs = [torch.empty(int(k), device=dev, dtype=torch.int32) for k in ks]
    n_arr = (ctypes.c_int * c)(*[int(p.shape[0]) for p in pts])
    k_arr = (ctypes.c_int * c)(*[int(k) for k in ks])
    x_arr = (ctypes.c_void_p * c)(*[p.data_ptr() for p in pts])
    o_arr = (ctypes.c_void_p * c)(*[o.data_ptr() for o in outs])
    if _ONE_AT_A_TIME is not None:
        with _ONE_AT_A_TIME:
            rc = _lib.on_device_of(pts[0], _L.genpc_fps_multi, c, ctypes.addressof(n_arr), ctypes.addressof(k_arr),
                                   ctypes.addressof(x_arr), ctypes.addressof(o_arr))
            torch.cuda.current_stream(dev).synchronize()
    else:
        rc = _lib.on_device_of(pts[0], _L.genpc_fps_multi, c, ctypes.addressof(n_arr), ctypes.addressof(k_arr),
                               ctypes.addressof(x_arr), ctypes.addressof(o_arr))
    if rc == -1:
        raise ValueError("fps_sampling_multi: need 0 < k <= N <= 262144 for every cloud")
    if rc != 1:
        raise RuntimeError("genpc_fps_multi failed: " + _lib.last_error())
    first = torch.stack([o[0] for o in outs]).tolist()          # (one host read for all clouds)
    bad = [j for j, f in enumerate(first) if f != 0]
    with _stats_lock:
        stats["clouds"] += c if _attempt == 0 else 0
        stats["timed_out"] += sum(1 for j in bad if first[j] != -2)
        stats["failed_check"] += sum(1 for j in bad if first[j] == -2)
    if bad and (c > 1 or _attempt < 3):
        # out[0] == -1: a hand-off timed out (something else on the GPU kept a cloud's workgroups from running together), or
        # the device-side verification (csrc/fps.hip: fps_verify_kernel) found a step whose sample is not the first arg-max --
        # seen twice for samplings running beside other streams' kernels, cause unknown.  Not an error yet: the clouds that
        # failed go again, one at a time (a launch to itself needs a fraction of the device)
        for j in bad:
            outs[j] = _fps_multi_direct([pts[j]], [ks[j]], _attempt=_attempt + 1 if c == 1 else _attempt)[0]
        return outs
    if bad:
        raise RuntimeError("genpc_fps: no verified sampling after %d attempts (hand-off timed out or the sequence failed its "
                           "device-side check every time)" % (_attempt + 1))
    return outs


def fps_subsample(points, k):
    """The helper metric.py calls but never defines: [B,N,3] -> [B,k,3]."""
    idx = fps_sampling(points, k).long()
    if points.dim() == 2:
        return points[idx]
    return torch.gather(points, 1, idx[..., None].expand(-1, -1, 3))
