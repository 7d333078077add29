"""Entry points called from several host threads, each on a stream of its own, give the single-threaded results.

Round 4 found a farthest-point sampling that ran next to the f16 nearest-neighbour filter (another stream, another thread)
drawing a sample one step early, silently -- not once in the suite's single-stream tests (csrc/fps.hip: the workers' pivot
reads).  The library keys its scratch by stream and its entry points take the stream as an argument, so this is supported
use; this is the test that was missing.
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(kinds, reps):
    import torch
    from genpc_amd.fps import fps_sampling_multi
    from genpc_amd import chamfer_3D
    from genpc_amd.metric import evaluate_scans
    z = np.load(os.path.join(HERE, "golden", "scans13_fps16384.npz"))

    def ops(k):
        P = torch.from_numpy(z["partial"][k].copy()).cuda()
        G = torch.from_numpy(z["gt"][k].copy()).cuda()

        def fps():
            a, b = fps_sampling_multi([torch.cat([P, G[:8192]]).contiguous(), G], [20000, 16384])
            return torch.cat([a.float(), b.float()])

        def chamfer():
            d1 = torch.empty(1, 16384, device="cuda"); d2 = torch.empty_like(d1)
            i1 = torch.empty(1, 16384, device="cuda", dtype=torch.int32); i2 = torch.empty_like(i1)
            chamfer_3D.forward(P[None].contiguous(), G[None].contiguous(), d1, d2, i1, i2)
            return torch.cat([d1.flatten(), i1.flatten().float(), d2.flatten(), i2.flatten().float()])

        def metric():
            return torch.as_tensor(evaluate_scans(P[None].contiguous(), G[None].contiguous())[0]).float().cuda()

        return {"fps": fps, "chamfer": chamfer, "metric": metric}

    ref = {}
    for k in range(3):
        o = ops(k)
        for n in set(kinds):
            ref[(k, n)] = o[n]().clone()
    torch.cuda.synchronize()
    bad = []

    def work(k, n):
        try:
            st = torch.cuda.Stream()
            o = ops(k)
            with torch.cuda.stream(st):
                for _ in range(reps):
                    r = o[n]()
                    st.synchronize()
                    if not torch.equal(r, ref[(k, n)]):
                        d = (r != ref[(k, n)]).nonzero().flatten()
                        bad.append((n, k, "first difference at", int(d[0]), r[d[0]:d[0] + 3].tolist(), ref[(k, n)][d[0]:d[0] + 3].tolist()))
        except BaseException as e:          # noqa: BLE001 -- reported by the assertion below
            bad.append((n, k, repr(e)))

    th = [threading.Thread(target=work, args=(i % 3, n)) for i, n in enumerate(kinds)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return bad


def test_sampling_next_to_the_f16_filter():
    """Failed on every run before the fix (an extra sample from lanes 48-63 of a worker wave, the rest shifted by one)."""
    assert _run(["fps", "chamfer", "fps", "chamfer"], reps=12) == []


def test_sampling_metric_and_chamfer_together():
    assert _run(["fps", "metric", "chamfer", "fps", "metric", "chamfer"], reps=8) == []
