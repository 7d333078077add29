"""The alignment loop's Adam update in its two forms -- a launch of its own, or inside the next step's transform
(csrc/pose.hip, GENPC_POSE_FUSE_UPDATE: state and accumulators ping-pong between two sets) -- with the starts of a scan side
by side or one after the other, for odd and even step counts (an odd count ends in the second set; a start that follows
must find both sets clear).  The switches are read once per process, so every form runs in a process of its own; the loss
histories and transforms of all forms agree (the sums are fp64 atomics: last-bit noise only)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import json, math, sys
import numpy as np, torch
sys.path.insert(0, %r)
from genpc_amd.optim_registration import diff_obj_pose as POSE
rng = np.random.default_rng(9)
u = rng.standard_normal((1200, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
complete = (u * np.array([0.5, 0.3, 0.2])).astype(np.float32)
complete[:200] += np.float32([0.15, 0.1, 0.0]) * np.abs(u[:200, :1]).astype(np.float32)
th = math.radians(12.0)
Rt = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
c = complete.mean(0)
full = ((complete - c) * 0.9) @ Rt.T + c + np.array([0.02, -0.01, 0.015])
partial = full[full[:, 2] > -0.05][:600].astype(np.float32)
C, P = torch.from_numpy(complete).cuda(), torch.from_numpy(partial).cuda()
out = {}
for iters in (7, 8):
    T, hist, bp = POSE.object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=iters, render_size=224, return_history=True)
    out[str(iters)] = dict(T=np.asarray(T).tolist(), hist=np.asarray(hist).tolist())
print("RESULT" + json.dumps(out))
''' % ROOT


def _run(fuse, lockstep, flags=1, dual=1, rides=1):
    env = dict(os.environ, GENPC_POSE_FUSE_UPDATE=str(fuse), GENPC_POSE_LOCKSTEP=str(lockstep), GENPC_POSE_DUAL_FLAGS=str(flags),
               GENPC_POSE_DUAL=str(dual), GENPC_POSE_GRAD_RIDES=str(rides))
    p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def test_update_forms_agree():
    ref = _run(0, 1)
    # (the last form: the loop's two streams hand over through events instead of device counters, csrc/pose.hip GENPC_POSE_DUAL_FLAGS)
    # (the last two: one stream -- pose_grad's blocks ride in the silhouette gradient's launch, or have a launch of their own)
    for fuse, lockstep, flags, dual, rides in ((1, 1, 1, 1, 1), (1, 0, 1, 1, 1), (0, 0, 1, 1, 1), (1, 1, 0, 1, 1), (1, 1, 1, 0, 1), (1, 1, 1, 0, 0)):
        got = _run(fuse, lockstep, flags, dual, rides)
        for iters in ("7", "8"):
            h, hr = np.array(got[iters]["hist"]), np.array(ref[iters]["hist"])
            assert h.shape == hr.shape == (4, int(iters) + 1) and np.isfinite(h).all()
            np.testing.assert_allclose(h, hr, rtol=2e-5, err_msg="fuse %d lockstep %d flags %d dual %d rides %d iters %s" % (fuse, lockstep, flags, dual, rides, iters))
            np.testing.assert_allclose(np.array(got[iters]["T"]), np.array(ref[iters]["T"]), atol=2e-5)
