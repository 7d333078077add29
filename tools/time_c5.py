import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
from genpc_amd.metric import evaluate_scans
dev = "cuda"
sc = [bench.synth_scan(k, 32768) for k in range(8)]
C5 = torch.from_numpy(np.stack([x[0] for x in sc])).to(dev)
P5 = torch.from_numpy(np.stack([x[1] for x in sc])).to(dev)
G5 = torch.from_numpy(np.stack([x[2] for x in sc])).to(dev)
print(C5.shape, P5.shape, G5.shape)
for hooks in (0, 2048):
    from genpc_amd import _lib
    _lib.lib.genpc_nn_tune(-1, hooks)
    object_pose_optimization(C5, P5, radius=0.02, lr=0.01, iters=20, render_size=224)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    object_pose_optimization(C5, P5, radius=0.02, lr=0.01, iters=200, render_size=224)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    evaluate_scans(C5, G5)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("hooks", hooks, "pose %.3f s, metric %.3f s" % (t1 - t0, t2 - t1))
