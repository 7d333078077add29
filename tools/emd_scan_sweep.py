"""13 bundled scans (partial vs GT), EMD 50 rounds through the launch-per-round path (tune 1): one timing per process
setting (the GENPC_* switches are read once).   python3 tools/emd_scan_sweep.py"""
import os, sys, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from genpc_amd import _lib
    from genpc_amd.loss_functions import emdModule
    em = emdModule(); _lib.lib.genpc_emd_tune(1, -1)
    z = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
    P, G = torch.from_numpy(z["partial"]).cuda(), torch.from_numpy(z["gt"]).cuda()
    for _ in range(2): d, a = em(P, G, 0.005, 50)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): d, a = em(P, G, 0.005, 50)
    e1.record(); e1.synchronize()
    print("%.3f ms  checksum %d" % (e0.elapsed_time(e1) / 3, int(a.long().sum())))
    sys.exit(0)
settings = [{}, {"GENPC_EMD_GRID_PPC_X10": "40"}, {"GENPC_EMD_GRID_PPC_X10": "80"}, {"GENPC_EMD_GRID_PPC_X10": "10"},
            {"GENPC_EMD_LPB": "16"}, {"GENPC_EMD_LPB": "32"}, {"GENPC_EMD_LPB": "64"}, {"GENPC_EMD_G": "512"}, {"GENPC_EMD_G": "160"},
            {"GENPC_EMD_RESOLVE_FROM": "4"}, {"GENPC_EMD_RESOLVE_FROM": "1000"}, {"GENPC_EMD_XCD": "1"}]
for s in settings:
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **s), capture_output=True, text=True)
    print(s, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
