"""pipeline.complete_scan at BASELINE config 2's shape for rocprofv3.   python3 tools/prof_c2.py"""
import os, sys
# (under rocprofv3's kernel tracing every launch costs tens of microseconds more, which biases the alignment loop's own
#  timing probe towards its single-launch path: the traced run takes the path the untraced run takes)
os.environ.setdefault("GENPC_POSE_SEEDED", "0")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import pipeline
from genpc_amd.DepthPrompting import DepthPrompting
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
A = torch.rand(16384, 3, device="cuda", generator=gen) - 0.5
part = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()
img = torch.rand(3, 1024, 1024, device="cuda", generator=gen)
cfg = pipeline.default_cfg("cuda", view_num=1024)
dp = DepthPrompting(cfg)
for _ in range(2):
    pipeline.complete_scan(part, A, img, A, cfg=cfg, dp=dp)
torch.cuda.synchronize()
