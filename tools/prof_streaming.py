"""Profiling driver (tools/collect_profiles.sh family streaming_64x32768): the HBM-bound rows of SURVEY 8(a) at 64 x 32768
points -- chamfer backward (a3), CalcDist (a8), EMD backward (a10), the pose point map, paintPixels (a14), the colour
gather (a15) -- a few launches each (genpc_amd/streaming_bench.py holds the byte models)."""
import os, sys, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import streaming_bench
print(json.dumps(streaming_bench.rooflines(torch.device("cuda:0"), reps=3)))
