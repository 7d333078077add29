python -m pytest tests/test_gpu_determinism.py tests/test_gpu_pipeline.py tests/test_gpu_waymo_c4.py -x -q 2>&1 | tail -3
for m in 2 0; do echo "== mode $m"; GENPC_POSE_SEEDED=$m python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.get('extra',{}).items():
    if any(s in k for s in ('c2_pipeline','registration','c5_rank')) and not isinstance(v,dict): print(k,v)
"; done
