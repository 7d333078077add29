python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
for m in 1 0; do echo "== cu mask $m"; GENPC_C2_CU_MASK=$m python tools/prof_c2_scan.py 2>&1 | tail -1
GENPC_C2_CU_MASK=$m python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.get('extra',{}).items():
    if any(s in k for s in ('c2_pipeline',)) and not isinstance(v,dict): print(k,v)
"; done
