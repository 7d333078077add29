python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_fps.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.get('extra',{}).items():
    if any(s in k for s in ('c2_pipeline','fps_')) and not isinstance(v,dict): print(k,v)
"
