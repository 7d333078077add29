python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.get('extra',{}).items():
    if any(s in k for s in ('registration',)) and not isinstance(v,dict): print(k,v)
"
