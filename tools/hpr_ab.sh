for w in c4 c5 c3; do for l in 1 2 3; do echo "== $w lanes $l"; GENPC_BENCH_LANES=$l python bench.py --workload $w --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['extra']['scan_table_checksum'])"; done; done
