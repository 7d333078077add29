bash tools/ktrace_any.sh c2scan tools/prof_c2_scan.py
tail -2 gpurun_out/kt_c2scan/t.log | head -1
grep "one scan" gpurun_out/kt_c2scan/t.log
