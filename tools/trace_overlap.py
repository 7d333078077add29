"""Start / end of consecutive kernels in a rocprofv3 kernel trace (do launches on two streams overlap?).
   python3 tools/trace_overlap.py gpurun_out/kt_<tag>/t/t_kernel_trace.csv [first] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 120
count = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sel = [r for r in rows if "genpc" in r["Kernel_Name"]]
t0 = int(sel[first]["Start_Timestamp"])
for r in sel[first:first + count]:
    name = r["Kernel_Name"].replace("void genpc::", "").replace("genpc::", "")[:36]
    print("%-36s q%-3s start %9.2f end %9.2f dur %7.2f" % (name, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                          (int(r["End_Timestamp"]) - t0) / 1e3,
                                                          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
