"""Timeline of the last `window_ms` of a rocprofv3 --kernel-trace CSV: start / duration / kernel, for looking at which kernels overlap.
  python tools/trace_timeline.py <kernel_trace.csv> [window_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows = [r for r in rows if "planar::" in r["Kernel_Name"]]
t1 = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t1 - win * 1e6]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("planar::", "").replace("void ", "")
    print("%8.2f +%7.2f ms  %-40s grid %s lds %s vgpr %s+%s" % ((s - t0) / 1e6, (e - s) / 1e6, name[:40], r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"]))
