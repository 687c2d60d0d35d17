"""Average kernel duration per (kernel, grid) from a rocprofv3 --kernel-trace directory: python tools/trace_by_grid.py <dir> [name-substring]"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if len(sys.argv) < 3 or sys.argv[2] in n:
        d[(n[:64], r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print(k, len(v), round(sum(v) / len(v) / 1000, 2))
