"""Durations (us) of one kernel's launches in time order from a rocprofv3 --kernel-trace directory: python tools/trace_seq.py <dir> <name-substring> [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 42
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, 1) for r in rows[-n:]])
