"""print a rocprofv3 kernel_stats.csv with short kernel names"""
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
print("%-70s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[1:]:
    name = re.sub(r"\(dr::[A-Za-z]*\)$|\(.*\)$", "", r[0].replace("(anonymous namespace)::", ""))[:70]
    print("%-70s %8s %12.1f %10.2f %7s" % (name, r[1], float(r[2]) / 1e3, float(r[3]) / 1e3, r[4]))
