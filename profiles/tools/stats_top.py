"""Print the top rows of a rocprofv3 kernel_stats.csv: python stats_top.py <dir> [rows]."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%s: total kernel time %.3f ms in %d launches" % (f, tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-84s %7s %9.3f ms %9.2f us" % (r["Name"][:84], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
