"""Print the top kernels of a rocprofv3 `--stats --output-format csv` kernel_stats file, per step.
usage: python tools/stats_top.py <kernel_stats.csv> <steps_in_trace> [top_n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = sum(int(r["Calls"]) for r in rows)
print("# GPU kernel time %.2f ms/step, %.0f launches/step" % (tot / steps / 1e6, n / steps))
print("%-92s %10s %9s %9s %6s" % ("kernel", "calls/step", "ms/step", "avg_us", "%"))
for r in rows[:top]:
    print("%-92s %10.1f %9.3f %9.1f %6.1f" % (r["Name"][:92], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6,
                                              float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
