"""Print the top kernels of a rocprofv3 `--stats --output-format csv` kernel_stats file, per step.
usage: python tools/stats_top.py <kernel_stats.csv> <steps_in_trace> [top_n] [summary.json ore_version]
With the last two arguments the per-step totals are also written as JSON (bench.py reads profiles/r*_train_step_*_summary.json into the
train legs' roofline: launches_per_step, kernel_ms_per_step -- only when the file's ore_version is the library's)."""
import csv
import json
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
if len(sys.argv) > 5:
    with open(sys.argv[4], "w") as f:
        json.dump({"ore_version": int(sys.argv[5]), "launches_per_step": round(n / steps, 1), "kernel_ms_per_step": round(tot / steps / 1e6, 3),
                   "steps_in_trace": steps, "from": sys.argv[1].split("/")[-1]}, f)
