#!/usr/bin/env python3
"""Every weight-gradient launch of the last traced training step with its grid and duration.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/bench_train.py --batch 16 ; python tools/wgrad_rows.py DIR"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ws = [r for r in rows if "wgrad" in r["Kernel_Name"]]
n = len(ws) // 8
for r in ws[-n:]:
    print("%-16s grid=(%s,%s,%s) wg=%s dur=%8.1f us" % (r["Kernel_Name"].split("::")[-1][:16], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
