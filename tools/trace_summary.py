#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel timeline of ONE image (between two stem_1 launches)
and per-kernel-name totals per image.  usage: trace_summary.py <dir-with-*_kernel_trace.csv> [image_index_from_end]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_stem1" in r["Kernel_Name"]]
a, b = idx[-k], idx[-k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev = None
tot = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev else 0
    prev = en
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Grid_Size_Z"]))
    print(f"{n:42s} grid={str(g):16s} dur={(en - st) / 1e3:7.2f} gap={gap:5.2f} t={(st - t0) / 1e3:7.1f}")
    c = tot.setdefault(n, [0, 0.0])
    c[0] += 1
    c[1] += (en - st) / 1e3
print("image total us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
for n, (c, t) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print(f"  {n:42s} x{c:3d} {t:8.1f} us")
