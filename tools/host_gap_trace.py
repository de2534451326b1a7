#!/usr/bin/env python3
"""From a `rocprofv3 --hip-trace --kernel-trace --output-format csv` run of tools/protocol_loop.py: where the time between the last
kernel of an image and the first kernel of the next goes (host API calls on the timeline of the device kernels).
usage: host_gap_trace.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
kt = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
stems = [i for i, k in enumerate(kt) if "k_stem1" in k[2]]
i0, i1 = stems[-12], stems[-11]
last_end = kt[i1 - 1][1]            # last kernel before the next image's stem_1 ... may be a copy kernel
prev_tail = max(k[1] for k in kt[i0:i1] if "k_roi_tail" in k[2])
t0 = prev_tail
print("gap roi_tail end -> next stem_1 start: %.1f us" % ((kt[i1][0] - prev_tail) / 1e3))
for k in kt[i0:i1 + 1]:
    if k[0] >= prev_tail:
        print("  kernel %-40s start +%.1f us  dur %.1f us" % (k[2][:40], (k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3))
for a in api:
    if a[0] >= t0 - 20000 and a[0] <= kt[i1][0] + 5000:
        print("  api    %-40s start %+.1f us  dur %.1f us" % (a[2][:40], (a[0] - t0) / 1e3, (a[1] - a[0]) / 1e3))
