#!/usr/bin/env python3
"""Mean PMC counters per launch of the conv kernels in a rocprofv3 --pmc counter_collection.csv: pmc_conv.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:46]
    if "conv" in n:
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n)
    for k, v in sorted(d.items()):
        print("   %-28s launches=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
