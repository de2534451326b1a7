#!/usr/bin/env python3
"""Mean PMC counters per launch of the kernels whose name matches a regex in a rocprofv3 --pmc run: pmc_any.py <dir> <regex>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
    if pat.search(n):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n)
    for k, v in sorted(d.items()):
        print("   %-28s launches=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
