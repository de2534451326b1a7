#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE per kernel family over the LAST forward of tools/pmc_pass.py.
usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [ore_version].  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts
128-B read requests at 64 B -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Units: the counters are in KiB? -- no:
rocprofv3 reports the raw derived value in BYTES/1024; we calibrate on k_maxpool (known bytes) and store the factor used."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    per_dispatch = defaultdict(float)
    names = {}
    for r in rows:
        k = int(r["Dispatch_Id"])
        per_dispatch[k] += float(r["Counter_Value"])
        names[k] = r["Kernel_Name"]
    ids = sorted(per_dispatch)
    start = max(i for i in ids if "k_stem1" in names[i])       # three identical forwards: keep the last one (warm), it starts at stem_1
    return [(names[i], per_dispatch[i]) for i in ids if i >= start]


def fam(n):
    for k in ("k_conv3x3_wino", "k_conv3x3_patch", "k_conv3x3_ws", "k_conv_igemm", "k_conv_kw", "k_conv_kd", "k_conv_gd", "k_conv_rf", "k_conv_gs", "k_head_pred", "k_stem1", "k_maxpool", "k_correlation", "k_roi_align", "k_nms", "k_level_select"):
        if k in n:
            return k
    return "other"


def main():
    fd, wd, out = sys.argv[1:4]
    ver = int(sys.argv[4]) if len(sys.argv) > 4 else -1        # ore_version() of the library the passes ran with (bench.py checks it)
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = defaultdict(lambda: {"launches": 0, "fetch_raw": 0.0, "write_raw": 0.0})
    for n, v in fetch:
        res[fam(n)]["launches"] += 1
        res[fam(n)]["fetch_raw"] += v
    for n, v in write:
        res[fam(n)]["write_raw"] += v
    tot = {}
    for k, v in res.items():
        v["hbm_bytes"] = (2.0 * v["fetch_raw"] + v["write_raw"]) * 1024.0      # KiB -> bytes, FETCH_SIZE doubled on gfx950
        tot[k] = v
    conv = sum(v["hbm_bytes"] for k, v in res.items() if k.startswith("k_conv"))          # every conv family (k_head_pred is priced with "other", as in rounds 1-3)
    json.dump({"per_image": tot, "conv_hbm_bytes_per_image": conv, "ore_version": ver,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over eager forwards (tools/pmc_pass.py), last "
                         "forward; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: FETCH_SIZE halves wide reads on gfx950)"},
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda t: -t[1]["hbm_bytes"]):
        print("%-18s launches=%3d FETCH_SIZE=%10.1f KiB WRITE_SIZE=%10.1f KiB -> %8.2f MB" % (k, v["launches"], v["fetch_raw"], v["write_raw"], v["hbm_bytes"] / 1e6))
    print("conv kernels: %.1f MB per image; all kernels: %.1f MB" % (conv / 1e6, sum(v["hbm_bytes"] for v in res.values()) / 1e6))
    print("-- launch by launch (last forward): 2*FETCH_SIZE + WRITE_SIZE")
    wd_ = dict()
    for i, (n, v) in enumerate(write):
        wd_[i] = v
    for i, (n, v) in enumerate(fetch):
        w = wd_.get(i, 0.0)
        print("  %-60s fetch %9.1f KiB x2  write %9.1f KiB -> %7.2f MB" % (n[:60], v, w, (2 * v + w) * 1024 / 1e6))


if __name__ == "__main__":
    main()
