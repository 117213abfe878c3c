#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh (gpurun_out/pmc/p*/)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for p in sorted(glob.glob(root + "/p*/")):
    fs = sorted(glob.glob(p + "*/*counter_collection.csv"), key=os.path.getmtime)[-1:]  # the newest run of the pass
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    print("==", p)
    for k, v in agg.items():
        n = len(disp[k])
        print("%-36s launches %5d  " % (k[:36], n) + "  ".join("%s=%.0f" % (a, b / n) for a, b in sorted(v.items())))
