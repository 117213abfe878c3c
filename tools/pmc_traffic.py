#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh into profiles/hbm_traffic.json.

Per kernel: HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB).  The factor 2 is the
gfx950 correction for wide coalesced reads: FETCH_SIZE tallies 128-byte requests at 64 bytes.
usage: tools/pmc_traffic.py <pmc dir> <pictures per launch> <fixture> <clips> > profiles/hbm_traffic.json
"""
import collections
import csv
import glob
import json
import sys

root, streams, fixture = sys.argv[1], int(sys.argv[2]), sys.argv[3]
clips = int(sys.argv[4]) if len(sys.argv) > 4 else 1
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for p in ("p3", "p4"):
    for fn in glob.glob("%s/%s/*/*counter_collection.csv" % (root, p)):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
import re


def short_name(k):
    m = re.match(r"k_inter_tile<(?:[^,]*, )?(\d+),", k)
    if m:
        return "k_inter_tile" + m.group(1)
    m = re.match(r"k_deblock<[^,]*, (\d)>", k)
    if m:
        return "k_deblock_v" if m.group(1) == "0" else "k_deblock_h"
    m = re.match(r"k_transform<[^,]*, (\d)>", k)
    if m:
        return "k_transform%d" % (1 << int(m.group(1)))
    return k.split("<")[0]


out = {"streams": streams, "fixture": fixture, "clips": clips, "unit": "bytes", "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch", "kernels": {}}
for k, v in acc.items():
    if not k.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    f = v["FETCH_SIZE"] / len(disp[k]["FETCH_SIZE"]) * 1024
    w = v["WRITE_SIZE"] / len(disp[k]["WRITE_SIZE"]) * 1024
    out["kernels"][short_name(k)] = {"fetch_size_bytes_raw": int(f), "write_size_bytes": int(w), "bytes_per_launch": int(2 * f + w)}
json.dump(out, sys.stdout, indent=1)
print()
