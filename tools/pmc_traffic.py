#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh into profiles/hbm_traffic.json.

Per kernel: HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB).  The factor 2 is the
gfx950 correction for wide coalesced reads: FETCH_SIZE tallies 128-byte requests at 64 bytes.
usage: tools/pmc_traffic.py <pmc dir> <streams> <fixture> > profiles/hbm_traffic.json
"""
import collections
import csv
import glob
import json
import sys

root, streams, fixture = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for p in ("p3", "p4"):
    for fn in glob.glob("%s/%s/*/*counter_collection.csv" % (root, p)):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = {"k_inter_tile<64, 256>": "k_inter_tile64", "k_inter_tile<32, 256>": "k_inter_tile32", "k_inter_tile<16, 64>": "k_inter_tile16",
         "k_inter_blk<unsigned char>": "k_inter_blk", "k_intra<unsigned char>": "k_intra", "k_sao<unsigned char>": "k_sao",
         "k_deblock<unsigned char, 0>": "k_deblock_v", "k_deblock<unsigned char, 1>": "k_deblock_h"}
out = {"streams": streams, "fixture": fixture, "unit": "bytes", "formula": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch", "kernels": {}}
for k, v in acc.items():
    if k not in names or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    f = v["FETCH_SIZE"] / len(disp[k]["FETCH_SIZE"]) * 1024
    w = v["WRITE_SIZE"] / len(disp[k]["WRITE_SIZE"]) * 1024
    out["kernels"][names[k]] = {"fetch_size_bytes_raw": int(f), "write_size_bytes": int(w), "bytes_per_launch": int(2 * f + w)}
json.dump(out, sys.stdout, indent=1)
print()
