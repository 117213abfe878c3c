#!/usr/bin/env python3
"""Turns the counter passes of tools/pmc_passes.sh into one configuration's entry of profiles/hbm_traffic.json.

Per kernel and launch:
  bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB).  The factor 2 is the gfx950 correction for
                     wide coalesced reads: FETCH_SIZE tallies 128-byte requests at 64 bytes.
  valu_per_wave    = SQ_INSTS_VALU / SQ_WAVES
  valu_issue_frac  = 4 * SQ_INSTS_VALU / (1024 * SQ_BUSY_CYCLES / 32): a vector instruction occupies one of the
                     device's 1024 SIMDs for 4 cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines, each of
                     which counts the cycles of the launch.  1.0 = every SIMD issues vector arithmetic all the time.
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of the resident waves' time spent parked on memory or a barrier

usage: tools/pmc_traffic.py <pmc dir> <pictures per launch> <fixture> <clips> [file to update in place]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, streams, fixture = sys.argv[1], int(sys.argv[2]), sys.argv[3]
clips = int(sys.argv[4]) if len(sys.argv) > 4 else 1
target = sys.argv[5] if len(sys.argv) > 5 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for p in ("p1", "p3", "p4"):
    # (the newest file of a pass: results merged back from several runs may lie side by side)
    for fn in sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (root, p)), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])


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


entry = {"streams": streams, "fixture": fixture, "clips": clips, "unit": "bytes", "kernels": {}}
# several instantiations can share a short name (k_intra with and without inter pictures): the one with more launches wins
best = {}
for k, v in acc.items():
    if not k.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    n = len(disp[k]["FETCH_SIZE"])
    if best.get(short_name(k), (0, None))[0] >= n:
        continue
    best[short_name(k)] = (n, k)
for name, (n, k) in sorted(best.items()):
    v = acc[k]
    f = v["FETCH_SIZE"] / len(disp[k]["FETCH_SIZE"]) * 1024
    w = v["WRITE_SIZE"] / len(disp[k]["WRITE_SIZE"]) * 1024
    e = {"launches": n, "fetch_size_bytes_raw": int(f), "write_size_bytes": int(w), "bytes_per_launch": int(2 * f + w)}
    if v.get("SQ_WAVES") and v.get("SQ_BUSY_CYCLES"):
        e["valu_per_wave"] = round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1)
        e["valu_issue_frac"] = round(4 * v["SQ_INSTS_VALU"] / (1024 * v["SQ_BUSY_CYCLES"] / 32), 4)
        e["wait_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4)
    entry["kernels"][name] = e
if target:
    try:
        doc = json.load(open(target))
    except (OSError, ValueError):
        doc = {}
    if "configs" not in doc:
        doc = {"formula": "bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; valu_issue_frac = 4*SQ_INSTS_VALU / (1024 * SQ_BUSY_CYCLES/32); see tools/pmc_traffic.py",
               "configs": []}
    doc["configs"] = [c for c in doc["configs"] if (c["fixture"], c["streams"], c["clips"]) != (fixture, streams, clips)] + [entry]
    json.dump(doc, open(target, "w"), indent=1)
    print("updated", target)
else:
    json.dump(entry, sys.stdout, indent=1)
    print()
