#!/usr/bin/env python3
"""What the submitting host thread pays per batch (experiment; run on a GPU machine from the repository root).

Submits fewer batches than the engine's ring holds (so no call ever waits for the device) after a full synchronisation
and times the calls alone: resident pictures, and pictures marked for another trip over the host link."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hevc_amd import syntax  # noqa: E402

golden = os.path.join(ROOT, "tests", "golden")
names = ["randomaccess_main_1920x1080"] + ["randomaccess_main_1920x1080_s%d" % k for k in range(1, 8)]
seqs = [syntax.read_dump(os.path.join(golden, n + ".syn.xz")) for n in names]
W = bench.Workload(seqs, 64, 1, 0, True, False, False)
g = W.passes(W.ramp_up(), 1)
out = {}
for upload in (False, True):
    for rep in range(2):
        W.synchronize()
        t0 = time.perf_counter()
        for gg in range(g, g + 12):
            W.step(gg, upload=upload)
        t1 = time.perf_counter()
        W.synchronize()
        t2 = time.perf_counter()
        g += 12
    out["upload" if upload else "resident"] = {"host_us_per_batch": round((t1 - t0) / 12 * 1e6, 1), "device_us_per_batch": round((t2 - t0) / 12 * 1e6, 1)}
print(json.dumps(out))
W.close()
