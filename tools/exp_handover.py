#!/usr/bin/env python3
"""Where a pass goes when every picture's packed syntax crosses the host link again: one context, per-kernel HIP-event
times with and without the hand-over, wall clock of both.  usage: tools/exp_handover.py [streams]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hevc_amd import syntax  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
golden = os.path.join(bench.ROOT, "tests", "golden")
names = ["randomaccess_main_1920x1080"] + ["randomaccess_main_1920x1080_s%d" % k for k in range(1, 8)]
seqs = [syntax.read_dump(os.path.join(golden, n + ".syn.xz")) for n in names]
W = bench.Workload(seqs, S, 1, 0, True, False)
eng = W.engs[0]
g = 0
for _ in range(W.npic):
    W.step(g)
    g += 1
W.synchronize()
for upload in (False, True, False, True):
    for prof in (False, True):
        eng.profile_enable(prof)
        t0 = time.perf_counter()
        for _ in range(W.npic):
            W.step(g, upload=upload)
            g += 1
        t_submit = time.perf_counter() - t0
        W.synchronize()
        dt = time.perf_counter() - t0
        line = "upload=%d profile=%d: %.1f ms per pass (host done submitting after %.1f ms), %.0f frames/s" % (upload, prof, dt * 1e3, t_submit * 1e3, W.npic * S / dt)
        if prof:
            p = eng.profile_read()
            line += "  " + ", ".join("%s %.2f" % (k, v[0]) for k, v in p.items() if v[0] > 0)
        print(line, flush=True)
W.close()
