#!/usr/bin/env python3
"""Where does the time go when uploads and reconstruction are interleaved?  (debug helper, GPU only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hevc_amd import engine, syntax
from bench import clone_for_stream, ID_STRIDE

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seq = syntax.read_dump(os.path.join(ROOT, "tests", "golden", "randomaccess_main_1920x1080.syn.xz"))
npic = len(seq.pictures)
eng = engine.Engine(seq.seq_params, device=0)
syns = [[None] * S for _ in range(npic)]
for t, pic in enumerate(seq.pictures):
    pic.ctu_structs()
    for s in range(S):
        syns[t][s] = eng.new_syntax().pack(clone_for_stream(pic, s))
    pic.drop_ctypes()
ids = [[pic.pic_id + s * ID_STRIDE for s in range(S)] for pic in seq.pictures]
prepared = [eng.prepare_batch(syns[t], ids[t]) for t in range(npic)]
eng.synchronize()
for t in range(npic):
    eng.reconstruct_prepared(prepared[t])
eng.synchronize()

def run(name, up, rec, every=1):
    t0 = time.perf_counter()
    tu = tr = 0.0
    worst = 0.0
    for t in range(npic):
        a = time.perf_counter()
        if up:
            for s in range(0, S, every):
                a1 = time.perf_counter()
                syns[t][s].upload()
                worst = max(worst, time.perf_counter() - a1)
        b = time.perf_counter()
        if rec:
            eng.reconstruct_prepared(prepared[t])
        c = time.perf_counter()
        tu += b - a
        tr += c - b
    t1 = time.perf_counter()
    print("   uploads %.1f ms (worst single %.2f ms), reconstruct calls %.1f ms" % (tu * 1e3, worst * 1e3, tr * 1e3))
    eng.synchronize()
    t2 = time.perf_counter()
    print("%-28s host loop %7.1f ms, until idle %7.1f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)

run("reconstruct only", False, True)
run("upload only", True, False)
run("upload + reconstruct", True, True)
run("upload 1/8 + reconstruct", True, True, 8)
run("upload 1/64 + reconstruct", True, True, 64)
run("reconstruct only", False, True)
