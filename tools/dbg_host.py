import sys, os, time
sys.path.insert(0, os.getcwd())
from hevc_amd import syntax, engine
seq = syntax.read_dump('tests/golden/' + (sys.argv[1] if len(sys.argv) > 1 else 'randomaccess_main_1920x1080') + '.syn.xz')
eng = engine.Engine(seq.seq_params)
syns = []
for pic in seq.pictures:
    syns.append(eng.new_syntax().pack(pic)); pic.drop_ctypes()
prep = [eng.prepare_batch([syns[t]], [seq.pictures[t].pic_id]) for t in range(len(syns))]
eng.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for p in prep: eng.reconstruct_prepared(p)
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    print('submit %.2f ms  total %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
eng.profile_enable(False)
import ctypes as C
n = sum(1 for _ in prep)
print('pictures', n)
