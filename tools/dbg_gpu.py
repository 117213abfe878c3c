import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from hevc_amd import syntax, engine
from oracle.refmodel import RefModel
seq = syntax.read_dump('tests/golden/lowdelay_P_main_416x240.syn.xz', 2)
from hevc_amd import abi
eng = engine.Engine(seq.seq_params, lib=abi.load_library(sys.argv[1]) if len(sys.argv) > 1 else None); eng.keep_stages(True)
ref = RefModel(seq.seq_params); syn = eng.new_syntax()
for pic in seq.pictures:
    ref.reconstruct(pic); syn.pack(pic); eng.reconstruct(syn, pic.pic_id)
    got = eng.download_stage(0); want = ref.download(0, pic.pic_id)
    for c in range(3):
        d = np.argwhere(got[c] != want[c])
        print('poc', pic.poc, 'comp', c, 'ndiff', len(d))
        if len(d):
            bs = 16 if c == 0 else 8
            blocks = sorted(set((int(y)//bs*bs, int(x)//bs*bs) for y, x in d))
            print(' blocks', blocks[:40])
            y, x = d[0]
            print(' first', y, x, got[c][y, x:x+8], want[c][y, x:x+8])
    if pic.poc == 1:
        d = np.argwhere(got[0] != want[0])
        print(len(d), d[:20].tolist())
    if pic.poc == 1:
        print('got ', got[0][14:20, 8:34])
        print('want', want[0][14:20, 8:34])
