"""Host-side packing logic (hevc_amd/csrc/packer.cpp) through the C ABI, without a GPU."""
import ctypes as C

import numpy as np
import pytest

from conftest import STREAMS_240, load_golden
from hevc_amd import abi, engine


def zscan():
    x = np.zeros(256, int)
    y = np.zeros(256, int)
    for z in range(256):
        for b in range(4):
            x[z] |= ((z >> (2 * b)) & 1) << b
            y[z] |= ((z >> (2 * b + 1)) & 1) << b
    return x, y


def expected_counts(seq, pic):
    """Independent count from the raw arrays: 4x4 records inside the picture, how many are intra, and the
    reference samples inter prediction reads (24 per 4x4 block and reference list used)."""
    w, h = seq.width, seq.height
    ctus_w = (w + 63) // 64
    zx, zy = zscan()
    n4 = intra4 = mc = 0
    for a in range(pic.nctu):
        px = (a % ctus_w) * 64 + 4 * zx
        py = (a // ctus_w) * 64 + 4 * zy
        inside = (px < w) & (py < h)
        intra = pic.a["pred_mode"][a] == 1
        n4 += inside.sum()
        intra4 += (inside & intra).sum()
        refs = (pic.a["ref_idx0"][a] >= 0).astype(int) + (pic.a["ref_idx1"][a] >= 0).astype(int)
        mc += (refs * (inside & ~intra)).sum() * 24
    return int(n4), int(intra4), int(mc)


@pytest.mark.parametrize("stream", STREAMS_240)
def test_packed_counts(stream):
    seq = load_golden(stream + "_416x240")
    syn = engine.detached_syntax(seq.seq_params)
    for pic in seq.pictures[:4]:
        syn.pack(pic)
        st = syn.stats()
        n4, intra4, mc = expected_counts(seq, pic)
        assert st[8] == (seq.width // 4) * (seq.height // 4) == n4
        assert st[9] == intra4
        assert st[7] == mc
        assert st[10] == 1
        # every coded coefficient belongs to exactly one block: sum of block areas
        assert st[4] == st[0] * 16 + st[1] * 64 + st[2] * 256 + st[3] * 1024
        if pic.pic_params.slice_type == 2:
            assert st[9] == n4 and st[7] == 0 and st[5] > 0 and st[6] > 0
        assert syn.algorithmic_bytes() > 0
    syn.close()


def test_ctu_order_is_enforced():
    seq = load_golden("intra_main_416x240", 1)
    pic = seq.pictures[0]
    lib = abi.load_library()
    syn = engine.detached_syntax(seq.seq_params)
    ctus = pic.ctu_structs()
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 1, C.byref(ctus[1])) != 0  # CTU 0 first
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 0, C.byref(ctus[0])) == 0
    assert lib.hevcrecon_syntax_finish(syn.h, C.byref(pic.pic_params)) != 0  # not all CTUs added
    syn.close()


def test_bad_sequence_parameters_are_rejected():
    lib = abi.load_library()
    sp = abi.SeqParams(width=416, height=240, bit_depth_luma=8, bit_depth_chroma=8, ctu_size=32, max_cu_depth=3)
    h = C.c_void_p()
    assert lib.hevcrecon_syntax_create_detached(C.byref(sp), C.byref(h)) != 0
    sp = abi.SeqParams(width=417, height=240, bit_depth_luma=8, bit_depth_chroma=8, ctu_size=64, max_cu_depth=4)
    assert lib.hevcrecon_syntax_create_detached(C.byref(sp), C.byref(h)) != 0
