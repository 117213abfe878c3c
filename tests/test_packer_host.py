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
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 1, C.byref(ctus[1])) != 0  # CTU 0 first (coding order = raster order without tiles)
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


def test_hand_over_is_compact():
    """The 4x4 records travel run-length coded and the coefficients sparse: far fewer runs than records, and the
    sparse stream never holds more words than one per coded coefficient plus one per block."""
    seq = load_golden("randomaccess_main_416x240")
    syn = engine.detached_syntax(seq.seq_params)
    for pic in seq.pictures:
        syn.pack(pic)
        st = syn.stats()
        nctu = pic.nctu
        assert nctu <= st[30] <= st[8]           # at least one run per CTU, at most one per record
        assert st[30] * 4 < st[8]                # and a real saving on this stream
        blocks = st[0] + st[1] + st[2] + st[3]
        nonzero = sum(int(np.count_nonzero(pic.a["coeff%d" % c])) for c in range(3))
        assert 4 * blocks <= st[31] <= 4 * (blocks + min(st[4], nonzero))
    syn.close()


def test_prediction_weights_are_range_checked():
    seq = load_golden("wpp_416x240", 2)
    lib = abi.load_library()
    syn = engine.detached_syntax(seq.seq_params)
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    w = abi.PredWeights.from_buffer_copy(bytes(seq.pictures[1].pred_weights))
    assert lib.hevcrecon_syntax_set_pred_weights(syn.h, C.addressof(w)) == 0
    w.log2_denom[0][0][0] = 8
    assert lib.hevcrecon_syntax_set_pred_weights(syn.h, C.addressof(w)) != 0
    w.log2_denom[0][0][0] = 6
    w.weight[1][3][2] = 300
    assert lib.hevcrecon_syntax_set_pred_weights(syn.h, C.addressof(w)) != 0
    assert lib.hevcrecon_syntax_set_pred_weights(syn.h, None) == 0
    syn.close()


def test_ipcm_unit_without_samples_is_refused():
    seq = load_golden("pcmlf_416x240", 1)
    pic = seq.pictures[0]
    lib = abi.load_library()
    syn = engine.detached_syntax(seq.seq_params)
    ctus = pic.ctu_structs()
    a = sorted(pic.pcm)[0]  # a CTU that holds IPCM units
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    for i in range(a):
        assert lib.hevcrecon_syntax_add_ctu(syn.h, i, C.byref(ctus[i])) == 0
    broken = abi.CtuSyntax.from_buffer_copy(bytes(ctus[a]))
    broken.pcm[0] = None
    assert lib.hevcrecon_syntax_add_ctu(syn.h, a, C.byref(broken)) != 0
    syn.close()


@pytest.mark.parametrize("field,value,what", [("qp", 52, "QP"), ("qp", -1, "QP"), ("luma_dir", 35, "intra prediction mode"),
                                              ("tr_idx", 5, "transform block size")])
def test_out_of_range_syntax_is_refused(field, value, what):
    """Values that would index tables on the device are refused by hevcrecon_syntax_add_ctu, and the picture stays
    refused until hevcrecon_syntax_begin starts it again."""
    seq = load_golden("intra_main_416x240", 1)
    pic = seq.pictures[0]
    lib = abi.load_library()
    syn = engine.detached_syntax(seq.seq_params)
    ctus = pic.ctu_structs()
    broken = abi.CtuSyntax.from_buffer_copy(bytes(ctus[0]))
    arr = pic.a[field][0].copy()
    arr[:] = value
    setattr(broken, field, arr.ctypes.data)
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 0, C.byref(broken)) != 0
    assert what in lib.hevcrecon_syntax_last_error(syn.h).decode()
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 0, C.byref(ctus[0])) != 0
    assert lib.hevcrecon_syntax_finish(syn.h, C.byref(pic.pic_params)) != 0
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 0, C.byref(ctus[0])) == 0
    syn.close()


def test_slice_table_is_checked():
    """hevcrecon_syntax_set_slices: any time between begin and finish (a decoder learns of a picture's slices as they
    arrive), values in range; when the picture closes every CTU's slice must be in the table and the slices must end
    at CTU boundaries."""
    seq = load_golden("slices_nolf_416x240", 1)
    pic = seq.pictures[0]
    lib = abi.load_library()
    syn = engine.detached_syntax(seq.seq_params)
    ctus = pic.ctu_structs()
    n = len(pic.slices)
    table = (abi.SliceParams * n).from_buffer_copy(pic.slices)
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    table[1].tc_offset_div2 = 7
    assert lib.hevcrecon_syntax_set_slices(syn.h, n, C.addressof(table)) != 0
    assert "out of range" in lib.hevcrecon_syntax_last_error(syn.h).decode()
    table[1].tc_offset_div2 = 0
    assert lib.hevcrecon_syntax_set_slices(syn.h, n, C.addressof(table)) == 0  # a refused table does not spoil the picture
    # the table may come after the CTUs: packed either way, the picture is the same
    early = engine.detached_syntax(seq.seq_params).pack(pic)
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    for a in range(pic.nctu):
        assert lib.hevcrecon_syntax_add_ctu(syn.h, a, C.byref(ctus[a])) == 0
    assert lib.hevcrecon_syntax_set_slices(syn.h, n, C.addressof(table)) == 0
    assert lib.hevcrecon_syntax_finish(syn.h, C.byref(pic.pic_params)) == 0
    assert syn.stats() == early.stats()
    assert lib.hevcrecon_syntax_set_slices(syn.h, n, C.addressof(table)) != 0  # too late: the picture is closed
    early.close()
    # a table without the second slice: refused when the picture closes
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    short = (abi.SliceParams * 1).from_buffer_copy(bytes(table[0]))
    assert lib.hevcrecon_syntax_set_slices(syn.h, 1, C.addressof(short)) == 0
    for a in range(pic.nctu):
        assert lib.hevcrecon_syntax_add_ctu(syn.h, a, C.byref(ctus[a])) == 0
    assert lib.hevcrecon_syntax_finish(syn.h, C.byref(pic.pic_params)) != 0
    assert "slice table" in lib.hevcrecon_syntax_last_error(syn.h).decode()
    # a slice that starts inside a CTU
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_set_slices(syn.h, n, C.addressof(table)) == 0
    starts = pic.slice_start[0].copy()
    starts[128:] = table[1].start_addr
    broken = abi.CtuSyntax.from_buffer_copy(bytes(ctus[0]))
    broken.slice_start = starts.ctypes.data
    assert lib.hevcrecon_syntax_add_ctu(syn.h, 0, C.byref(broken)) == 0
    for a in range(1, pic.nctu):
        assert lib.hevcrecon_syntax_add_ctu(syn.h, a, C.byref(ctus[a])) == 0
    assert lib.hevcrecon_syntax_finish(syn.h, C.byref(pic.pic_params)) != 0
    assert "CTU boundaries" in lib.hevcrecon_syntax_last_error(syn.h).decode()
    # n = 0 returns to one set of parameters
    assert lib.hevcrecon_syntax_begin(syn.h) == 0
    assert lib.hevcrecon_syntax_set_slices(syn.h, 0, None) == 0
    syn.close()


@pytest.mark.parametrize("stream", ["randomaccess_main", "intra_main", "slices_tiles_nolf", "wpb", "scalinglist", "pcm"])
def test_packed_picture_survives_export_and_import(stream):
    """hevcrecon_syntax_export / _import: a finished picture as one block of bytes, read back into another (host-only)
    object: same statistics, same algorithmic bytes, and the same block when exported again.  Damaged blocks are
    refused with a message."""
    seq = load_golden(stream + "_416x240", 2)
    lib = abi.load_library()
    for pic in seq.pictures:
        a = engine.detached_syntax(seq.seq_params).pack(pic)
        n = C.c_int64(0)
        assert lib.hevcrecon_syntax_export(a.h, None, 0, C.byref(n)) == 0 and n.value > 0
        blob = (C.c_uint8 * n.value)()
        assert lib.hevcrecon_syntax_export(a.h, blob, n.value - 1, C.byref(n)) != 0  # too little room
        assert lib.hevcrecon_syntax_export(a.h, blob, n.value, C.byref(n)) == 0
        b = engine.detached_syntax(seq.seq_params)
        assert lib.hevcrecon_syntax_import(b.h, blob, n.value) == 0, lib.hevcrecon_syntax_last_error(b.h)
        sa, sb = a.stats(), b.stats()
        sa[12] = sb[12] = 0  # a packing statistic that does not travel (dependency levels per CTU, summed)
        assert sb == sa and b.algorithmic_bytes() == a.algorithmic_bytes()
        again = (C.c_uint8 * n.value)()
        m = C.c_int64(0)
        assert lib.hevcrecon_syntax_export(b.h, again, n.value, C.byref(m)) == 0 and m.value == n.value
        assert bytes(again) == bytes(blob)
        # damage: wrong length, a count that does not add up, an index beyond its array
        assert lib.hevcrecon_syntax_import(b.h, blob, n.value - 16) != 0
        bad = (C.c_uint8 * n.value).from_buffer_copy(blob)
        bad[0] ^= 0xFF
        assert lib.hevcrecon_syntax_import(b.h, bad, n.value) != 0
        assert "packed picture" in lib.hevcrecon_syntax_last_error(b.h).decode()
        other = abi.SeqParams.from_buffer_copy(seq.seq_params)
        other.width += 64
        c = engine.detached_syntax(other)
        assert lib.hevcrecon_syntax_import(c.h, blob, n.value) != 0
        assert "sequence parameters" in lib.hevcrecon_syntax_last_error(c.h).decode()
        for y in (a, b, c):
            y.close()
