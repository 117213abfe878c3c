"""Kernel LOGIC under CPU emulation (tests/emu): the engine's own sources compiled against a HIP look-alike
and run on CPU threads, compared with the oracle.  This is a development aid for a machine without a GPU;
it says nothing about the gfx950 build, which the -m gpu tests cover on real hardware."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from hevc_amd import abi, engine
from oracle.refmodel import RefModel

EMU = os.path.join(ROOT, "tests", "emu", "libhevcrecon_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return abi.load_library(EMU)


@pytest.mark.parametrize("stream,npics", [("randomaccess_he10", 3), ("lowdelay_P_main", 2), ("scalinglist", 2), ("pcm", 1), ("pcmlf", 2), ("wpp", 3), ("wpb", 3), ("cip", 2), ("bd12", 3), ("slices", 2), ("tiles", 2),
                                          ("slices_nolf", 2), ("tiles_nolf", 2), ("slices_tiles_nolf", 2)])
def test_emulated_engine_matches_oracle(emu_lib, stream, npics):
    seq = load_golden(stream + "_416x240", npics)
    eng = engine.Engine(seq.seq_params, lib=emu_lib)
    eng.keep_stages(True)
    ref = RefModel(seq.seq_params)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        ref.reconstruct(pic)
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        for stage in range(3):
            got = eng.download(pic.pic_id) if stage == 2 else eng.download_stage(stage)
            want = ref.download(stage, pic.pic_id)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), (stream, pic.poc, stage, c)
    syn.close()
    eng.close()
    ref.close()


@pytest.mark.parametrize("stream,depth", [("randomaccess_he10", 8), ("lowdelay_P_main", 10)])
def test_emulated_output_file_format(emu_lib, stream, depth):
    """hevcrecon_picture_download_file against the bytes the reference decoder wrote with -d <depth> (digest per frame
    of its output file, display order): scaling down with rounding and clipping, and scaling up."""
    import hashlib
    import json

    from conftest import GOLDEN

    want = json.load(open(os.path.join(GOLDEN, "output_md5.json")))["%s_240_d%d" % (stream, depth)]
    seq = load_golden(stream + "_416x240", 3)
    eng = engine.Engine(seq.seq_params, lib=emu_lib)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        data = eng.download_file(pic.pic_id, (depth, depth))
        assert hashlib.md5(data.tobytes()).hexdigest() == want[pic.poc], pic.poc
    syn.close()
    eng.close()


@pytest.mark.parametrize("stream", ["randomaccess_he10"])  # 8-bit planes: the padding test below and the GPU suite
def test_emulated_device_md5(emu_lib, stream):
    """hevcrecon_picture_md5_batch: digests of all resident pictures in one call against the digests of the reference
    decoder's own pictures (8-bit: one byte per sample; 10-bit: two)."""
    seq = load_golden(stream + "_416x240", 3)
    eng = engine.Engine(seq.seq_params, lib=emu_lib)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
    ids = [p.pic_id for p in seq.pictures]
    assert eng.md5(ids) == [p.stage_md5[2] for p in seq.pictures]
    assert eng.md5(ids[-1:]) == [seq.pictures[-1].stage_md5[2]]
    with pytest.raises(engine.EngineError):
        eng.md5([999])
    # the same through tickets (hevcrecon_picture_md5_submit / _poll / _wait); a ticket is spent when collected
    t1, t2 = eng.md5_submit(ids), eng.md5_submit(ids[:1])
    assert eng.md5_wait(t2) == [seq.pictures[0].stage_md5[2]]
    assert eng.md5_poll(t1) == [p.stage_md5[2] for p in seq.pictures]
    with pytest.raises(engine.EngineError):
        eng.md5_poll(t1)
    # more tickets outstanding than the engine keeps track of are refused, and collecting makes room again
    held = [eng.md5_submit(ids[:1]) for _ in range(64)]
    with pytest.raises(engine.EngineError):
        eng.md5_submit(ids[:1])
    for t in held:
        assert eng.md5_wait(t) == [seq.pictures[0].stage_md5[2]]
    assert eng.md5_wait(eng.md5_submit(ids[:1])) == [seq.pictures[0].stage_md5[2]]
    syn.close()
    eng.close()


def _md5_of_uploaded_pictures(lib, width, height, bit_depth):
    """Random pictures uploaded, hashed on the device and with hashlib: sizes whose planes do not end on a 64-byte
    block boundary (the golden streams' planes all do)."""
    from hevc_amd import syntax

    sp = abi.SeqParams(width=width, height=height, bit_depth_luma=bit_depth, bit_depth_chroma=bit_depth, ctu_size=64, max_cu_depth=4)
    eng = engine.Engine(sp, lib=lib)
    rng = np.random.default_rng(width * 31 + height + bit_depth)
    want = []
    for i in range(3):
        planes = [rng.integers(0, 1 << bit_depth, size=(h, w)).astype(np.int16) for (w, h) in ((width, height), (width // 2, height // 2), (width // 2, height // 2))]
        eng.upload(i, i, planes)
        want.append([syntax.plane_md5(p, bit_depth) for p in planes])
    assert eng.md5([0, 1, 2]) == want
    assert eng.md5([2, 2, 0]) == [want[2], want[2], want[0]]
    eng.close()


@pytest.mark.parametrize("width,height,bit_depth", [(8, 8, 8), (72, 40, 8), (424, 248, 8), (72, 40, 10), (8, 8, 12)])
def test_emulated_device_md5_padding(emu_lib, width, height, bit_depth):
    _md5_of_uploaded_pictures(emu_lib, width, height, bit_depth)


def _ring_sizes_give_the_same_pictures(lib, ring_bytes):
    """The upload ring in its corner cases: none at all, one that wraps around and overflows (pictures that find no
    room stage in buffers of their own), and the default; three syntax objects in rotation, pictures reconstructed
    two at a time where the stream allows it."""
    seq = load_golden("randomaccess_main_416x240")
    eng = engine.Engine(seq.seq_params, lib=lib)
    if ring_bytes is not None:
        eng.set_upload_ring(ring_bytes)
    syns = [eng.new_syntax() for _ in range(3)]
    for i, pic in enumerate(seq.pictures):
        syn = syns[i % 3]
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        assert eng.md5([pic.pic_id]) == [pic.stage_md5[2]], (ring_bytes, pic.poc)
    with pytest.raises(engine.EngineError):
        eng.set_upload_ring(1 << 20) if ring_bytes != 0 else eng.md5([999])
    for s in syns:
        s.close()
    eng.close()


@pytest.mark.parametrize("ring_bytes", [48 << 10])  # 48 KB: overflow, then wrap-around at the last picture (more sizes in the GPU suite)
def test_emulated_upload_ring(emu_lib, ring_bytes):
    _ring_sizes_give_the_same_pictures(emu_lib, ring_bytes)


def _clone_with_id_offset(pic, off):
    """A copy of a picture's syntax whose own id and reference ids are shifted by `off` (a second stream's DPB)."""
    import copy

    p = copy.copy(pic)
    p._keep = None
    p.pic_id = pic.pic_id + off
    p.pic_params = type(pic.pic_params).from_buffer_copy(pic.pic_params)
    for l in range(2):
        for r in range(p.pic_params.num_ref[l]):
            p.pic_params.ref_pic_id[l][r] = pic.pic_params.ref_pic_id[l][r] + off
    return p


def _heterogeneous_batches_match_their_own_streams(lib, streams, npics, check_stages):
    """Different streams at DIFFERENT picture positions in one hevcrecon_reconstruct_batch call: I next to P next to B
    pictures, SAO on next to SAO off, explicit weights next to none, different numbers of intra CTU waves and list
    sizes.  Stream k starts k batches late, so every batch mixes positions.  Every picture is checked against the
    digests of the reference decoder's own picture (final; with `check_stages` also before / after deblocking for the
    first picture of the batch, the one whose stages the engine keeps)."""
    from hevc_amd import syntax

    seqs = [load_golden(name + "_416x240", npics) for name in streams]
    sp = seqs[0].seq_params
    for q in seqs:
        assert bytes(q.seq_params) == bytes(sp), "streams of one batch share the sequence parameters"
    bd = [sp.bit_depth_luma, sp.bit_depth_chroma, sp.bit_depth_chroma]
    eng = engine.Engine(sp, lib=lib)
    eng.keep_stages(check_stages)
    n = len(seqs)
    syns = [eng.new_syntax() for _ in range(n)]
    steps = max(len(q.pictures) for q in seqs) + n - 1
    kinds = set()
    for b in range(steps):
        members = [(k, b - k) for k in range(n) if 0 <= b - k < len(seqs[k].pictures)]
        # rotate so that a different stream comes first (and has its stages kept) in every batch
        members = members[b % len(members):] + members[:b % len(members)]
        pics = [_clone_with_id_offset(seqs[k].pictures[t], 64 * k) for k, t in members]
        for (k, t), p in zip(members, pics):
            syns[k].pack(p)
        kinds.add(tuple(sorted({int(p.pic_params.slice_type) for p in pics})))
        eng.reconstruct_batch([syns[k] for k, _ in members], [p.pic_id for p in pics])
        for i, ((k, t), p) in enumerate(zip(members, pics)):
            final = eng.download(p.pic_id)
            assert [syntax.plane_md5(final[c], bd[c]) for c in range(3)] == p.stage_md5[2], (streams[k], p.poc, "final")
            if check_stages and i == 0:
                for stage in range(2):
                    got = eng.download_stage(stage)
                    assert [syntax.plane_md5(got[c], bd[c]) for c in range(3)] == p.stage_md5[stage], (streams[k], p.poc, stage)
    assert any(len(kk) > 1 for kk in kinds), "no batch mixed slice types"
    for y in syns:
        y.close()
    eng.close()


def _batch_of_intra_pictures_matches_reference(lib, stream, npics):
    """The pictures of an all-intra stream read no other picture, so all of them can go into ONE
    hevcrecon_reconstruct_batch call: a batch without a single inter block (the intra kernel's instantiation that
    stages CTU borders only; with eight pictures or more also the launch with more workgroups).  Every picture is
    checked against the digests of the reference decoder's own picture."""
    from hevc_amd import syntax

    seq = load_golden(stream, npics)
    assert len(seq.pictures) == npics and all(int(p.pic_params.slice_type) == 2 for p in seq.pictures)
    npics = len(seq.pictures)
    sp = seq.seq_params
    bd = [sp.bit_depth_luma, sp.bit_depth_chroma, sp.bit_depth_chroma]
    eng = engine.Engine(sp, lib=lib)
    syns = [eng.new_syntax() for _ in range(npics)]
    pics = [_clone_with_id_offset(p, 64 * i) for i, p in enumerate(seq.pictures)]  # the decoder recycles its picture ids
    for y, p in zip(syns, pics):
        y.pack(p)
    eng.reconstruct_batch(syns, [p.pic_id for p in pics])
    for p in pics:
        final = eng.download(p.pic_id)
        assert [syntax.plane_md5(final[c], bd[c]) for c in range(3)] == p.stage_md5[2], (stream, p.poc)
    for y in syns:
        y.close()
    eng.close()


def test_emulated_batch_of_intra_pictures(emu_lib):
    _batch_of_intra_pictures_matches_reference(emu_lib, "intra_main_416x240", 4)  # (the GPU suite batches all of them)


def test_emulated_heterogeneous_batch(emu_lib):
    _heterogeneous_batches_match_their_own_streams(emu_lib, ["lowdelay_P_main", "wpp", "intra_main", "slices"], 3, True)


def _picture_copy_gives_the_same_picture(lib):
    seq = load_golden("lowdelay_P_main_416x240", 2)
    eng = engine.Engine(seq.seq_params, lib=lib)
    syn = eng.new_syntax()
    pic = seq.pictures[0]
    syn.pack(pic)
    eng.reconstruct(syn, pic.pic_id)
    eng.copy_picture(40, pic.pic_id, 77)
    a, b = eng.download(pic.pic_id), eng.download(40)
    for c in range(3):
        assert np.array_equal(a[c], b[c])
    # the copy serves as a reference like any other picture: the P picture predicted from it equals the original
    p1 = _clone_with_id_offset(seq.pictures[1], 0)
    for l in range(2):
        for r in range(p1.pic_params.num_ref[l]):
            p1.pic_params.ref_pic_id[l][r] = 40
    syn.pack(p1)
    eng.reconstruct(syn, 41)
    assert eng.md5([41]) == [seq.pictures[1].stage_md5[2]]
    with pytest.raises(engine.EngineError):
        eng.copy_picture(42, 999, 0)
    syn.close()
    eng.close()


def test_emulated_picture_copy(emu_lib):
    _picture_copy_gives_the_same_picture(emu_lib)


def _noalloc_call(lib, eng, syn, dst):
    return lib.hevcrecon_reconstruct_noalloc(eng.ctx, syn.h, dst)


def _noalloc_matches_regular(lib, first, second, turns=3):
    """hevcrecon_reconstruct_noalloc against hevcrecon_reconstruct, two calls in a row on the device's one fixed work
    area, the second with a smaller input: (fixture, index of the picture to reconstruct both ways) each; the
    pictures before it are reconstructed the regular way so that its references are resident.  Then the same
    picture from a freshly packed syntax object whose first reconstruction is the allocation-free one (its hand-over
    and the expansion of its motion records run inside that call), and two contexts taking turns on the work area
    without the host waiting in between."""
    counter = getattr(lib, "hevcrecon_emu_allocations", None)  # the CPU emulation counts what the engine allocates
    if counter is not None:
        counter.restype = __import__("ctypes").c_long
    for name, idx in (first, second):
        seq = load_golden(name, idx + 1)
        eng = engine.Engine(seq.seq_params, lib=lib)
        syn = eng.new_syntax()
        for pic in seq.pictures[:idx]:
            syn.pack(pic)
            eng.reconstruct(syn, pic.pic_id)
        pic = seq.pictures[idx]
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        want = eng.download(pic.pic_id)
        assert eng.md5([pic.pic_id]) == [pic.stage_md5[2]]
        dst = 50
        assert _noalloc_call(lib, eng, syn, dst) != 0, "a destination that does not exist yet must be refused"
        eng.upload(dst, 0, [np.full_like(p, 3) for p in want])  # creates the destination picture
        assert _noalloc_call(lib, eng, syn, dst) == 0, lib.hevcrecon_last_error(eng.ctx)
        got = eng.download(dst)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (name, pic.poc, c)
        assert eng.md5([dst]) == [pic.stage_md5[2]]
        # a fresh syntax object: nothing of it is on the device yet when the allocation-free call comes
        fresh = eng.new_syntax().pack(pic)
        eng.upload(dst + 1, 0, [np.full_like(p, 5) for p in want])
        before = counter() if counter else 0
        assert _noalloc_call(lib, eng, fresh, dst + 1) == 0, lib.hevcrecon_last_error(eng.ctx)
        if counter:
            assert counter() == before, "hevcrecon_reconstruct_noalloc allocated something"
        got = eng.download(dst + 1)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (name, pic.poc, c, "fresh syntax")
        # two contexts, one work area: calls queued alternately, nobody waits on the host in between
        eng2 = engine.Engine(seq.seq_params, lib=lib)
        syn2 = eng2.new_syntax()
        for p in seq.pictures[:idx + 1]:
            syn2.pack(p)
            eng2.reconstruct(syn2, p.pic_id)
        eng2.upload(dst, 0, [np.full_like(p, 9) for p in want])
        for rep in range(turns):
            assert _noalloc_call(lib, eng, syn, dst) == 0, lib.hevcrecon_last_error(eng.ctx)
            assert _noalloc_call(lib, eng2, syn2, dst) == 0, lib.hevcrecon_last_error(eng2.ctx)
        for e in (eng, eng2):
            got = e.download(dst)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), (name, pic.poc, c, "two contexts")
        syn2.close()
        eng2.close()
        fresh.close()
        syn.close()
        eng.close()


def _synthetic_dc_picture(sp):
    """A picture made up here, not decoded: every CTU split into 32x32 intra coding units predicted with the DC mode,
    no residual, SAO off.  With no neighbours available anywhere before the first block, every sample comes out as
    1 << (bit depth - 1): a known answer that needs no reference decoder."""
    from hevc_amd import syntax

    P = abi.PARTS_PER_CTU
    nctu = ((sp.width + 63) // 64) * ((sp.height + 63) // 64)
    pic = syntax.PictureSyntax()
    pic.nctu, pic.pic_id = nctu, 0
    z = lambda dt, per=1: np.zeros((nctu, P * per), dt)
    pic.a = {"depth": z(np.uint8) + 1, "pred_mode": z(np.int8) + 1, "part_size": z(np.int8), "tr_idx": z(np.uint8),
             "cbf0": z(np.uint8), "cbf1": z(np.uint8), "cbf2": z(np.uint8), "qp": z(np.int8) + 30,
             "luma_dir": z(np.uint8) + 1, "chroma_dir": z(np.uint8) + 1,
             "tskip0": z(np.uint8), "tskip1": z(np.uint8), "tskip2": z(np.uint8),
             "mv0": z(np.int16, 2), "mv1": z(np.int16, 2), "ref_idx0": z(np.int8) - 1, "ref_idx1": z(np.int8) - 1,
             "tq_bypass": z(np.uint8), "ipcm": z(np.uint8),
             "coeff0": z(np.int16, 16), "coeff1": z(np.int16, 4), "coeff2": z(np.int16, 4)}
    pic.sao = np.zeros((nctu, 3, 6), np.int32)
    pic.sao[:, :, 0] = -1
    pic.pic_params.slice_type = 2
    pic.pic_params.deblocking_disable = 0
    return pic


def _known_answer_dc_picture(lib, width, height, bit_depth):
    sp = abi.SeqParams(width=width, height=height, bit_depth_luma=bit_depth, bit_depth_chroma=bit_depth, ctu_size=64, max_cu_depth=4)
    eng = engine.Engine(sp, lib=lib)
    syn = eng.new_syntax().pack(_synthetic_dc_picture(sp))
    eng.reconstruct(syn, 0)
    for plane in eng.download(0):
        assert (plane == 1 << (bit_depth - 1)).all()
    return eng, syn


def test_emulated_known_answer_dc_picture(emu_lib):
    eng, syn = _known_answer_dc_picture(emu_lib, 136, 72, 10)  # sizes that are not multiples of the CTU
    syn.close()
    eng.close()


def test_emulated_noalloc_entry_point(emu_lib):
    """hevcrecon_reconstruct_noalloc on the emulator: two calls in a row on the one fixed work area (an I picture, then
    a B picture with fewer blocks), each against hevcrecon_reconstruct."""
    _noalloc_matches_regular(emu_lib, ("cip_416x240", 0), ("cip_416x240", 2), turns=1)  # (a reconstruction takes seconds here)


def _per_slice_parameters_match_oracle(lib, stream="slices_tiles_nolf", npics=2):
    """Slices that differ in everything the slice table carries (deblocking on / off, beta and tc offsets, chroma QP
    deltas, filtering across their boundaries), with an edge-offset SAO class forced on every CTU so that every slice
    and tile boundary is exercised in all four directions: engine against oracle at all three stages.  (No encoder of
    the reference writes such a stream; the two sides are independent restatements of the reference's rules.)"""
    import copy

    seq = load_golden(stream + "_416x240", npics)
    eng = engine.Engine(seq.seq_params, lib=lib)
    eng.keep_stages(True)
    ref = RefModel(seq.seq_params)
    syn = eng.new_syntax()
    for n, src in enumerate(seq.pictures):
        pic = copy.copy(src)
        pic._keep = None
        pic.pic_params = type(src.pic_params).from_buffer_copy(src.pic_params)
        pic.pic_params.sao_luma = pic.pic_params.sao_chroma = 1
        pic.slices = (abi.SliceParams * len(src.slices)).from_buffer_copy(src.slices)
        for i, sl in enumerate(pic.slices):
            sl.deblocking_disable = 1 if i % 4 == 3 else 0
            sl.beta_offset_div2 = (i * 5) % 13 - 6
            sl.tc_offset_div2 = (i * 3) % 13 - 6
            sl.cb_qp_delta_slice = (i * 7) % 25 - 12
            sl.cr_qp_delta_slice = (i * 11) % 25 - 12
            sl.loop_filter_across_slices = (i + n) % 2
        pic.sao = src.sao.copy()
        for a in range(pic.nctu):
            for c in range(3):
                pic.sao[a, c, 0] = (a + c + n) % 5          # all four edge classes and the band class
                pic.sao[a, c, 1] = (3 * a + c) % 29          # band position
                pic.sao[a, c, 2:6] = [3, 1, -1, -3] if pic.sao[a, c, 0] < 4 else [5, -4, 3, -2]
        ref.reconstruct(pic)
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        for stage in range(3):
            got = eng.download(pic.pic_id) if stage == 2 else eng.download_stage(stage)
            want = ref.download(stage, pic.pic_id)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), (pic.poc, stage, c)
    syn.close()
    eng.close()
    ref.close()


def test_emulated_per_slice_parameters(emu_lib):
    _per_slice_parameters_match_oracle(emu_lib)
