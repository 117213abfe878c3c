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


@pytest.mark.parametrize("stream,npics", [("randomaccess_he10", 3), ("lowdelay_P_main", 2), ("scalinglist", 2), ("pcm", 1), ("pcmlf", 2), ("wpp", 3), ("wpb", 3), ("cip", 2), ("bd12", 3), ("slices", 2), ("tiles", 2)])
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
