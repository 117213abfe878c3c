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
