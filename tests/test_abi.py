"""The C-ABI library: loads, exports every symbol include/hevcrecon.h declares, and refuses to work without a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from hevc_amd import abi


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hevcrecon.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hevcrecon_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export " + n
        assert n in abi.EXPORTS, "abi.py does not type " + n
    assert sorted(abi.EXPORTS) == names


def test_struct_sizes_match_header():
    assert C.sizeof(abi.SeqParams) == 16 * 4
    assert C.sizeof(abi.PicParams) == (13 + 2 * 16 * 2 + 8) * 4
    assert C.sizeof(abi.CtuSyntax) == 26 * 8 + (3 + 3 + 12 + 2) * 4


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError):
        abi.load_library(str(tmp_path / "nope.so"))


def test_kernel_names():
    lib = abi.load_library()
    names = [lib.hevcrecon_kernel_name(i).decode() for i in range(abi.NUM_KERNELS)]
    assert names[:6] == ["k_inter_tile64", "k_transform", "k_intra", "k_deblock_v", "k_deblock_h", "k_sao"]
    assert names[8:11] == ["k_inter_tile32", "k_inter_tile16", "k_inter_blk"]
