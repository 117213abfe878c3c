import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# "scalinglist": encoder_randomaccess_main.cfg with --ScalingList=1 (default scaling lists)
# "lossless": encoder_lowdelay_P_main.cfg, 6 frames, --TransquantBypassEnableFlag=1 --CUTransquantBypassFlagValue=1
STREAMS_240 = ["lowdelay_P_main", "randomaccess_main", "randomaccess_he10", "intra_main", "scalinglist", "lossless"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU machine)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_cache = {}


def load_golden(name, max_pictures=None):
    """Read a golden syntax dump (cached per session)."""
    from hevc_amd import syntax

    key = (name, max_pictures)
    if key not in _cache:
        _cache[key] = syntax.read_dump(os.path.join(GOLDEN, name + ".syn.xz"), max_pictures)
    return _cache[key]


def bit_depths(seq):
    sp = seq.seq_params
    return [sp.bit_depth_luma, sp.bit_depth_chroma, sp.bit_depth_chroma]
