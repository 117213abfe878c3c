import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# "scalinglist": encoder_randomaccess_main.cfg with --ScalingList=1 (default scaling lists)
# "lossless": encoder_lowdelay_P_main.cfg, 6 frames, --TransquantBypassEnableFlag=1 --CUTransquantBypassFlagValue=1
# "pcm": encoder_randomaccess_he10.cfg, 5 frames of the clip with a white-noise patch, QP 6, --PCMEnabledFlag=1
#        --PCMFilterDisableFlag=1 (10-bit pictures, 8-bit PCM samples, IPCM units exempt from the in-loop filters)
# "pcmlf": encoder_lowdelay_P_main.cfg, 4 frames of the same clip, QP 6, --PCMEnabledFlag=1 --PCMFilterDisableFlag=0
# "cip": encoder_randomaccess_main.cfg, 6 frames of the clip with the noise patch, QP 30, --ConstrainedIntraPred=1
# "bd12": encoder_randomaccess_he10.cfg, 5 frames, --InternalBitDepth=12
# "slices": encoder_randomaccess_main.cfg, 5 frames of the noise-patch clip, QP 30, --SliceMode=1 --SliceArgument=5
# "tiles": the same clip and QP, --UniformSpacingIdc=1 --NumTileColumnsMinus1=2 --NumTileRowsMinus1=1 (3 x 2 tiles)
# "slices_nolf": as "slices" with --LFCrossSliceBoundaryFlag=0; "tiles_nolf": as "tiles" with --LFCrossTileBoundaryFlag=0
#        (deblocking and SAO stop at slice / tile boundaries)
# "slices_tiles_nolf": encoder_randomaccess_main.cfg, 9 frames of the plain clip, QP 36, 2 x 2 tiles, slices of 3 CTUs, neither
#        crossed by the in-loop filters
# "wpp": encoder_lowdelay_P_main.cfg on a clip that fades to black, -wpP 1 (explicit weights in P slices, 8-bit)
# "wpb": encoder_randomaccess_he10.cfg on the same clip, -wpP 1 -wpB 1 (explicit weights in B slices, 10-bit)
STREAMS_240 = ["lowdelay_P_main", "randomaccess_main", "randomaccess_he10", "intra_main", "scalinglist", "lossless", "pcm", "pcmlf",
               "wpp", "wpb", "cip", "bd12", "slices", "tiles", "slices_nolf", "tiles_nolf", "slices_tiles_nolf"]
NUM_PICTURES_240 = {"lossless": 6, "pcm": 5, "pcmlf": 4, "cip": 6, "bd12": 5, "slices": 5, "tiles": 5, "slices_nolf": 5, "tiles_nolf": 5}  # every other stream has 9


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
