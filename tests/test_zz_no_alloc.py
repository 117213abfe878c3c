"""hevcrecon_reconstruct_noalloc, the entry point that allocates nothing (job descriptor, working picture and residual
planes in the library's fixed device array), against hevcrecon_reconstruct on the GPU, through the C ABI: two calls in
a row on the one work area of the device, the second with a smaller picture, then the size limit."""
import numpy as np
import pytest

from hevc_amd import abi
from test_emulated_kernels import _known_answer_dc_picture, _noalloc_call, _noalloc_matches_regular

pytestmark = pytest.mark.gpu


def test_noalloc_matches_regular_entry_point_twice_in_a_row():
    # a 1080p I picture (62 CTU waves), then a 416x240 B picture with intra blocks (much smaller)
    _noalloc_matches_regular(abi.load_library(), ("randomaccess_main_1920x1080", 0), ("cip_416x240", 2))


def test_noalloc_refuses_what_does_not_fit():
    """A 4096x2304 picture needs 42 MB of working picture and residual planes: the regular entry point takes it, the
    allocation-free one refuses it with a message and leaves the destination alone."""
    lib = abi.load_library()
    eng, syn = _known_answer_dc_picture(lib, 4096, 2304, 8)
    planes = [np.full((2304 >> (c > 0), 4096 >> (c > 0)), 7, np.int16) for c in range(3)]
    eng.upload(9, 0, planes)
    assert _noalloc_call(lib, eng, syn, 9) != 0
    assert "too large" in lib.hevcrecon_last_error(eng.ctx).decode()
    for plane in eng.download(9):
        assert (plane == 7).all()
    syn.close()
    eng.close()
