"""The oracle (oracle/recon_ref.c) against the reference decoder's own outputs.

Golden data (tests/golden):
  *.syn.xz     syntax dumps written by the bridge decoder (oracle/hm_bridge.cpp, dump mode) while the
               reference decoder (HM 9.1) decoded streams made by the reference encoder from the synthetic
               clip of tools/gen_clip.py; each picture carries MD5s of the reference's picture before
               deblocking, after deblocking and final.
  hm_md5.json  the per-picture MD5s the unmodified reference decoder printed for the same streams, each
               verified "(OK)" by it against the encoder's SEI digest.
Generator: tools/make_fixtures.sh.
"""
import json
import os

import pytest

from conftest import GOLDEN, NUM_PICTURES_240, STREAMS_240, bit_depths, load_golden
from hevc_amd import syntax
from oracle.refmodel import RefModel


@pytest.mark.parametrize("stream", STREAMS_240)
def test_oracle_matches_reference_all_stages(stream):
    seq = load_golden(stream + "_416x240")
    bd = bit_depths(seq)
    ref = RefModel(seq.seq_params)
    hm = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))[stream + "_240"]
    assert len(hm) == len(seq.pictures) == NUM_PICTURES_240.get(stream, 9)
    for pic, (poc, my, mu, mv) in zip(seq.pictures, hm):
        assert pic.poc == poc
        ref.reconstruct(pic)
        for stage in range(3):
            planes = ref.download(stage, pic.pic_id)
            for c in range(3):
                assert syntax.plane_md5(planes[c], bd[c]) == pic.stage_md5[stage][c], (stream, poc, stage, c)
        # the dump's final digest is the one the reference decoder printed and checked against the SEI
        assert pic.stage_md5[2] == [my, mu, mv]
    ref.close()


def test_oracle_known_answers():
    """Self-evident cases that need no reference: a lone DC coefficient, transform skip."""
    import numpy as np

    from hevc_amd import abi

    seq = load_golden("intra_main_416x240", 1)
    ref = RefModel(seq.seq_params)
    pp = abi.PicParams()
    for log2n in (2, 3, 4, 5):
        n = 1 << log2n
        coef = np.zeros((n, n), np.int32)
        coef[0, 0] = 64
        out = ref.inverse_tu(pp, coef, log2n, 0, 32, 0, 0)
        assert (out == out[0, 0]).all() and out[0, 0] > 0  # flat block
    coef = np.arange(16, dtype=np.int32).reshape(4, 4)
    out = ref.inverse_tu(pp, coef, 2, 0, 28, 0, 1)  # QP 28: scale 64 << 4, shift 1 -> dq = coef * 512; tskip >> 5
    assert (out == coef * 16).all()
    ref.close()


STREAMS_1080 = [("randomaccess_main", 60), ("randomaccess_he10", 60), ("intra_main", 16)]


@pytest.mark.parametrize("stream,npics", STREAMS_1080)
def test_oracle_matches_reference_1080p(stream, npics):
    """Full-size 1080p streams (all-intra, random access, 10-bit random access): every picture, every stage, against the reference decoder."""
    seq = load_golden(stream + "_1920x1080")
    bd = bit_depths(seq)
    hm = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))[stream + "_1080"]
    assert len(seq.pictures) == len(hm) == npics
    ref = RefModel(seq.seq_params)
    for pic, (poc, my, mu, mv) in zip(seq.pictures, hm):
        assert pic.poc == poc and pic.stage_md5[2] == [my, mu, mv]
        ref.reconstruct(pic)
        for stage in range(3):
            planes = ref.download(stage, pic.pic_id)
            for c in range(3):
                assert syntax.plane_md5(planes[c], bd[c]) == pic.stage_md5[stage][c], (stream, poc, stage, c)
        pic.drop_ctypes()
    ref.close()


@pytest.mark.parametrize("seed", range(1, 8))
def test_oracle_matches_reference_on_the_other_seeds(seed):
    """The seven further 1080p random-access streams of the multi-stream benchmark (clips of seeds 20240607 + k): the
    dump's final digests are the ones the reference decoder printed, for all 60 pictures, and the oracle reproduces
    the first pictures (an I and a B picture) at every stage; the GPU suite checks all 60."""
    seq = load_golden("randomaccess_main_1920x1080_s%d" % seed)
    bd = bit_depths(seq)
    hm = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))["randomaccess_main_1080_s%d" % seed]
    assert len(seq.pictures) == len(hm) == 60
    for pic, (poc, my, mu, mv) in zip(seq.pictures, hm):
        assert pic.poc == poc and pic.stage_md5[2] == [my, mu, mv]
    ref = RefModel(seq.seq_params)
    for pic in seq.pictures[:2]:
        ref.reconstruct(pic)
        for stage in range(3):
            planes = ref.download(stage, pic.pic_id)
            assert [syntax.plane_md5(planes[c], bd[c]) for c in range(3)] == pic.stage_md5[stage], (seed, pic.poc, stage)
        pic.drop_ctypes()
    ref.close()


def test_scaling_lists_are_what_makes_the_scalinglist_stream_match():
    """Negative control for the scaling-list fixture: de-quantised flat, its first picture must NOT reproduce the
    reference's digest (so the stream really exercises the scaling-list branch of the de-quantiser)."""
    import copy

    seq = load_golden("scalinglist_416x240", 1)
    pic = copy.copy(seq.pictures[0])
    assert pic.scaling is not None
    pic.scaling = None
    ref = RefModel(seq.seq_params)
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


@pytest.mark.parametrize("stream", ["wpp", "wpb"])
def test_weights_are_what_makes_the_weighted_streams_match(stream):
    """Negative control for the weighted-prediction fixtures: predicted without the weights, the first inter picture
    must NOT reproduce the reference's digest."""
    import copy

    seq = load_golden(stream + "_416x240", 2)
    ref = RefModel(seq.seq_params)
    ref.reconstruct(seq.pictures[0])
    pic = copy.copy(seq.pictures[1])
    assert pic.pred_weights is not None
    pic.pred_weights = None
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


def test_pcm_samples_are_what_makes_the_pcm_stream_match():
    """Negative control for the IPCM fixture: with the PCM sample arrays zeroed its first picture must NOT match."""
    import copy

    seq = load_golden("pcm_416x240", 1)
    pic = copy.copy(seq.pictures[0])
    assert pic.pcm
    pic.pcm = {a: tuple(p * 0 for p in planes) for a, planes in pic.pcm.items()}
    pic._keep = None
    ref = RefModel(seq.seq_params)
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


def test_constraint_is_what_makes_the_cip_stream_match():
    """Negative control for the constrained-intra fixture: with the flag cleared, its first B picture must NOT match."""
    import copy

    seq = load_golden("cip_416x240", 2)
    ref = RefModel(seq.seq_params)
    ref.reconstruct(seq.pictures[0])
    pic = copy.copy(seq.pictures[1])
    from hevc_amd import abi

    pp = abi.PicParams.from_buffer_copy(bytes(pic.pic_params))
    assert pp.constrained_intra_pred == 1
    pp.constrained_intra_pred = 0
    pic.pic_params = pp
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


def test_slice_starts_are_what_makes_the_slices_stream_match():
    """Negative control for the multi-slice fixture: treated as one slice, its first picture must NOT match."""
    import copy

    seq = load_golden("slices_416x240", 1)
    pic = copy.copy(seq.pictures[0])
    assert pic.slice_start is not None and len(set(pic.slice_start.ravel().tolist())) == 6
    pic.slice_start = None
    pic._keep = None
    ref = RefModel(seq.seq_params)
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


def test_tile_ids_are_what_makes_the_tiles_stream_match():
    """Negative control for the tiles fixture: with every CTU in one tile (coding order kept), its first picture
    must NOT match."""
    import copy

    seq = load_golden("tiles_416x240", 1)
    pic = copy.copy(seq.pictures[0])
    assert pic.tiles is not None and len(set(pic.tiles[:, 1].tolist())) == 6
    pic.tiles = pic.tiles.copy()
    pic.tiles[:, 1] = 0
    pic._keep = None
    ref = RefModel(seq.seq_params)
    ref.reconstruct(pic)
    luma = ref.download(2, pic.pic_id)[0]
    assert syntax.plane_md5(luma, seq.seq_params.bit_depth_luma) != pic.stage_md5[2][0]


@pytest.mark.parametrize("stream", ["slices_nolf", "tiles_nolf", "slices_tiles_nolf"])
def test_uncrossed_boundaries_are_what_makes_these_streams_match(stream):
    """Negative control for the streams whose in-loop filters stop at slice / tile boundaries: filtered across them
    (no slice table, tile flag cleared), the first picture matches before deblocking and must NOT match after."""
    import copy

    seq = load_golden(stream + "_416x240", 1)
    pic = copy.copy(seq.pictures[0])
    assert (pic.slices is not None and any(not s.loop_filter_across_slices for s in pic.slices)) or pic.pic_params.no_loop_filter_across_tiles
    pic.slices = None
    pic.pic_params = type(pic.pic_params).from_buffer_copy(pic.pic_params)
    pic.pic_params.no_loop_filter_across_tiles = 0
    ref = RefModel(seq.seq_params)
    ref.reconstruct(pic)
    bd = seq.seq_params.bit_depth_luma
    assert syntax.plane_md5(ref.download(0, pic.pic_id)[0], bd) == pic.stage_md5[0][0]
    assert syntax.plane_md5(ref.download(1, pic.pic_id)[0], bd) != pic.stage_md5[1][0]
    assert syntax.plane_md5(ref.download(2, pic.pic_id)[0], bd) != pic.stage_md5[2][0]
    ref.close()
