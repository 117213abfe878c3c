"""Parity of the HIP engine on a real MI355X, through the C ABI.

Every sample of every picture is compared (bit-exact) with the oracle at three stages (before deblocking,
after deblocking, final), and the final picture's MD5 with the digest the reference decoder printed.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, STREAMS_240, bit_depths, load_golden
from hevc_amd import engine, syntax
from oracle.refmodel import RefModel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stream", STREAMS_240)
def test_all_stages_bit_exact_416x240(stream):
    seq = load_golden(stream + "_416x240")
    bd = bit_depths(seq)
    hm = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))[stream + "_240"]
    eng = engine.Engine(seq.seq_params)
    eng.keep_stages(True)
    ref = RefModel(seq.seq_params)
    syn = eng.new_syntax()
    for pic, (poc, my, mu, mv) in zip(seq.pictures, hm):
        ref.reconstruct(pic)
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        for stage in range(3):
            got = eng.download(pic.pic_id) if stage == 2 else eng.download_stage(stage)
            want = ref.download(stage, pic.pic_id)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), (stream, pic.poc, stage, c)
        final = eng.download(pic.pic_id)
        assert [syntax.plane_md5(final[c], bd[c]) for c in range(3)] == [my, mu, mv], (stream, poc)
    syn.close()
    eng.close()
    ref.close()


def test_batch_of_independent_streams_equals_one_by_one():
    """Two copies of a stream reconstructed as a batch give the same pictures as one stream alone."""
    seq = load_golden("randomaccess_main_416x240")
    bd = bit_depths(seq)
    eng = engine.Engine(seq.seq_params)
    S, base = 3, 100
    syns = [eng.new_syntax() for _ in range(S)]
    for pic in seq.pictures:
        import copy

        for s in range(S):
            p = copy.copy(pic)
            p._keep = None
            p.pic_params = type(pic.pic_params).from_buffer_copy(pic.pic_params)
            for l in range(2):
                for r in range(p.pic_params.num_ref[l]):
                    p.pic_params.ref_pic_id[l][r] = pic.pic_params.ref_pic_id[l][r] + base * s
            syns[s].pack(p)
        eng.reconstruct_batch(syns, [pic.pic_id + base * s for s in range(S)])
        for s in range(S):
            final = eng.download(pic.pic_id + base * s)
            assert [syntax.plane_md5(final[c], bd[c]) for c in range(3)] == pic.stage_md5[2], (pic.poc, s)
    for s in syns:
        s.close()
    eng.close()


def test_errors_are_reported_not_crashed():
    seq = load_golden("randomaccess_main_416x240", 2)
    eng = engine.Engine(seq.seq_params)
    syn = eng.new_syntax()
    syn.pack(seq.pictures[1])  # a B picture whose references were never reconstructed
    with pytest.raises(engine.EngineError):
        eng.reconstruct(syn, seq.pictures[1].pic_id)
    with pytest.raises(engine.EngineError):
        eng.download(999)
    syn.close()
    eng.close()


@pytest.mark.parametrize("stream,depth", [("randomaccess_he10", 8), ("randomaccess_he10", 10), ("lowdelay_P_main", 10), ("bd12", 8)])
def test_output_file_format(stream, depth):
    """hevcrecon_picture_download_file against the bytes the reference decoder wrote with -d <depth> (digest per frame
    of its output file, display order)."""
    import hashlib

    want = json.load(open(os.path.join(GOLDEN, "output_md5.json")))["%s_240_d%d" % (stream, depth)]
    seq = load_golden(stream + "_416x240")
    eng = engine.Engine(seq.seq_params)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        data = eng.download_file(pic.pic_id, (depth, depth))
        assert hashlib.md5(data.tobytes()).hexdigest() == want[pic.poc], pic.poc
    with pytest.raises(engine.EngineError):
        eng.download_file(seq.pictures[0].pic_id, (depth, depth), width=4096)
    syn.close()
    eng.close()


@pytest.mark.parametrize("stream", ["randomaccess_main_416x240", "randomaccess_he10_416x240", "bd12_416x240", "randomaccess_main_1920x1080"])
def test_device_md5(stream):
    """hevcrecon_picture_md5_batch: the picture hash computed on the device, all pictures of the stream in one call,
    against the digests of the reference decoder's own pictures."""
    seq = load_golden(stream)
    eng = engine.Engine(seq.seq_params)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        pic.drop_ctypes()
    last = {p.pic_id: p for p in seq.pictures}  # picture ids are buffer slots: a later picture takes over an earlier one's
    ids = sorted(last)
    assert eng.md5(ids) == [last[i].stage_md5[2] for i in ids]
    assert eng.md5(ids[:1]) == [last[ids[0]].stage_md5[2]]
    with pytest.raises(engine.EngineError):
        eng.md5([999])
    syn.close()
    eng.close()


def test_device_md5_without_blocking():
    """hevcrecon_picture_md5_submit / _poll / _wait: every picture of the 1080p stream is queued for hashing right after
    its reconstruction was, beside the reconstruction of the following pictures (which overwrite picture ids that
    earlier requests still read: the engine orders those), tickets collected later, one by polling."""
    import time

    seq = load_golden("randomaccess_main_1920x1080", 24)
    eng = engine.Engine(seq.seq_params)
    syns = [eng.new_syntax() for _ in range(4)]
    tickets = []
    for k, pic in enumerate(seq.pictures):
        syns[k % 4].pack(pic)
        eng.reconstruct(syns[k % 4], pic.pic_id)
        tickets.append((eng.md5_submit([pic.pic_id]), pic))
        pic.drop_ctypes()
    first, pic0 = tickets[0]
    got = None
    for _ in range(2000):
        got = eng.md5_poll(first)
        if got is not None:
            break
        time.sleep(0.005)
    assert got == [pic0.stage_md5[2]]
    with pytest.raises(engine.EngineError):
        eng.md5_wait(first)  # a ticket is collected once
    for t, pic in tickets[1:]:
        assert eng.md5_wait(t) == [pic.stage_md5[2]], pic.poc
    # a batch in one request, and the blocking call beside it
    ids = sorted({p.pic_id: p for p in seq.pictures})
    last = {p.pic_id: p for p in seq.pictures}
    t = eng.md5_submit(ids)
    assert eng.md5(ids) == [last[i].stage_md5[2] for i in ids]
    assert eng.md5_wait(t) == [last[i].stage_md5[2] for i in ids]
    with pytest.raises(engine.EngineError):
        eng.md5_submit([999])
    for y in syns:
        y.close()
    eng.close()


@pytest.mark.parametrize("bits,npics", [("randomaccess_main_1080", 60), ("randomaccess_he10_1080", 60), ("slices_tiles_nolf_240", 9),
                                        ("pcmlf_240", 4), ("wpb_240", 9), ("intra_main_240", 9)])
def test_drop_in_with_the_device_hash(bits, npics):
    """The decoder CLI with HEVCRECON_HASH=device: CTUs packed while the reference parses, no picture crosses the host
    link, the engine hashes on the GPU beside the reconstruction of later pictures, and the status lines (the
    reference's format, its (OK) verdicts, its exit code) carry the digests of the unmodified reference decoder."""
    from test_bridge_cli import BRIDGE, run_bridge, status_lines, timing_of
    from hevc_amd import abi

    if not os.path.exists(BRIDGE):
        pytest.skip("bridge decoder not built (needs the reference tree at build time)")
    rc, out, err = run_bridge(os.path.join(GOLDEN, bits + ".bin"), abi.LIB_PATH, hash_mode="device")
    assert rc == 0, (out[-1500:], err[-1500:])
    assert out.count("(OK)") == npics and "ERROR" not in out
    assert status_lines(out) == json.load(open(os.path.join(GOLDEN, "hm_md5.json")))[bits]
    t = timing_of(err)
    assert t["hash"] == "device" and t["repacked"] == 0 and t["download_s"] == 0 and t["md5_host_s"] == 0 and t["pictures"] == npics


def test_decoders_behind_a_hub_on_the_gpu():
    """Six decoder processes (four 1080p streams of two kinds, two small ones) behind hevc_amd/hevcrecon_hub: the only
    process that opens the GPU.  Every status line carries the unmodified reference decoder's digests, and pictures of
    different streams shared batches."""
    from test_bridge_cli import BRIDGE, decoders_behind_a_hub
    from conftest import ROOT
    from hevc_amd import abi

    hub = os.path.join(ROOT, "hevc_amd", "hevcrecon_hub")
    if not (os.path.exists(BRIDGE) and os.path.exists(hub)):
        pytest.skip("bridge decoder or hub not built")
    batches, pictures = decoders_behind_a_hub(hub, abi.LIB_PATH, ["randomaccess_main_1080", "randomaccess_main_1080", "randomaccess_main_1080",
                                                                  "randomaccess_he10_1080", "wpb_240", "slices_tiles_nolf_240"])
    assert batches < pictures


def test_import_of_an_exported_picture_reconstructs_the_same():
    """hevcrecon_syntax_export on a host-only object, hevcrecon_syntax_import on an object of a context: the pictures of a
    stream reconstructed from imported blocks carry the reference decoder's digests."""
    import ctypes as C
    from hevc_amd import abi

    seq = load_golden("randomaccess_main_416x240")
    lib = abi.load_library()
    eng = engine.Engine(seq.seq_params)
    host = engine.detached_syntax(seq.seq_params)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        host.pack(pic)
        n = C.c_int64(0)
        assert lib.hevcrecon_syntax_export(host.h, None, 0, C.byref(n)) == 0
        blob = (C.c_uint8 * n.value)()
        assert lib.hevcrecon_syntax_export(host.h, blob, n.value, C.byref(n)) == 0
        assert lib.hevcrecon_syntax_import(syn.h, blob, n.value) == 0, lib.hevcrecon_last_error(eng.ctx)
        eng.reconstruct(syn, pic.pic_id)
        assert eng.md5([pic.pic_id]) == [pic.stage_md5[2]], pic.poc
    host.close()
    syn.close()
    eng.close()


def test_lost_reference_is_concealed_on_the_gpu(tmp_path):
    from test_bridge_cli import BRIDGE, lost_picture_case
    from hevc_amd import abi

    if not os.path.exists(BRIDGE):
        pytest.skip("bridge decoder not built (needs the reference tree at build time)")
    lost_picture_case(abi.LIB_PATH, tmp_path)


def test_packed_pictures_can_be_uploaded_again():
    """hevcrecon_syntax_upload: a packed picture copied to the device a second time (runs expanded again, while
    other pictures are being reconstructed) gives the same picture; an unfinished one is refused."""
    seq = load_golden("randomaccess_main_416x240")
    bd = bit_depths(seq)
    eng = engine.Engine(seq.seq_params)
    syns = []
    for pic in seq.pictures:
        syns.append(eng.new_syntax().pack(pic))
        eng.reconstruct(syns[-1], pic.pic_id)
    for pic, syn in zip(seq.pictures, syns):
        syn.upload()
    for pic, syn in zip(seq.pictures, syns):
        syn.upload()  # twice in a row, with the first copy possibly still in flight
        eng.reconstruct(syn, pic.pic_id)
        final = eng.download(pic.pic_id)
        assert [syntax.plane_md5(final[c], bd[c]) for c in range(3)] == pic.stage_md5[2], pic.poc
    fresh = eng.new_syntax()
    with pytest.raises(engine.EngineError):
        fresh.upload()
    fresh.close()
    for s in syns:
        s.close()
    eng.close()


@pytest.mark.parametrize("stream", ["randomaccess_main", "randomaccess_he10", "intra_main"] + ["randomaccess_main_1920x1080_s%d" % k for k in range(1, 8)])
def test_all_stages_md5_exact_1080p(stream):
    """The 1080p streams (all-intra, random access, 10-bit random access) at full size: every picture's three stages against the MD5s of the reference
    decoder's own pictures (before deblocking, after deblocking, final = the digest it checks against the SEI)."""
    seq = load_golden(stream if "1920x1080" in stream else stream + "_1920x1080")
    bd = bit_depths(seq)
    eng = engine.Engine(seq.seq_params)
    eng.keep_stages(True)
    syn = eng.new_syntax()
    for pic in seq.pictures:
        syn.pack(pic)
        eng.reconstruct(syn, pic.pic_id)
        for stage in range(3):
            got = eng.download(pic.pic_id) if stage == 2 else eng.download_stage(stage)
            assert [syntax.plane_md5(got[c], bd[c]) for c in range(3)] == pic.stage_md5[stage], (stream, pic.poc, stage)
        pic.drop_ctypes()
    syn.close()
    eng.close()


@pytest.mark.parametrize("bits,npics", [("lowdelay_P_main_240", 9), ("randomaccess_he10_240", 9), ("scalinglist_240", 9), ("lossless_240", 6), ("pcm_240", 5), ("pcmlf_240", 4), ("wpp_240", 9), ("wpb_240", 9), ("cip_240", 6), ("bd12_240", 5), ("slices_240", 5), ("tiles_240", 5), ("slices_nolf_240", 5), ("tiles_nolf_240", 5), ("slices_tiles_nolf_240", 9), ("randomaccess_main_1080", 60), ("randomaccess_he10_1080", 60),
                                        # all 60 frames of the all-intra 1080p configuration: the 10.9 MB bitstream is not committed, it lies
                                        # beside the compiled reference (tools/make_fixtures.sh puts it there); its digests are in hm_md5.json
                                        ("oracle/_ref/fixtures/intra_main_1080_f60", 60),
                                        # encoder option sweep (tools/make_fixtures.sh): per-CU QP changes + chroma QP offsets + deblocking
                                        # offsets; QP 22 with picture-based SAO; QP 45 without deblocking; transform trees of depth 3
                                        # without AMP / SAO / transform skip; low-delay B
                                        ("sweep_dqp", 5), ("sweep_q22", 4), ("sweep_q45", 5), ("sweep_tu", 5), ("sweep_lb", 6)])
def test_drop_in_behind_reference_decoder_cli(bits, npics):
    """The reference decoder's own CLI with its reconstruction call sites redirected to the engine
    (oracle/_ref/TAppDecoderBridge, HEVCRECON_MODE=engine): the reference parses, the GPU reconstructs, and the
    reference's unchanged MD5 check must print (OK) for every picture and exit 0."""
    import subprocess

    from conftest import ROOT
    from hevc_amd import abi

    exe = os.path.join(ROOT, "oracle", "_ref", "TAppDecoderBridge")
    if not os.path.exists(exe):
        pytest.skip("bridge decoder not built (needs the reference tree at build time)")
    env = dict(os.environ, HEVCRECON_MODE="engine", HEVCRECON_LIB=abi.LIB_PATH)
    path = os.path.join(ROOT, bits + ".bin") if "/" in bits else os.path.join(GOLDEN, bits + ".bin")
    if not os.path.exists(path):
        pytest.skip("bitstream not generated here (tools/make_fixtures.sh)")
    r = subprocess.run([exe, "-b", path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
    out = r.stdout.decode("latin1")
    assert r.returncode == 0, out[-2000:]
    assert out.count("(OK)") == npics and "ERROR" not in out
    key = {"oracle/_ref/fixtures/intra_main_1080_f60": "intra_main_1080_f60"}.get(bits, bits)
    want = json.load(open(os.path.join(GOLDEN, "hm_md5.json"))).get(key)
    if want is not None:  # the digests printed are the ones the unmodified reference decoder printed for this stream
        import re

        got = [[int(m.group(1)), m.group(2), m.group(3), m.group(4)]
               for m in re.finditer(r"POC\s+(\d+).*?\[MD5:([0-9a-f]{32}),([0-9a-f]{32}),([0-9a-f]{32}),\(OK\)\]", out)]
        assert got == want


def test_per_slice_parameters():
    """Slices with parameters of their own and uncrossed slice / tile boundaries, SAO forced on everywhere."""
    from test_emulated_kernels import _per_slice_parameters_match_oracle
    from hevc_amd import abi

    _per_slice_parameters_match_oracle(abi.load_library(), npics=4)


@pytest.mark.parametrize("width,height,bit_depth", [(8, 8, 8), (72, 40, 8), (424, 248, 8), (72, 40, 10), (8, 8, 12)])
def test_device_md5_padding(width, height, bit_depth):
    """Plane sizes that do not end on a 64-byte block boundary (the golden streams' planes all do): random pictures
    uploaded, hashed on the device and with hashlib."""
    from hevc_amd import abi

    sp = abi.SeqParams(width=width, height=height, bit_depth_luma=bit_depth, bit_depth_chroma=bit_depth, ctu_size=64, max_cu_depth=4)
    eng = engine.Engine(sp)
    rng = np.random.default_rng(width * 31 + height + bit_depth)
    want = []
    for i in range(3):
        planes = [rng.integers(0, 1 << bit_depth, size=(h, w)).astype(np.int16) for (w, h) in ((width, height), (width // 2, height // 2), (width // 2, height // 2))]
        eng.upload(i, i, planes)
        want.append([syntax.plane_md5(p, bit_depth) for p in planes])
    assert eng.md5([0, 1, 2]) == want
    assert eng.md5([2, 2, 0]) == [want[2], want[2], want[0]]
    eng.close()


@pytest.mark.parametrize("ring_bytes", [0, 48 << 10, 96 << 10, None])
def test_upload_ring(ring_bytes):
    """The upload ring in its corner cases (none, overflow and wrap-around, default) gives the same pictures."""
    from test_emulated_kernels import _ring_sizes_give_the_same_pictures
    from hevc_amd import abi

    _ring_sizes_give_the_same_pictures(abi.load_library(), ring_bytes)


BENCH_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"}


def _bench_line(cmd, env=None):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable] + cmd, cwd=root, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode("latin1")
    assert out.returncode == 0, text[-2000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-2000:]  # exactly one JSON line (rank 0)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_contract_single_gpu():
    d = _bench_line(["bench.py", "--steps", "1", "--warmup", "1", "--streams", "2", "--no-cpu-baseline", "--e2e-procs", "2"])
    assert BENCH_KEYS <= set(d)
    e2e = d.get("end_to_end")
    if e2e is not None:  # the decoder binaries travel with the tree when they were built
        assert "error" not in e2e, e2e
        assert e2e["concurrent"]["processes"] == 2 and e2e["concurrent"]["frames"] == 2 * 8 * 60 and d["packer_us_per_picture_c"] > 0
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and d["config"]["md5_exact"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_bench_batches_by_dependency_level():
    """`--group-by-level`: the pictures of a stream that do not refer to each other are reconstructed in one batch, in
    another order than they were decoded in and under picture ids of their own; the benchmark's parity pass checks the
    digest of every picture of every stream against the reference decoder's."""
    d = _bench_line(["bench.py", "--steps", "1", "--warmup", "1", "--streams", "3", "--group-by-level", "--no-cpu-baseline", "--no-extras"])
    assert d["config"]["group_by_level"] and d["config"]["md5_exact"] and d["config"]["md5_checked_pictures"] == 3 * 60
    assert d["value"] > 0 and d["roofline"]["pictures_per_launch"] > 2


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 path of the benchmark (rendezvous, barrier, max-over-ranks time, one line from rank 0), rehearsed
    with two ranks on the one GPU of the test machine; the real multi-GPU run uses RCCL and one GPU per rank."""
    d = _bench_line(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                     "29533", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--streams", "2", "--no-cpu-baseline"],
                    env={"HEVCRECON_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.gpu
def test_bench_starts_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver runs it: the script starts its two ranks itself; here both share the one
    GPU of the test machine over gloo.  64 streams over 8 GPUs = 8 per GPU is `--gpus 8 --total-streams 64`."""
    d = _bench_line(["bench.py", "--gpus", "2", "--total-streams", "5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                    env={"HEVCRECON_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["streams_per_gpu"] == 3  # rank 0 owns streams 0, 2, 4


def test_heterogeneous_batches():
    """Different streams at different picture positions in one batch (see the helper); two groups because a batch
    shares its sequence parameters: 8-bit (I / P / B, explicit weights in P slices, slices, tiles, constrained intra,
    scaling lists) and 10-bit (B slices with and without explicit weights)."""
    from test_emulated_kernels import _heterogeneous_batches_match_their_own_streams
    from hevc_amd import abi

    lib = abi.load_library()
    _heterogeneous_batches_match_their_own_streams(lib, ["lowdelay_P_main", "randomaccess_main", "intra_main", "wpp", "cip", "slices", "tiles", "scalinglist"], None, True)
    _heterogeneous_batches_match_their_own_streams(lib, ["randomaccess_he10", "wpb"], None, True)


@pytest.mark.parametrize("stream,npics", [("intra_main_416x240", 9), ("intra_main_1920x1080", 16)])
def test_batch_of_intra_pictures(stream, npics):
    """All pictures of an all-intra stream in one batch (see the helper): the intra kernel's borders-only instantiation,
    and the launch with more workgroups that batches of eight intra pictures and more get."""
    from test_emulated_kernels import _batch_of_intra_pictures_matches_reference
    from hevc_amd import abi

    _batch_of_intra_pictures_matches_reference(abi.load_library(), stream, npics)


def test_picture_copy():
    from test_emulated_kernels import _picture_copy_gives_the_same_picture
    from hevc_amd import abi

    _picture_copy_gives_the_same_picture(abi.load_library())


def test_known_answer_dc_picture():
    from test_emulated_kernels import _known_answer_dc_picture
    from hevc_amd import abi

    for (w, h, bd) in ((136, 72, 10), (1920, 1080, 8)):
        eng, syn = _known_answer_dc_picture(abi.load_library(), w, h, bd)
        syn.close()
        eng.close()
