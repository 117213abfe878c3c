"""The reference decoder's CLI with the engine behind it (oracle/_ref/TAppDecoderBridge), host side: what can be
checked without a GPU.  Engine mode runs here against the CPU emulation of the kernels (tests/emu), which is slow but
exercises exactly the host path a GPU run takes: CTUs handed to the packer while the reference parses, the picture
finish, the deferred hash check, the concealment of a lost reference.  The same runs on a real MI355X are in
tests/test_gpu_parity.py."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT
from hevc_amd import abi

BRIDGE = os.path.join(ROOT, "oracle", "_ref", "TAppDecoderBridge")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder")
EMU = os.path.join(ROOT, "tests", "emu", "libhevcrecon_emu.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))

needs_bridge = pytest.mark.skipif(not os.path.exists(BRIDGE), reason="bridge decoder not built (needs the reference tree at build time)")


@pytest.fixture(scope="module")
def emu_path():
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return EMU


def run_bridge(bitstream, lib, mode="engine", hash_mode=None, out=None, timing=True, hub=None):
    env = dict(os.environ, HEVCRECON_MODE=mode, HEVCRECON_LIB=lib)
    if hub:
        env["HEVCRECON_HUB"] = hub
    if hash_mode:
        env["HEVCRECON_HASH"] = hash_mode
    if timing:
        env["HEVCRECON_TIMING"] = "1"
    cmd = [BRIDGE, "-b", bitstream] + (["-o", out] if out else [])
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return r.returncode, r.stdout.decode("latin1"), r.stderr.decode("latin1")


def status_lines(text):
    return [[int(m.group(1)), m.group(2), m.group(3), m.group(4)]
            for m in re.finditer(r"POC\s+(\d+).*?\[MD5:([0-9a-f]{32}),([0-9a-f]{32}),([0-9a-f]{32}),\(OK\)\]", text)]


def timing_of(stderr):
    m = re.search(r"hm_bridge timing: (\{.*\})", stderr)
    assert m, stderr[-500:]
    return json.loads(m.group(1))


@needs_bridge
@pytest.mark.parametrize("stream,hash_mode", [("randomaccess_main_240", "device"), ("slices_tiles_nolf_240", "host")])
def test_ctus_are_packed_while_the_reference_parses(emu_path, stream, hash_mode):
    """Engine mode: every CTU reaches the packer from the decoder's CTU loop (none is packed again at the end), the
    status lines and digests are those of the unmodified reference decoder, exit code 0; with the device hash no
    picture is fetched or hashed on the host."""
    rc, out, err = run_bridge(os.path.join(GOLDEN, stream + ".bin"), emu_path, hash_mode=hash_mode)
    assert rc == 0, (out[-1000:], err[-1000:])
    want = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))[stream]
    assert status_lines(out) == want and "ERROR" not in out
    t = timing_of(err)
    assert t["pictures"] == len(want) and t["repacked"] == 0 and t["ctus"] == 28 * len(want)
    if hash_mode == "device":
        assert t["hash"] == "device" and t["hash_batches"] >= 1 and t["download_s"] == 0 and t["md5_host_s"] == 0
    else:
        assert t["hash"] == "host" and t["download_s"] > 0 and t["md5_host_s"] > 0


@needs_bridge
def test_pack_mode_times_the_host_side_without_a_gpu():
    """HEVCRECON_MODE=pack: the reference decodes by itself and every CTU is packed beside it into a host-only syntax
    object; the in-tree library loads without a GPU for that."""
    rc, out, err = run_bridge(os.path.join(GOLDEN, "randomaccess_main_240.bin"), abi.LIB_PATH, mode="pack")
    assert rc == 0 and out.count("(OK)") == 9
    t = timing_of(err)
    assert t["mode"] == "pack" and t["pictures"] == 9 and t["ctus"] == 9 * 28 and t["pack_s"] > 0 and t["finish_s"] > 0


@needs_bridge
def test_device_hash_steps_back_when_an_output_file_is_written(emu_path, tmp_path):
    """With -o the pictures have to reach the host anyway: HEVCRECON_HASH=device is ignored with a note, and the file is
    the reference decoder's."""
    bits = os.path.join(GOLDEN, "lowdelay_P_main_240.bin")
    rc, out, err = run_bridge(bits, emu_path, hash_mode="device", out=str(tmp_path / "eng.yuv"))
    assert rc == 0 and "fetched and hashed on the host" in err and out.count("(OK)") == 9
    subprocess.run([REFDEC, "-b", bits, "-o", str(tmp_path / "ref.yuv")], stdout=subprocess.DEVNULL, check=True)
    assert (tmp_path / "eng.yuv").read_bytes() == (tmp_path / "ref.yuv").read_bytes()


def lost_picture_case(lib, tmp_path):
    """A stream whose third picture (and the SEI message behind it) is missing: the reference conceals the lost
    reference with a copy of the nearest picture (TDecTop::xCreateLostPicture, TLibDecoder/TDecTop.cpp:214-261), and the
    engine's picture buffer must get the same copy (hevcrecon_picture_copy), or every later picture differs.
    The unmodified reference decoder itself does not survive such a stream: a few pictures on, its picture list is
    full, TDecTop::xGetNewPicBuffer extends it with a picture it has never created and destroys that one first
    (TLibDecoder/TDecTop.cpp:178-185), a null-pointer access in the reference's own code that engine mode inherits
    unchanged.  What both decoders have written by then -- the concealed picture and the pictures predicted from it
    -- must be the same."""
    import drop_nal

    bits = tmp_path / "lost.bin"
    bits.write_bytes(drop_nal.drop(open(os.path.join(GOLDEN, "lowdelay_P_main_240.bin"), "rb").read(), {7, 8}))
    ref = subprocess.run([REFDEC, "-b", str(bits), "-o", str(tmp_path / "ref.yuv")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rc, out, err = run_bridge(str(bits), lib, out=str(tmp_path / "eng.yuv"), timing=False)
    assert "hm_bridge" not in err, err[-500:]
    assert rc == ref.returncode
    want, got = (tmp_path / "ref.yuv").read_bytes(), (tmp_path / "eng.yuv").read_bytes()
    assert len(want) >= 5 * 416 * 240 * 3 // 2, "the reference decoder wrote fewer pictures than the concealment test needs"
    assert got == want


@needs_bridge
def test_lost_reference_is_concealed_in_the_engine_too(emu_path, tmp_path):
    lost_picture_case(emu_path, tmp_path)


def decoders_behind_a_hub(hub_binary, lib, streams):
    """Several decoder processes that never open the GPU, one hub process that does (hevc_amd/csrc/hub.cpp): every
    decoder packs on its own, hands its pictures over through shared memory, gets the digests back and prints the
    reference's status lines with them.  Streams with different sequence parameters get contexts of their own in the
    hub; pictures of streams that share them share batches."""
    import threading

    name = "test_%d" % os.getpid()
    hub = subprocess.Popen([hub_binary, "--name", name, "--idle-exit", "1"], stderr=subprocess.PIPE)
    results = [None] * len(streams)

    def one(k):
        results[k] = run_bridge(os.path.join(GOLDEN, streams[k] + ".bin"), lib, mode="hub", hub=name)

    threads = [threading.Thread(target=one, args=(k,)) for k in range(len(streams))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    try:
        hub_err = hub.communicate(timeout=60)[1].decode("latin1")
    except subprocess.TimeoutExpired:
        hub.kill()
        raise AssertionError("the hub did not leave after its last client")
    assert hub.returncode == 0, hub_err
    want = json.load(open(os.path.join(GOLDEN, "hm_md5.json")))
    total = 0
    for stream, (rc, out, err) in zip(streams, results):
        assert rc == 0, (stream, out[-500:], err[-500:])
        assert status_lines(out) == want[stream] and "ERROR" not in out, stream
        t = timing_of(err)
        assert t["mode"] == "hub" and t["repacked"] == 0 and t["pictures"] == len(want[stream])
        total += len(want[stream])
    m = re.search(r"(\d+) pictures in (\d+) batches", hub_err)
    assert m and int(m.group(1)) == total, hub_err
    return int(m.group(2)), total


@needs_bridge
def test_decoders_behind_a_hub(emu_path):
    hub = os.path.join(os.path.dirname(emu_path), "hevcrecon_hub_emu")
    decoders_behind_a_hub(hub, emu_path, ["randomaccess_main_240", "lowdelay_P_main_240", "randomaccess_he10_240"])


@needs_bridge
def test_a_decoder_without_a_hub_says_so(emu_path):
    rc, out, err = run_bridge(os.path.join(GOLDEN, "intra_main_240.bin"), emu_path, mode="hub", hub="nobody_%d" % os.getpid())
    assert rc != 0 and "no hub" in err
