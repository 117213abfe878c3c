#!/usr/bin/env python3
"""Benchmark of the picture-reconstruction engine on MI355X.

Workload: 1920x1080, 60-frame random-access streams encoded by the reference encoder
(cfg/encoder_randomaccess_main.cfg) from the synthetic clips of tools/gen_clip.py, eight different clips (seeds
20240607 .. 20240614), as parsed by the reference decoder (tests/golden/randomaccess_main_1920x1080*.syn.xz).
Each GPU reconstructs `--streams` independent streams (the stream-sharded multi-stream case): stream s plays clip
s mod 8, over and over, in decoding order.  The streams of a GPU are spread over `--contexts` engine contexts; a
context batches the pictures its streams have reached into one set of launches per step.  Streams of one context
run at the same decoding position (a host that batches groups streams by position, so that the long dependency
chain of intra pictures is shared by a whole batch); the contexts are half a stream apart.  `--stagger` instead
starts every stream at its own position.

A step = every stream advances by 60 pictures.  The packed syntax is resident in HBM before the timed region
starts; after it, one more pass checks every final picture of every stream against the reference decoder's MD5.

Prints one JSON line (rank 0).
"""
import argparse
import copy
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hevc_amd import engine, sharding, syntax  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
ID_STRIDE = 128         # picture ids of stream s are offset by s * ID_STRIDE
KERNELS = ["k_inter_tile64", "k_inter_tile32", "k_inter_tile16", "k_inter_blk", "k_transform", "k_intra", "k_deblock_v",
           "k_deblock_h", "k_sao"]


_T0 = time.time()


def progress(what):
    """A line on stderr now and then: a run that stays silent for minutes is taken for hung by the job runner."""
    sys.stderr.write("[bench %6.1f s] %s\n" % (time.time() - _T0, what))
    sys.stderr.flush()


def clone_for_stream(pic, s, ref_index=None):
    """The picture as stream s sees it: its reference ids moved into the stream's range.  With `ref_index` (decoding
    index of every reference, see reference_indices) pictures are named by their decoding index instead of by the
    decoder's picture buffer: every picture of a pass then has an id of its own, and pictures can be reconstructed in
    another order than they were decoded in without one overwriting a buffer that another still reads."""
    p = copy.copy(pic)
    p.pic_params = type(pic.pic_params).from_buffer_copy(pic.pic_params)
    for l in range(2):
        for r in range(p.pic_params.num_ref[l]):
            p.pic_params.ref_pic_id[l][r] = (pic.pic_params.ref_pic_id[l][r] if ref_index is None else ref_index[l][r]) + s * ID_STRIDE
    return p


def reference_indices(seq):
    """Per picture (decoding order) and list, the decoding index of every reference: the latest earlier picture that was
    decoded into the buffer the list names."""
    last, out = {}, []
    for t, pic in enumerate(seq.pictures):
        pp = pic.pic_params
        out.append([[last[pp.ref_pic_id[l][r]] for r in range(pp.num_ref[l])] for l in range(2)])
        last[pic.pic_id] = t
    return out


def level_schedule(seq, window=8):
    """Batches of mutually independent pictures of ONE stream.  Parsing never reads samples, so a decoder can parse a
    group of pictures ahead and hand its pictures over by dependency level instead of one by one: of the random-access
    hierarchy's eight pictures per group (decoded 8 4 2 1 3 6 5 7) first 8, then 4, then 2 and 6, then 1 3 5 7, which
    refer to nothing inside their batch (TLibCommon/TComSlice.cpp:450-465 builds the lists; the dependencies are those
    of the GOP table in cfg/encoder_randomaccess_main.cfg:29-36).  The first picture stands alone; after it, windows of `window`
    pictures in decoding order, inside a window a picture's level is one more than the highest level among its
    references in the window."""
    refs = reference_indices(seq)
    n = len(seq.pictures)
    batches = [[0]]
    for w0 in range(1, n, window):
        idx = list(range(w0, min(n, w0 + window)))
        level = {}
        for t in idx:
            level[t] = 1 + max([level[r] for l in range(2) for r in refs[t][l] if r in level], default=-1)
        for lv in range(max(level.values()) + 1):
            batches.append([t for t in idx if level[t] == lv])
    return batches


def kernel_bytes(sp, st, sao):
    """Algorithmic bytes per kernel for one picture (its packing statistics `st`): the terms of
    S*b*(1 store + 2 deblock + 2 SAO) + MC reference reads + 2*coefficients + 16 per 4x4 record (or 16 per
    uniform-motion region), attributed to the kernel that moves them.  Luma + chroma = 1.5 bytes per luma
    sample at 8 bits."""
    b = 2 if sp.bit_depth_luma > 8 else 1
    s_bytes = sp.width * sp.height * 3 // 2 * b
    out = dict.fromkeys(KERNELS, 0)
    n4, intra4 = st[8], st[9]
    tile_samples = tile_ref_samples = 0
    for t, name in enumerate(KERNELS[:3]):
        out[name] += (st[23 + t] + st[26 + t]) * 3 // 2 * b + 16 * st[16 + t]
        tile_samples += st[23 + t]
        tile_ref_samples += st[26 + t]
    blk_units = (n4 - intra4) - tile_samples // 16
    out["k_inter_blk"] += blk_units * 24 * b + (st[7] - tile_ref_samples * 3 // 2) * b + 16 * blk_units
    out["k_intra"] += intra4 * 24 * b
    out["k_transform"] += 2 * st[4]
    out["k_deblock_v"] += s_bytes
    out["k_deblock_h"] += s_bytes
    out["k_sao"] += 2 * s_bytes if sao else 0
    return out


REF_DECODER = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder")
BRIDGE_DECODER = os.path.join(ROOT, "oracle", "_ref", "TAppDecoderBridge")
RA_BITSTREAM = os.path.join(ROOT, "tests", "golden", "randomaccess_main_1080.bin")


def _decode_once(exe, env=None):
    t0 = time.time()
    out = subprocess.run([exe, "-b", RA_BITSTREAM], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True, env=env).stdout.decode("latin1")
    if out.count("(OK)") != 60:
        raise RuntimeError("%s did not report 60 matching pictures" % os.path.basename(exe))
    return time.time() - t0


def cpu_baseline(seconds_target=10.0):
    """Reference decoder (HM 9.1 TAppDecoder, -O3, single-threaded; parse + reconstruction + MD5 check) on the 1080p RA
    bitstream, on the host cores of this machine: N concurrent decoders, one stream per core, N = the cores this job
    may use (at most 16), and the single-process figure."""
    if os.path.exists(REF_DECODER) and os.access(REF_DECODER, os.X_OK) and os.path.exists(RA_BITSTREAM):
        frames, t0 = 0, time.time()
        while time.time() - t0 < seconds_target / 2:
            _decode_once(REF_DECODER)
            frames += 60
        t_single = time.time() - t0
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        n = max(1, min(16, ncores))
        loop = ("import subprocess,sys,time\nt0=time.time();k=0\nwhile time.time()-t0<%f:\n"
                "    o=subprocess.run([%r,'-b',%r],stdout=subprocess.PIPE,stderr=subprocess.STDOUT).stdout\n"
                "    assert o.count(b'(OK)')==60\n    k+=60\nprint(k, time.time()-t0)\n") % (seconds_target / 2, REF_DECODER, RA_BITSTREAM)
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, "-c", loop], stdout=subprocess.PIPE) for _ in range(n)]
        outs = [p.communicate()[0].split() for p in procs]
        wall = time.time() - t0
        total = sum(int(o[0]) for o in outs)
        return {"value": round(total / wall, 2), "unit": "frames/s", "cores": n, "kind": "reference",
                "single_process_value": round(frames / t_single, 2),
                "sample": "%d concurrent single-threaded oracle/_ref/TAppDecoder processes decoding the 60-frame 1080p RA stream "
                          "(parse + reconstruction + MD5 check) for %.1f s: %d frames; one process alone: %d frames in %.1f s"
                          % (n, wall, total, frames, t_single)}
    # the reference binary did not travel: time the C restatement (reconstruction only) instead
    from oracle.refmodel import RefModel

    seq = syntax.read_dump(os.path.join(ROOT, "tests", "golden", "randomaccess_main_1920x1080.syn.xz"), max_pictures=24)
    ref = RefModel(seq.seq_params)
    t0 = time.time()
    for pic in seq.pictures:
        ref.reconstruct(pic)
    dt = time.time() - t0
    return {"value": round(len(seq.pictures) / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "first %d pictures of the 1080p RA stream through oracle/recon_ref.c (reconstruction only), %.1f s" % (len(seq.pictures), dt)}


HUB_BINARY = os.path.join(ROOT, "hevc_amd", "hevcrecon_hub")


def _bridge_env(device, hash_mode, hub=None):
    from hevc_amd import abi

    if hub is not None:  # the decoder never opens the GPU: it hands its packed pictures to the hub process
        return dict(os.environ, HEVCRECON_MODE="hub", HEVCRECON_HUB=hub, HEVCRECON_LIB=abi.LIB_PATH, HEVCRECON_TIMING="1")
    env = dict(os.environ, HEVCRECON_MODE="engine", HEVCRECON_LIB=abi.LIB_PATH, HEVCRECON_DEVICE=str(device), HEVCRECON_HASH=hash_mode,
               HEVCRECON_TIMING="1")
    if os.environ.get("HEVCRECON_E2E_HWQ"):
        env["GPU_MAX_HW_QUEUES"] = os.environ["HEVCRECON_E2E_HWQ"]
    return env


def _bridge_timing(stderr_bytes):
    import re

    m = re.search(r"hm_bridge timing: (\{.*\})", stderr_bytes.decode("latin1"))
    return json.loads(m.group(1)) if m else None


def _per_picture_ms(ts):
    """Average host time per picture (ms) over the timing records of one or more decoder processes."""
    n = sum(t["pictures"] for t in ts)
    keys = ["parse_and_rest_s", "pack_s", "finish_s", "submit_s", "motion_s", "wait_s", "download_s", "md5_host_s", "hash_device_s"]
    return {k[:-2]: round(sum(t[k] for t in ts) / n * 1e3, 3) for k in keys}


def end_to_end(device, n_proc, loops=8):
    """Decoded frames/s through the reference decoder's own CLI with the engine behind it (oracle/_ref/TAppDecoderBridge,
    HEVCRECON_MODE=engine: the reference parses, every CTU is packed where the reference would have reconstructed it,
    the GPU reconstructs).  Must run before this process has touched the GPU (the decoders are GPU processes of their
    own).
      concurrent: n_proc decoder processes share the GPU, one stream per host core like the CPU baseline, each
        decoding the 60-frame 1080p RA stream `loops` times over (the bitstream concatenated with itself: every copy
        starts with parameter sets and an IDR picture), pictures hashed on the GPU (HEVCRECON_HASH=device) and checked
        against the SEI digests by the decoder itself; timed from the first process's first picture to the last
        process's last one (wall clocks of the processes), process start and GPU initialisation reported apart;
      one_process: the same stream once, (a) with the device hash, (b) with every picture fetched and hashed by the
        reference on the host thread (what writing an output file costs), (c) the unmodified reference decoder;
      per_picture_ms: where a decoder's host thread spends a picture's time (clock_gettime around each stage
        inside the bridge)."""
    if not (os.path.exists(BRIDGE_DECODER) and os.path.exists(REF_DECODER) and os.path.exists(RA_BITSTREAM)):
        return None
    import tempfile

    out = {"frames_per_stream": 60}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            looped = os.path.join(tmp, "looped.bin")
            one = open(RA_BITSTREAM, "rb").read()
            with open(looped, "wb") as f:
                for _ in range(loops):
                    f.write(one)
            stages = os.environ.get("HEVCRECON_E2E_STAGES", "direct,hub,host,one").split(",")

            launched = [0]

            def start(bits, env):
                """A decoder process writing into files of its own (a pipe that nobody drains stalls the decoder as soon as
                it holds 64 KB of status lines)."""
                k = launched[0]
                launched[0] += 1
                fo, fe = open(os.path.join(tmp, "out_%d" % k), "wb"), open(os.path.join(tmp, "err_%d" % k), "wb")
                p = subprocess.Popen([BRIDGE_DECODER, "-b", bits], env=env, stdout=fo, stderr=fe)
                p.files = (fo, fe)
                return p

            def run_all(procs, limit=240):
                """stdout / stderr of every decoder; a decoder that does not finish in time is killed (and fails the stage)"""
                deadline = time.time() + limit
                res = []
                for p in procs:
                    try:
                        p.wait(timeout=max(1.0, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                    for f in p.files:
                        f.close()
                    res.append((open(p.files[0].name, "rb").read(), open(p.files[1].name, "rb").read()))
                return res

            env = _bridge_env(device, "device")
            progress("end to end: %d decoders, each with a GPU context of its own" % n_proc)
            t0 = time.time()
            procs = [start(looped if "direct" in stages else RA_BITSTREAM, env) for _ in range(n_proc)]
            res = run_all(procs)
            if "direct" not in stages:
                loops_direct = 1
            else:
                loops_direct = loops
            wall = time.time() - t0
            ts = []
            for p, (so, se) in zip(procs, res):
                t = _bridge_timing(se)
                if p.returncode != 0 or so.count(b"(OK)") != 60 * loops_direct or t is None:
                    raise RuntimeError("a concurrent decoder failed (exit %d, %d of %d pictures OK): %s" % (p.returncode, so.count(b"(OK)"), 60 * loops_direct, se[-300:]))
                ts.append(t)
            span = max(t["wall_last"] for t in ts) - min(t["wall_first"] for t in ts)
            frames = sum(t["pictures"] for t in ts)
            out["concurrent"] = {
                "processes": n_proc, "frames": frames, "frames_per_s": round(frames / span, 1), "first_to_last_picture_s": round(span, 3),
                "frames_per_s_with_start_up": round(frames / wall, 1),
                "start_up_s": round(sum(t["startup_s"] for t in ts) / n_proc, 3), "gpu_init_s": round(sum(t["engine_init_s"] for t in ts) / n_proc, 3),
                "hash": "device", "pictures_per_hash_request": round(frames / max(1, sum(t["hash_batches"] for t in ts)), 1),
                "per_picture_ms": _per_picture_ms(ts)}
            # the same decoders behind ONE process that owns the GPU (hevc_amd/hevcrecon_hub): the decoder processes pack on
            # their cores and never open the device; the hub reconstructs what all of them have ready as one batch
            if os.path.exists(HUB_BINARY) and "hub" in stages:
                progress("end to end: %d decoders behind one hub process" % n_proc)
                hub_name = "bench_%d" % os.getpid()
                hub = subprocess.Popen([HUB_BINARY, "--name", hub_name, "--device", str(device), "--idle-exit", "1"], stderr=subprocess.PIPE)
                procs = [start(looped, _bridge_env(device, "device", hub_name)) for _ in range(n_proc)]
                res = run_all(procs, 120)
                try:
                    hub_err = hub.communicate(timeout=30)[1].decode("latin1")
                except subprocess.TimeoutExpired:
                    hub.kill()
                    hub_err = "hub did not leave"
                ts = [_bridge_timing(se) for _, se in res]
                if all(p.returncode == 0 for p in procs) and all(t is not None for t in ts) and all(so.count(b"(OK)") == 60 * loops for so, _ in res):
                    span = max(t["wall_last"] for t in ts) - min(t["wall_first"] for t in ts)
                    frames = sum(t["pictures"] for t in ts)
                    import re

                    m = re.search(r"\(([0-9.]+) per batch\)", hub_err)
                    progress(hub_err.strip().replace("\n", " | "))
                    out["concurrent_through_hub"] = {"processes": n_proc, "gpu_processes": 1, "frames": frames, "frames_per_s": round(frames / span, 1),
                                                     "pictures_per_batch": float(m.group(1)) if m else None, "hash": "device",
                                                     "per_picture_ms": _per_picture_ms(ts)}
                    # packing alone, from C, on cores that are all busy decoding (these processes never touch the device:
                    # with a context of its own a decoder's begin() can wait for the device, which its timer would count)
                    out["packer_us_per_picture_c"] = round(sum(t["pack_s"] + t["finish_s"] for t in ts) / frames * 1e6, 1)
                else:
                    out["concurrent_through_hub"] = {"error": (hub_err + " ".join(se.decode("latin1")[-200:] for _, se in res if se))[-400:]}
            # the same with every picture fetched and hashed by the reference on the host thread
            progress("end to end: %d decoders that fetch and hash on the host" % n_proc)
            procs = [start(looped if "host" in stages else RA_BITSTREAM, _bridge_env(device, "host")) for _ in range(n_proc)]
            res = run_all(procs)
            ts = [_bridge_timing(se) for _, se in res]
            if "host" in stages and all(p.returncode == 0 for p in procs) and all(t is not None for t in ts) and all(so.count(b"(OK)") == 60 * loops for so, _ in res):
                span = max(t["wall_last"] for t in ts) - min(t["wall_first"] for t in ts)
                frames = sum(t["pictures"] for t in ts)
                out["concurrent_host_hash"] = {"processes": n_proc, "frames": frames, "frames_per_s": round(frames / span, 1),
                                               "per_picture_ms": _per_picture_ms(ts)}
            # one process alone
            progress("end to end: one decoder alone")
            single = {}
            for name, hash_mode in (("device_hash", "device"), ("host_hash", "host")):
                r = subprocess.run([BRIDGE_DECODER, "-b", RA_BITSTREAM], env=_bridge_env(device, hash_mode), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                t = _bridge_timing(r.stderr)
                if r.returncode != 0 or r.stdout.count(b"(OK)") != 60 or t is None:
                    raise RuntimeError("the %s decoder failed: %s" % (name, r.stderr[-300:]))
                single[name] = {"frames_per_s": round(60 / t["first_to_last_s"], 1), "start_up_s": round(t["startup_s"], 3),
                                "per_picture_ms": _per_picture_ms([t])}
            single["engine_behind_reference_cli_frames_per_s_with_start_up"] = round(60 / min(_decode_once(BRIDGE_DECODER, _bridge_env(device, "host")) for _ in range(2)), 2)
            single["reference_cli_frames_per_s_with_start_up"] = round(60 / min(_decode_once(REF_DECODER) for _ in range(2)), 2)
            out["one_process"] = single
            if "packer_us_per_picture_c" not in out:
                pm = single["device_hash"]["per_picture_ms"]
                out["packer_us_per_picture_c"] = round((pm["pack"] + pm["finish"]) * 1e3, 1)
    except (RuntimeError, subprocess.CalledProcessError, OSError) as e:
        out["error"] = str(e)[:300]
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh copies of this script, one rank per GPU, before this
    process has touched a GPU (it never does), pass rank 0's line through and return the worst exit status."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class Workload:
    """The streams of one GPU: packed pictures resident in HBM, and the batches that advance every stream by one
    picture per step.  Stream s starts at step off[s] and then plays its 60 pictures over and over (picture 0 is an
    intra picture, so every pass is a correct decode and every picture can be checked at any time)."""

    def __init__(self, seqs, S, C, device, upload_ring, stagger, by_level=False):
        self.seqs, self.S, self.C = seqs, S, C
        self.sp = seqs[0].seq_params
        self.npic = len(seqs[0].pictures)
        # what a step reconstructs of every stream: picture t (decoding order), or, by dependency level, a batch of
        # pictures that do not refer to each other (the same for every clip: they share the encoder's configuration)
        self.by_level = by_level
        self.schedule = [[t] for t in range(self.npic)]
        self.ref_index = [None] * len(seqs)
        if by_level:
            self.schedule = level_schedule(seqs[0])
            for seq in seqs[1:]:
                if level_schedule(seq) != self.schedule:
                    raise SystemExit("bench: the clips differ in their reference structure")
            self.ref_index = [reference_indices(seq) for seq in seqs]
        self.period = len(self.schedule)  # steps per pass
        nseeds = len(seqs)
        self.seed_of = [s % nseeds for s in range(S)]
        self.ctx_of = [(s // nseeds) % C if S >= nseeds * C else s % C for s in range(S)]
        self.groups = [[s for s in range(S) if self.ctx_of[s] == c] for c in range(C)]
        self.engs = [engine.Engine(self.sp, device=device) for _ in range(C)]
        for c in range(C):
            self.engs[c].set_upload_ring(min(4 << 30, len(self.groups[c]) * self.npic * (256 << 10)) if upload_ring else 0)
        # pack every picture of every stream (untimed); statistics and packing time per (clip, picture)
        self.syns = [[None] * S for _ in range(self.npic)]
        self.stats = [[None] * self.npic for _ in range(nseeds)]
        self.pack_s, self.pack_n = 0.0, 0
        # in the order a host with one decoder per stream finishes them: picture t of every stream, then picture t + 1
        # (what is finished together is staged side by side in a context's upload ring and travels in one copy)
        for t in range(self.npic):
            for k, seq in enumerate(seqs):
                pic = seq.pictures[t]
                pic.ctu_structs()
                for s in range(S):
                    if self.seed_of[s] == k:
                        y = self.engs[self.ctx_of[s]].new_syntax().pack(clone_for_stream(pic, s, self.ref_index[k][t] if by_level else None))
                        self.syns[t][s] = y
                        self.pack_s += y.pack_seconds
                        self.pack_n += 1
                        if self.stats[k][t] is None:
                            self.stats[k][t] = y.stats()
                pic.drop_ctypes()
            if t % 10 == 9:
                progress("packed %d of %d pictures of %d streams" % (t + 1, self.npic, S))
        self.set_offsets(stagger)

    def set_offsets(self, stagger):
        """Grouped (default): the streams of a context share their position, the contexts are 60 / C pictures apart.
        Staggered: every stream of a context at its own position."""
        npic = self.period
        self.off = [0] * self.S
        for c, g in enumerate(self.groups):
            for j, s in enumerate(g):
                self.off[s] = (c * npic // self.C + (j * npic // len(g) if stagger else 0)) % npic
        self.max_off = max(self.off)
        self._prepared = {}

    def members(self, c, g):
        return [(s, t) for s in self.groups[c] if g >= self.off[s] for t in self.schedule[(g - self.off[s]) % self.period]]

    def prepared(self, c, g):
        """Argument arrays of context c's batch at step g (steps past the ramp-up repeat with the period of a pass)."""
        key = (c, g if g < self.max_off else self.max_off + (g - self.max_off) % self.period)
        if key not in self._prepared:
            m = self.members(c, g)
            self._prepared[key] = self.engs[c].prepare_batch([self.syns[t][s] for s, t in m], [self.pic_id(s, t) for s, t in m]) if m else None
        return self._prepared[key]

    def pic_id(self, s, t):
        return (t if self.by_level else self.seqs[self.seed_of[s]].pictures[t].pic_id) + s * ID_STRIDE

    def passes(self, first_step, n):
        """n passes of every stream over its pictures, from step first_step on; returns the next step."""
        for g in range(first_step, first_step + n * self.period):
            self.step(g)
        return first_step + n * self.period

    def ramp_up(self):
        """Streams start one after the other; returns the first step at which all of them run."""
        for g in range(self.max_off):
            self.step(g)
        return self.max_off

    def timed_passes(self, n=2, warm=1):
        """Frames per second over n passes (after the ramp-up and `warm` passes)."""
        g = self.passes(self.ramp_up(), warm)
        self.synchronize()
        t0 = time.perf_counter()
        self.passes(g, n)
        self.synchronize()
        return n * self.npic * self.S / (time.perf_counter() - t0)

    def step(self, g, upload=False):
        for c in range(self.C):
            p = self.prepared(c, g)
            if p is not None:
                if upload:
                    self.engs[c].upload_prepared(p)
                self.engs[c].reconstruct_prepared(p)

    def synchronize(self):
        for e in self.engs:
            e.synchronize()

    def close(self):
        for row in self.syns:
            for y in row:
                if y is not None:
                    y.close()
        for e in self.engs:
            e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=128,
                    help="independent streams per GPU; larger batches amortise the latency-bound intra chains")
    ap.add_argument("--contexts", type=int, default=2,
                    help="engine contexts (HIP streams) the streams of this GPU are spread over; with several, one context's "
                         "latency-bound intra kernel runs beside the others' throughput-bound kernels")
    ap.add_argument("--seeds", type=int, default=8, help="different clips among the streams (the fixture and its _s1 .. _s7 variants)")
    ap.add_argument("--group-by-level", action="store_true",
                    help="a step reconstructs, of every stream, a batch of pictures that do not refer to each other (dependency levels of "
                         "the stream's group of pictures) instead of one picture in decoding order")
    ap.add_argument("--stagger", action="store_true", help="every stream at its own decoding position (default: one position per context)")
    ap.add_argument("--fixture", default="randomaccess_main_1920x1080")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (hand-over, 1 and 8 streams, staggered, end to end)")
    ap.add_argument("--no-handover", action="store_true", help="skip the passes that time the syntax hand-over")
    ap.add_argument("--extras", default="e2e,handover,fewer,stagger",
                    help="which side measurements to run (comma separated): e2e (decoder processes through the CLI), handover, "
                         "fewer (1 and 8 streams), stagger (every stream at its own position)")
    ap.add_argument("--no-upload-ring", dest="upload_ring", action="store_false",
                    help="every packed picture in pinned buffers of its own instead of the context's upload ring (one copy command per picture "
                         "instead of a few per batch)")
    ap.add_argument("--check-two-streams-only", action="store_true",
                    help="parity pass without the device hash of every stream (profiling runs: keeps k_md5 out of the kernel statistics)")
    ap.add_argument("--e2e-procs", type=int, default=0,
                    help="decoder processes of the end-to-end measurement (default: the host cores of this job, at most 16)")
    ap.add_argument("--rehearse", action="store_true", help="no GPU work: only the multi-rank plumbing (used by the CPU tests)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="streams of the whole job, dealt to the GPUs round-robin (stream i -> rank i mod N); overrides --streams: "
                         "64 on 8 GPUs is 8 independent streams per GPU")
    args = ap.parse_args()
    extras = set() if args.no_extras else set(x for x in args.extras.split(",") if x)
    if args.no_handover:
        extras.discard("handover")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    backend = os.environ.get("HEVCRECON_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on fewer GPUs
    if world > 1:
        import torch
        import torch.distributed as dist

        ndev = torch.cuda.device_count()
        if args.rehearse:
            backend = "gloo"
        elif backend == "nccl" and local_rank >= ndev:
            raise SystemExit("bench: rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
        local_rank = local_rank % max(ndev, 1)
        if ndev:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    if args.rehearse:
        # the N > 1 plumbing alone (spawn, rendezvous, stream sharding, barrier, reductions), for machines without a GPU
        mine = sharding.streams_for_rank(args.total_streams or args.streams * world, rank, world)
        if dist is not None:
            dist.barrier()
        t = sharding.max_over_ranks(1.0 + rank, dist)
        total = sharding.sum_over_ranks(len(mine), dist)
        if rank == 0:
            print(json.dumps({"rehearsal": True, "value": None, "n_gpus": world, "streams_total": int(total),
                              "streams_rank0": mine, "max_over_ranks": t}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.gpus != world:
        raise SystemExit("bench: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.total_streams:
        args.streams = len(sharding.streams_for_rank(args.total_streams, rank, world))
        if args.streams == 0:
            raise SystemExit("bench: fewer streams than GPUs")

    # end to end through the decoder CLI: N decoder processes of their own on the GPU, measured before this process
    # opens the device (the machine allows only so many GPU processes at once)
    e2e = None
    if "e2e" in extras and world == 1 and args.fixture == "randomaccess_main_1920x1080":
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        progress("end to end: decoder processes sharing the GPU")
        e2e = end_to_end(local_rank, args.e2e_procs or max(1, min(16, ncores)))
        progress("end to end done")

    golden = os.path.join(ROOT, "tests", "golden")
    names = [args.fixture] + ["%s_s%d" % (args.fixture, k) for k in range(1, 8)]
    names = [n for n in names if os.path.exists(os.path.join(golden, n + ".syn.xz"))][:max(1, args.seeds)]
    seqs = [syntax.read_dump(os.path.join(golden, n + ".syn.xz")) for n in names[:max(1, min(len(names), args.streams))]]
    sp = seqs[0].seq_params
    S = args.streams
    C = max(1, min(args.contexts, S))
    W = Workload(seqs, S, C, local_rank, args.upload_ring, args.stagger, args.group_by_level)
    npic, engs, groups, period = W.npic, W.engs, W.groups, W.period
    W.synchronize()
    alg_bytes_stream = [sum(W.syns[t][s].algorithmic_bytes() for t in range(npic)) for s in range(S)]

    def sync():
        W.synchronize()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    # ramp-up (streams start one after the other) and warm-up passes, then the timed passes
    progress("warm-up and timed passes")
    g = W.passes(W.ramp_up(), max(1, args.warmup))
    sync()
    t0 = time.perf_counter()
    g = W.passes(g, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, device=red_dev)

    # ---- parity: one more (untimed) pass; after every step the picture every stream has just finished is hashed on
    # the device and compared with the reference decoder's MD5 (the decoder recycles picture ids, so the check cannot
    # wait); the first and last stream are also downloaded and hashed on the host
    progress("timed passes done: %.1f frames/s; checking every picture" % (npic * S * args.steps / elapsed))
    bd = [sp.bit_depth_luma, sp.bit_depth_chroma, sp.bit_depth_chroma]
    checked = 0
    for gg in range(g, g + period):
        W.step(gg)
        for c in range(C):
            m = W.members(c, gg)
            want = [W.seqs[W.seed_of[s]].pictures[t].stage_md5[2] for s, t in m]
            for (s, t), w in zip(m, want):
                if s in (0, S - 1):
                    planes = engs[c].download(W.pic_id(s, t))
                    if [syntax.plane_md5(planes[k], bd[k]) for k in range(3)] != w:
                        raise SystemExit("bench: stream %d picture %d differs from the reference decoder's MD5" % (s, t))
            if not args.check_two_streams_only:
                for (s, t), got, w in zip(m, engs[c].md5([W.pic_id(s, t) for s, t in m]), want):
                    if got != w:
                        raise SystemExit("bench: stream %d picture %d differs from the reference decoder's MD5 (device hash)" % (s, t))
                    checked += 1
    g += period

    # ---- per-kernel time with HIP events on the engine's stream (one extra, untimed pass of context 0 alone: with
    # several contexts the kernels of different contexts overlap, so their times are taken without the others) ----
    progress("per-kernel times")
    eng = engs[0]
    eng.profile_enable(True)
    for gg in range(g, g + period):
        p = W.prepared(0, gg)
        if p is not None:
            eng.reconstruct_prepared(p)
    prof = eng.profile_read()
    eng.profile_enable(False)
    S_prof = len(groups[0])
    kb = dict.fromkeys(KERNELS, 0)
    for s in groups[0]:
        k = W.seed_of[s]
        for t, pic in enumerate(W.seqs[k].pictures):
            one = kernel_bytes(sp, W.stats[k][t], bool(pic.pic_params.sao_luma or pic.pic_params.sao_chroma))
            for name in KERNELS:
                kb[name] += one[name]
    # hardware-counter figures of this very configuration, when they have been collected and committed
    # (tools/pmc_passes.sh -> tools/pmc_traffic.py -> profiles/hbm_traffic.json): HBM bytes and vector-ALU occupancy per launch
    counters = {}
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            for cfg in json.load(f).get("configs", []):
                if cfg.get("streams") == S_prof and cfg.get("fixture") == args.fixture and cfg.get("clips", 1) == len(seqs):
                    counters = cfg.get("kernels", {})
    except (OSError, ValueError):
        pass
    kernels = {}
    for name, nbytes in kb.items():
        ms, launches = prof[name]
        kernels[name] = {"ms_per_step": round(ms, 4), "launches": int(launches),
                         "alg_gbps": round(nbytes / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}
        # what the kernel really moved to and from HBM (the algorithmic figure also counts samples a filter skips and
        # bytes that never leave the caches)
        if name in counters and ms > 0:
            kernels[name]["hbm_traffic_gbps"] = round(counters[name]["bytes_per_launch"] * launches / (ms * 1e-3) / 1e9, 2)
    dom = max(kb, key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    achieved = kb[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    gpu_ms = sum(prof[k][0] for k in kb)

    # HBM traffic and vector-ALU occupancy of the dominant kernel: hardware counters, not measurable from inside the
    # process; taken from the committed PMC passes (tools/pmc_passes.sh -> tools/pmc_traffic.py) when they were
    # collected on this very configuration.  The kernel is called bound by whichever of the two resources it uses more
    # of (or by latency, below): HBM (counter bytes per launch over the launch time measured here, against the peak) or vector-ALU issue
    # slots; `achieved` / `frac` stay the algorithmic bytes against the HBM peak either way.
    traffic, valu, bound = None, None, "hbm"
    if dom in counters:
        e = counters[dom]
        traffic = e["bytes_per_launch"]
        if "valu_issue_frac" in e:
            valu = {"issue_slot_frac": e["valu_issue_frac"], "insts_per_wave": e["valu_per_wave"], "wait_frac": e["wait_frac"],
                    "source": "profiles/hbm_traffic.json (PMC passes of this configuration)"}
            if dom_ms > 0:
                valu["hbm_traffic_frac"] = round(traffic * dom_launches / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                if e["valu_issue_frac"] > valu["hbm_traffic_frac"]:
                    bound = "valu"
                # neither resource a quarter used and the waves parked most of the time: a chain of dependent steps
                # (the intra wavefront), bound by the latency of its hand-overs
                if max(e["valu_issue_frac"], valu["hbm_traffic_frac"]) < 0.25 and e["wait_frac"] > 0.6:
                    bound = "latency"

    frames_total = sharding.sum_over_ranks(npic * S * args.steps, dist, device=red_dev)
    value = frames_total / elapsed
    result = {
        "metric": "1080p_ra_reconstructed_frames_per_s" if args.fixture.startswith("randomaccess_main") else "reconstructed_frames_per_s",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8" if sp.bit_depth_luma <= 8 else "u16",
        "data": "synthetic",
        "config": {"workload": "%dx%d %d-frame streams (encoder_%s.cfg) of %d different clips, %d independent streams per GPU "
                               "over %d engine contexts, %s%s, stream-sharded, no collectives"
                               % (sp.width, sp.height, npic, args.fixture.rsplit("_", 1)[0], len(seqs), S, C,
                                  "every stream at its own decoding position" if args.stagger else
                                  "the streams of a context at one decoding position, the contexts %d pictures apart" % (npic // C),
                                  ", pictures of a stream that do not refer to each other in one batch (%d batches per %d pictures)" % (period, npic)
                                  if args.group_by_level else ""),
                   "streams_per_gpu": S, "contexts_per_gpu": C, "frames_per_stream": npic, "clips": len(seqs), "staggered": bool(args.stagger), "group_by_level": bool(args.group_by_level),
                   "upload_ring": bool(args.upload_ring), "md5_exact": True, "md5_checked_pictures": checked if not args.check_two_streams_only else 2 * npic},
        "recon_alg_gbps": round(sum(alg_bytes_stream) * args.steps / elapsed / 1e9, 2),
        "recon_alg_gbps_gpu_time": round(sum(alg_bytes_stream[s] for s in groups[0]) / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else None,
        "alg_bytes_per_frame": sum(alg_bytes_stream) // (npic * S),
        "roofline": {"bound": bound, "valu": valu, "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                     "alg_bytes_per_launch": int(kb[dom] / max(dom_launches, 1)),
                     "launches_per_step": int(dom_launches), "avg_launch_us": round(dom_ms * 1e3 / max(dom_launches, 1), 3),
                     "pictures_per_launch": round(S_prof * npic / period, 1), "timed_in": "a pass of one context alone (HIP events on its stream)"},
        "kernels": kernels,
        "packer_us_per_picture": round(W.pack_s / max(W.pack_n, 1) * 1e6, 1),
    }

    if extras and world == 1:
        # ---- syntax hand-over (never `value`): the packed pictures cross the host link again ----
        # Two passes in which every picture is marked for upload before its batch is submitted: the engine copies a
        # batch's pictures from their pinned buffers on its copy streams, beside the previous batch's kernels, and
        # expands the run-length coded records.  (Packing itself, host work of the decoder, is not repeated.)
        progress("side measurements: hand-over")
        if "handover" in extras:
            up_bytes = sum(W.syns[t][s].stats()[29] for t in range(npic) for s in range(S))
            sync()
            t0 = time.perf_counter()
            for gg in range(g, g + 2 * period):
                W.step(gg, upload=True)
            t_host = (time.perf_counter() - t0) / 2  # the submitting thread's share: it runs ahead of the device until the rings are full
            sync()
            t_both = (time.perf_counter() - t0) / 2
            g += 2 * period
            result["handover"] = {"bytes_per_frame": up_bytes // (npic * S), "with_reconstruction_frames_per_s": round(npic * S / t_both, 1),
                                  "link_gbps": round(up_bytes / t_both / 1e9, 2), "host_submit_share": round(t_host / t_both, 3)}
        # ---- fewer streams: 1 and 8 streams of context 0 alone (what one decoder process, or 64 streams over 8 GPUs
        # = 8 per GPU, get out of the device), and all streams each at its own position ----
        progress("side measurements: fewer streams")
        side = {}
        if "fewer" in extras:
            # What a few streams get out of the device (one decoder process; 64 streams over 8 GPUs = 8 per GPU): in
            # decoding order, one picture of every stream per batch; and by dependency level, the pictures of a stream
            # that do not refer to each other in one batch (level_schedule).  8 streams also over two and four contexts
            # (one context's launch latency and intra chain beside the others' kernels).
            def few(n_str, n_ctx, by_level):
                Wf = Workload(seqs, n_str, n_ctx, local_rank, args.upload_ring, False, by_level)
                fps = Wf.timed_passes(2)
                Wf.close()
                return round(fps, 1)

            for n_small in (1, 8):
                if n_small <= S:
                    side["streams_%d_frames_per_s" % n_small] = few(n_small, 1, False)
                    side["streams_%d_by_level_frames_per_s" % n_small] = few(n_small, 1, True)
            if S >= 8:
                side["streams_8_over_two_contexts_frames_per_s"] = few(8, 2, False)
                side["streams_8_over_four_contexts_frames_per_s"] = few(8, 4, False)
                side["streams_8_by_level_over_two_contexts_frames_per_s"] = few(8, 2, True)
            side["batches_per_pass_by_level"] = len(level_schedule(seqs[0]))
        progress("side measurements: every stream at its own position")
        if not args.stagger and S >= 2 * C and "stagger" in extras:
            W.set_offsets(True)
            side["every_stream_at_its_own_position_frames_per_s"] = round(W.timed_passes(2), 1)
            if C != 4 and S >= 8:
                # the same over four contexts: with an I picture in every batch the intra kernel's dependency chains are
                # what a batch waits for, and four batches in flight hide more of them than two
                W.close()
                W = None
                W4 = Workload(seqs, S, 4, local_rank, args.upload_ring, True, args.group_by_level)
                side["every_stream_at_its_own_position_over_four_contexts_frames_per_s"] = round(W4.timed_passes(2), 1)
                W4.close()
        result["fewer_streams"] = side
        if e2e is not None:
            result["end_to_end"] = e2e
            if "packer_us_per_picture_c" in e2e:
                result["packer_us_per_picture_c"] = e2e.pop("packer_us_per_picture_c")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("CPU baseline")
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if W is not None:
        W.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
