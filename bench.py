#!/usr/bin/env python3
"""Benchmark of the picture-reconstruction engine on MI355X.

Workload: the 1920x1080, 60-frame random-access stream encoded by the reference
encoder (cfg/encoder_randomaccess_main.cfg) from the synthetic clip of tools/gen_clip.py, as parsed by the
reference decoder (tests/golden/randomaccess_main_1920x1080.syn.xz).  Each GPU reconstructs `--streams`
independent copies of that stream (the stream-sharded multi-stream case), pictures of equal decoding
position batched into one set of launches.  A step = one pass over all 60 pictures of every stream.
The packed syntax is resident in HBM before the timed region starts; every final picture is MD5-checked
against the reference decoder's digest after it.

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


def clone_for_stream(pic, s):
    p = copy.copy(pic)
    p.pic_params = type(pic.pic_params).from_buffer_copy(pic.pic_params)
    for l in range(2):
        for r in range(p.pic_params.num_ref[l]):
            p.pic_params.ref_pic_id[l][r] = pic.pic_params.ref_pic_id[l][r] + s * ID_STRIDE
    return p


def kernel_bytes(seq, stats_list, sao_flags):
    """Algorithmic bytes per kernel for a list of pictures: the terms of
    S*b*(1 store + 2 deblock + 2 SAO) + MC reference reads + 2*coefficients + 16 per 4x4 record (or 16 per
    uniform-motion region), attributed to the kernel that moves them.  Luma + chroma = 1.5 bytes per luma
    sample at 8 bits."""
    sp = seq.seq_params
    b = 2 if sp.bit_depth_luma > 8 else 1
    s_bytes = sp.width * sp.height * 3 // 2 * b
    names = ["k_inter_tile64", "k_inter_tile32", "k_inter_tile16", "k_inter_blk", "k_transform", "k_intra", "k_deblock_v",
             "k_deblock_h", "k_sao"]
    out = dict.fromkeys(names, 0)
    for st, sao in zip(stats_list, sao_flags):
        n4, intra4 = st[8], st[9]
        tile_samples = tile_ref_samples = 0
        for t, name in enumerate(names[:3]):
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


def cpu_baseline(seconds_target=12.0):
    """Reference decoder (HM 9.1 TAppDecoder, -O3, one thread) on the same bitstream, host cores of this machine."""
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder")
    bits = os.path.join(ROOT, "tests", "golden", "randomaccess_main_1080.bin")
    if os.path.exists(exe) and os.access(exe, os.X_OK) and os.path.exists(bits):
        frames, t0 = 0, time.time()
        while time.time() - t0 < seconds_target:
            out = subprocess.run([exe, "-b", bits], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode("latin1")
            n = out.count("(OK)")
            if n != 60:
                raise RuntimeError("reference decoder did not report 60 matching pictures")
            frames += n
        dt = time.time() - t0
        return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": 1, "kind": "reference",
                "sample": "%d runs of the 60-frame 1080p RA stream through oracle/_ref/TAppDecoder (parse + reconstruction + MD5 check), %.1f s" % (frames // 60, dt)}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=128,
                    help="independent streams per GPU; larger batches amortise the latency-bound intra launches "
                         "(measured: 64 -> 72.9 k, 128 -> 81.6 k, 256 -> 86.4 k frames/s)")
    ap.add_argument("--contexts", type=int, default=2,
                    help="engine contexts (HIP streams) the streams of this GPU are spread over; with several, one context's "
                         "latency-bound intra launches run beside the others' throughput-bound kernels "
                         "(measured at 128 streams: 1 -> 81.9 k, 2 -> 88.8 k, 3 -> 91.6 k, 4 -> 92.5 k frames/s; beyond two the "
                         "launches get so small that the latency-bound intra kernel tops the per-kernel times, and the "
                         "hand-over of fresh pictures suffers from queue sharing)")
    ap.add_argument("--fixture", default="randomaccess_main_1920x1080")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--merge", action="store_true",
                    help="batch independent pictures of a stream (hierarchical-B siblings) together: 33 instead of 60 batches "
                         "per pass of the RA stream; measured +0.4 %% with one context, -2 %% with two")
    ap.add_argument("--no-handover", action="store_true", help="skip the passes that time the syntax hand-over")
    ap.add_argument("--upload-ring", action="store_true", help="stage all packed pictures in the context's upload ring (coalesced hand-over copies)")
    ap.add_argument("--check-two-streams-only", action="store_true",
                    help="parity pass without the device hash of every stream (profiling runs: keeps k_md5 out of the kernel statistics)")
    ap.add_argument("--rehearse", action="store_true", help="no GPU work: only the multi-rank plumbing (used by the CPU tests)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="streams of the whole job, dealt to the GPUs round-robin (stream i -> rank i mod N); overrides --streams: "
                         "64 on 8 GPUs is 8 independent streams per GPU")
    args = ap.parse_args()

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
    seq = syntax.read_dump(os.path.join(ROOT, "tests", "golden", args.fixture + ".syn.xz"))
    sp = seq.seq_params
    npic, S = len(seq.pictures), args.streams
    C = max(1, min(args.contexts, S))
    engs = [engine.Engine(sp, device=local_rank) for _ in range(C)]
    eng = engs[0]
    ctx_of = [s * C // S for s in range(S)]  # stream -> context
    # Every packed picture stays alive here.  With --upload-ring the ring has room for all of them, so the hand-over
    # passes copy a batch in a few pieces (measured: hand-over 53 k -> 69 k frames/s, resident rate 88.9 k -> 87.1 k
    # with two contexts); without, every picture stages in buffers of its own.
    for c in range(C):
        engs[c].set_upload_ring(min(4 << 30, ctx_of.count(c) * npic * (256 << 10)) if args.upload_ring else 0)

    # ---- pack every picture of every stream and make it resident in HBM (untimed) ----
    syns = [[None] * S for _ in range(npic)]
    stats, sao_flags = [], []
    for t, pic in enumerate(seq.pictures):
        pic.ctu_structs()
        for s in range(S):
            syns[t][s] = engs[ctx_of[s]].new_syntax().pack(clone_for_stream(pic, s))
        stats.append(syns[t][0].stats())
        sao_flags.append(bool(pic.pic_params.sao_luma or pic.pic_params.sao_chroma))
        pic.drop_ctypes()
    ids = [[pic.pic_id + s * ID_STRIDE for s in range(S)] for pic in seq.pictures]
    alg_bytes_stream = sum(syns[t][0].algorithmic_bytes() for t in range(npic))
    groups = [[s for s in range(S) if ctx_of[s] == c] for c in range(C)]
    for e in engs:
        e.synchronize()

    # Batches: by default one picture position of all streams per set of launches.  With --merge, pictures of one
    # stream that do not depend on each other (hierarchical-B siblings) share a batch: a picture's level is one more
    # than the highest level among the pictures it reads, the previous writer of its buffer and that buffer's readers.
    level, last_writer, readers = [], {}, {}
    for t, pic in enumerate(seq.pictures):
        pp = pic.pic_params
        refs = {pp.ref_pic_id[l][r] for l in range(2) for r in range(pp.num_ref[l])}
        lv = max([level[last_writer[r]] + 1 for r in refs] + [0])
        if pic.pic_id in last_writer:
            lv = max([lv, level[last_writer[pic.pic_id]] + 1] + [level[r] + 1 for r in readers[pic.pic_id]])
        level.append(lv)
        for r in refs:
            readers[r].append(t)
        last_writer[pic.pic_id], readers[pic.pic_id] = t, []
    if not args.merge:
        level = list(range(npic))
    nb = max(level) + 1
    members = [[t for t in range(npic) if level[t] == b] for b in range(nb)]
    prepared = [[engs[c].prepare_batch([syns[t][s] for t in members[b] for s in groups[c]],
                                       [ids[t][s] for t in members[b] for s in groups[c]]) for c in range(C)]
                for b in range(nb)]

    # Contexts run on their own HIP streams, so their kernels overlap on the GPU.  Independent streams are not in
    # lock step in practice: context c works `phase[c]` pictures ahead (cyclically; the pictures a position refers
    # to are resident from the previous pass), which lets one context's latency-bound intra waves run beside
    # another's throughput-bound kernels.
    phase = [c * nb // C for c in range(C)]

    def one_step():
        for b in range(nb):
            for c in range(C):
                engs[c].reconstruct_prepared(prepared[(b + phase[c]) % nb][c])

    def sync():
        for e in engs:
            e.synchronize()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()

    if C > 1:  # one pass in decoding order makes every picture resident before the phase-shifted passes start
        for b in range(nb):
            for c in range(C):
                engs[c].reconstruct_prepared(prepared[b][c])
    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, device="cuda" if backend == "nccl" else "cpu")

    # ---- parity: one more (untimed) pass, every final picture against the reference decoder's MD5 right after its
    # batch (the decoder recycles picture ids, so the check cannot wait): the first and last stream downloaded and
    # hashed on the host, and every stream hashed on the device (skipped with --check-two-streams-only)
    bd = [sp.bit_depth_luma, sp.bit_depth_chroma, sp.bit_depth_chroma]
    for b in range(nb):
        for c in range(C):
            engs[c].reconstruct_prepared(prepared[b][c])
        for t in members[b]:
            pic = seq.pictures[t]
            for s in sorted({0, S - 1}):
                planes = engs[ctx_of[s]].download(ids[t][s])
                got = [syntax.plane_md5(planes[c], bd[c]) for c in range(3)]
                if got != pic.stage_md5[2]:
                    raise SystemExit("bench: stream %d POC %d differs from the reference decoder's MD5" % (s, pic.poc))
            for c in range(C if not args.check_two_streams_only else 0):
                for s, got in zip(groups[c], engs[c].md5([ids[t][s] for s in groups[c]])):
                    if got != pic.stage_md5[2]:
                        raise SystemExit("bench: stream %d POC %d differs from the reference decoder's MD5 (device hash)" % (s, pic.poc))

    # ---- per-kernel time with HIP events on the engine's stream (one extra, untimed step) ----
    # (first context only, scaled by nothing: with several contexts the kernels of different contexts overlap,
    # so per-kernel times are taken from a single-context pass over that context's streams)
    eng.profile_enable(True)
    for b in range(nb):
        eng.reconstruct_prepared(prepared[b][0])
    prof = eng.profile_read()
    eng.profile_enable(False)
    S_prof = len(groups[0])
    kb = kernel_bytes(seq, stats, sao_flags)
    kernels = {}
    for name, nbytes in kb.items():
        ms, launches = prof[name]
        kernels[name] = {"ms_per_step": round(ms, 4), "launches": int(launches),
                         "alg_gbps": round(nbytes * S_prof / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}
    dom = max(kb, key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    achieved = kb[dom] * S_prof / (dom_ms * 1e-3) / 1e9
    gpu_ms = sum(prof[k][0] for k in kb)

    # HBM traffic of the dominant kernel: not measurable from inside the process; taken from the committed PMC
    # passes (tools/pmc_passes.sh -> tools/pmc_traffic.py) when they were collected on this very configuration
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            tj = json.load(f)
        if tj.get("streams") == S_prof and tj.get("fixture") == args.fixture and dom in tj.get("kernels", {}):
            traffic = tj["kernels"][dom]["bytes_per_launch"]
    except (OSError, ValueError):
        pass

    # ---- syntax hand-over (untimed for `value`): the packed pictures cross the host link again ----
    # Two steps in which every picture is marked for upload before its batch is submitted: the engine copies a
    # batch's pictures from their pinned buffers on its copy streams, beside the previous batch's kernels, and
    # expands the run-length coded records.  (Packing itself, host work of the decoder, is not repeated.)
    handover = None
    if not args.no_handover:
        up_bytes = sum(syns[t][0].stats()[29] for t in range(npic))
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            for b in range(nb):
                for c in range(C):
                    engs[c].upload_prepared(prepared[b][c])
                    engs[c].reconstruct_prepared(prepared[b][c])
        sync()
        t_both = (time.perf_counter() - t0) / 2
        handover = {"bytes_per_frame": up_bytes // npic, "with_reconstruction_frames_per_s": round(npic * S / t_both, 1),
                    "link_gbps": round(up_bytes * S / t_both / 1e9, 2)}

    frames_total = sharding.sum_over_ranks(npic * S * args.steps, dist, device="cuda" if backend == "nccl" else "cpu")
    value = frames_total / elapsed
    result = {
        "metric": "1080p_ra_reconstructed_frames_per_s",
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
        "config": {"workload": "%dx%d %d-frame RA stream (encoder_randomaccess_main.cfg), %d independent streams per GPU, "
                               "stream-sharded, no collectives" % (sp.width, sp.height, npic, S),
                   "streams_per_gpu": S, "contexts_per_gpu": C, "frames_per_stream": npic, "batches_per_stream_pass": nb,
                   "upload_ring": bool(args.upload_ring), "md5_exact": True, "md5_checked_streams": min(S, 2) if args.check_two_streams_only else S},
        "recon_alg_gbps": round(alg_bytes_stream * S * args.steps / elapsed / 1e9, 2),
        "recon_alg_gbps_gpu_time": round(alg_bytes_stream * S_prof / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else None,
        "alg_bytes_per_frame": alg_bytes_stream // npic,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                     "alg_bytes_per_launch": int(kb[dom] * S_prof / max(dom_launches, 1)),
                     "launches_per_step": int(dom_launches), "avg_launch_us": round(dom_ms * 1e3 / max(dom_launches, 1), 3),
                     "pictures_per_launch": S_prof, "timed_in": "a pass of one context alone (HIP events on its stream)"},
        "kernels": kernels,
    }
    if handover is not None:
        result["handover"] = handover
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    for row in syns:
        for y in row:
            y.close()
    for e in engs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
