"""Times hevcrecon_picture_md5_batch on resident 1080p pictures: the call's duration for batches of 1 .. 4096 pictures."""
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import load_golden  # noqa: E402
from hevc_amd import engine  # noqa: E402

seq = load_golden("randomaccess_main_1920x1080", 4)
eng = engine.Engine(seq.seq_params)
syn = eng.new_syntax()
for pic in seq.pictures:
    syn.pack(pic)
    eng.reconstruct(syn, pic.pic_id)
want = [p.stage_md5[2] for p in seq.pictures]
ids = [p.pic_id for p in seq.pictures]
for n in (1, 64, 256, 1024, 4096):
    batch = (ids * 1024)[:n]
    eng.md5(batch)
    t = time.perf_counter()
    got = eng.md5(batch)
    dt = time.perf_counter() - t
    assert got == (want * 1024)[:n]
    print("md5 of %3d pictures: %.1f ms, %.0f pictures/s" % (n, dt * 1e3, n / dt), flush=True)
