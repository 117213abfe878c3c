"""Thin object wrapper over the engine's C ABI (include/hevcrecon.h), for tests and the benchmark.

Method names follow the C entry points one for one.  All compute happens in
libhevcrecon.so (HIP, gfx950); nothing here touches samples except to copy
them in and out of numpy arrays.
"""
import ctypes as C
import time

import numpy as np

from . import abi


class EngineError(RuntimeError):
    pass


class Syntax:
    """One hevcrecon_syntax object: packs a PictureSyntax and keeps its copy in HBM."""

    def __init__(self, engine, handle):
        self.engine = engine
        self.lib = engine.lib
        self.h = handle
        self.poc = None
        self.pack_seconds = 0.0

    def pack(self, pic):
        """begin + add_ctu for every CTU + finish (which queues the upload)."""
        lib = self.lib
        self.engine._check(lib.hevcrecon_syntax_begin(self.h), "hevcrecon_syntax_begin")
        if getattr(pic, "scaling", None) is not None:
            self.engine._check(lib.hevcrecon_syntax_set_scaling_lists(self.h, C.addressof(pic.scaling)), "hevcrecon_syntax_set_scaling_lists")
        if getattr(pic, "pred_weights", None) is not None:
            self.engine._check(lib.hevcrecon_syntax_set_pred_weights(self.h, C.addressof(pic.pred_weights)), "hevcrecon_syntax_set_pred_weights")
        if getattr(pic, "slices", None) is not None:
            self.engine._check(lib.hevcrecon_syntax_set_slices(self.h, len(pic.slices), C.addressof(pic.slices)), "hevcrecon_syntax_set_slices")
        ctus = pic.ctu_structs()
        order = range(pic.nctu)
        if getattr(pic, "tiles", None) is not None:  # CTUs go in coding order (tile scan)
            order = sorted(range(pic.nctu), key=lambda a: int(pic.tiles[a, 0]))
        add, check = lib.hevcrecon_syntax_add_ctu, self.engine._check
        t0 = time.perf_counter()
        for a in order:
            rc = add(self.h, a, C.byref(ctus[a]))
            if rc:
                check(rc, "hevcrecon_syntax_add_ctu")
        check(lib.hevcrecon_syntax_finish(self.h, C.byref(pic.pic_params)), "hevcrecon_syntax_finish")
        # host time of packing this picture: the add_ctu calls and finish (with the foreign-function overhead of one
        # call per CTU, about a microsecond each)
        self.pack_seconds = time.perf_counter() - t0
        self.poc = pic.poc
        return self

    def stats(self):
        out = (C.c_int64 * 32)()
        self.lib.hevcrecon_syntax_stats(self.h, out)
        return list(out)

    def upload(self):
        """Queue the host-to-device copy of the packed picture again (pack() queued it once)."""
        self.engine._check(self.lib.hevcrecon_syntax_upload(self.h), "hevcrecon_syntax_upload")

    def algorithmic_bytes(self):
        return int(self.lib.hevcrecon_syntax_algorithmic_bytes(self.h))

    def close(self):
        if self.h:
            self.lib.hevcrecon_syntax_destroy(self.h)
            self.h = None


class Engine:
    def __init__(self, seq_params, device=0, lib=None):
        self.lib = lib if lib is not None else abi.load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.hevcrecon_create(C.byref(self.ctx), device)
        if rc:
            raise EngineError("hevcrecon_create failed (%d): no usable GPU" % rc)
        self.sp = seq_params
        self._check(self.lib.hevcrecon_sequence_init(self.ctx, C.byref(seq_params)), "hevcrecon_sequence_init")
        self.w, self.h = seq_params.width, seq_params.height

    def _check(self, rc, what):
        if rc:
            msg = self.lib.hevcrecon_last_error(self.ctx)
            raise EngineError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def close(self):
        if self.ctx:
            self.lib.hevcrecon_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def new_syntax(self):
        h = C.c_void_p()
        self._check(self.lib.hevcrecon_syntax_create(self.ctx, C.byref(h)), "hevcrecon_syntax_create")
        return Syntax(self, h)

    def reconstruct(self, syn, dst_pic_id):
        self._check(self.lib.hevcrecon_reconstruct(self.ctx, syn.h, dst_pic_id), "hevcrecon_reconstruct")

    def reconstruct_noalloc(self, syn, dst_pic_id):
        """hevcrecon_reconstruct for callers that cannot allocate (the destination picture must exist already)."""
        self._check(self.lib.hevcrecon_reconstruct_noalloc(self.ctx, syn.h, dst_pic_id), "hevcrecon_reconstruct_noalloc")

    def reconstruct_batch(self, syns, dst_pic_ids):
        n = len(syns)
        hs = (C.c_void_p * n)(*[s.h for s in syns])
        ids = (C.c_int * n)(*dst_pic_ids)
        self._check(self.lib.hevcrecon_reconstruct_batch(self.ctx, n, hs, ids), "hevcrecon_reconstruct_batch")

    def prepare_batch(self, syns, dst_pic_ids):
        """Argument arrays of a reconstruct_batch call, built once for a batch that is submitted repeatedly."""
        n = len(syns)
        return (n, (C.c_void_p * n)(*[s.h for s in syns]), (C.c_int * n)(*dst_pic_ids))

    def upload_prepared(self, prepared):
        """Mark every picture of a prepared batch for another trip to the device."""
        n, hs, _ = prepared
        self._check(self.lib.hevcrecon_syntax_upload_batch(n, hs), "hevcrecon_syntax_upload_batch")

    def reconstruct_prepared(self, prepared):
        n, hs, ids = prepared
        self._check(self.lib.hevcrecon_reconstruct_batch(self.ctx, n, hs, ids), "hevcrecon_reconstruct_batch")

    def _planes(self):
        return [np.empty((self.h, self.w), np.int16), np.empty((self.h // 2, self.w // 2), np.int16),
                np.empty((self.h // 2, self.w // 2), np.int16)]

    def download(self, pic_id):
        p = self._planes()
        self._check(self.lib.hevcrecon_picture_download(self.ctx, pic_id, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data,
                                                        self.w, self.w // 2), "hevcrecon_picture_download")
        return p

    def download_file(self, pic_id, file_bit_depths, width=None, height=None):
        """The picture as bytes of the reference decoder's output file (see hevcrecon_picture_download_file)."""
        w, h = width or self.w, height or self.h
        wide = max(file_bit_depths) > 8
        out = np.empty((w * h + 2 * (w // 2) * (h // 2)) * (2 if wide else 1), np.uint8)
        self._check(self.lib.hevcrecon_picture_download_file(self.ctx, pic_id, file_bit_depths[0], file_bit_depths[1], w, h,
                                                             out.ctypes.data, out.nbytes), "hevcrecon_picture_download_file")
        return out

    def set_upload_ring(self, nbytes):
        self._check(self.lib.hevcrecon_set_upload_ring(self.ctx, int(nbytes)), "hevcrecon_set_upload_ring")

    def md5(self, pic_ids):
        """Per picture the three plane digests (hex), computed on the device."""
        n = len(pic_ids)
        out = np.empty(48 * n, np.uint8)
        self._check(self.lib.hevcrecon_picture_md5_batch(self.ctx, n, (C.c_int * n)(*pic_ids), out.ctypes.data), "hevcrecon_picture_md5_batch")
        return [[out[48 * i + 16 * c:48 * i + 16 * c + 16].tobytes().hex() for c in range(3)] for i in range(n)]

    def md5_submit(self, pic_ids):
        """Queue the hash of these pictures beside later reconstruction; returns a ticket for md5_poll / md5_wait."""
        n = len(pic_ids)
        t = C.c_int64(-1)
        self._check(self.lib.hevcrecon_picture_md5_submit(self.ctx, n, (C.c_int * n)(*pic_ids), C.byref(t)), "hevcrecon_picture_md5_submit")
        return (t.value, n)

    @staticmethod
    def _digests(out, n):
        return [[out[48 * i + 16 * c:48 * i + 16 * c + 16].tobytes().hex() for c in range(3)] for i in range(n)]

    def md5_poll(self, ticket):
        """None while the digests are not back yet, else the digests (the ticket is then spent)."""
        t, n = ticket
        out = np.empty(48 * n, np.uint8)
        done = C.c_int(0)
        self._check(self.lib.hevcrecon_picture_md5_poll(self.ctx, t, C.byref(done), out.ctypes.data), "hevcrecon_picture_md5_poll")
        return self._digests(out, n) if done.value else None

    def md5_wait(self, ticket):
        t, n = ticket
        out = np.empty(48 * n, np.uint8)
        self._check(self.lib.hevcrecon_picture_md5_wait(self.ctx, t, out.ctypes.data), "hevcrecon_picture_md5_wait")
        return self._digests(out, n)

    def keep_stages(self, on=True):
        self._check(self.lib.hevcrecon_debug_keep_stages(self.ctx, 1 if on else 0), "hevcrecon_debug_keep_stages")

    def download_stage(self, stage):
        p = self._planes()
        self._check(self.lib.hevcrecon_debug_download_stage(self.ctx, stage, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data,
                                                            self.w, self.w // 2), "hevcrecon_debug_download_stage")
        return p

    def upload(self, pic_id, poc, planes):
        p = [np.ascontiguousarray(x, dtype=np.int16) for x in planes]
        self._check(self.lib.hevcrecon_picture_upload(self.ctx, pic_id, poc, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data,
                                                      p[0].shape[1], p[1].shape[1]), "hevcrecon_picture_upload")

    def copy_picture(self, dst_pic_id, src_pic_id, dst_poc):
        self._check(self.lib.hevcrecon_picture_copy(self.ctx, dst_pic_id, src_pic_id, dst_poc), "hevcrecon_picture_copy")

    def release(self, pic_id):
        self._check(self.lib.hevcrecon_picture_release(self.ctx, pic_id), "hevcrecon_picture_release")

    def synchronize(self):
        self._check(self.lib.hevcrecon_synchronize(self.ctx), "hevcrecon_synchronize")

    def profile_enable(self, on=True):
        self._check(self.lib.hevcrecon_profile_enable(self.ctx, 1 if on else 0), "hevcrecon_profile_enable")

    def profile_read(self):
        ms = (C.c_double * abi.NUM_KERNELS)()
        n = (C.c_int64 * abi.NUM_KERNELS)()
        self._check(self.lib.hevcrecon_profile_read(self.ctx, ms, n), "hevcrecon_profile_read")
        return {self.lib.hevcrecon_kernel_name(i).decode(): (ms[i], n[i]) for i in range(abi.NUM_KERNELS)}


def detached_syntax(seq_params, lib=None):
    """Host-only packer object (no GPU needed): for host-logic tests."""
    lib = lib if lib is not None else abi.load_library()
    h = C.c_void_p()
    rc = lib.hevcrecon_syntax_create_detached(C.byref(seq_params), C.byref(h))
    if rc:
        raise EngineError("hevcrecon_syntax_create_detached failed (%d)" % rc)

    class _Host:
        pass

    host = _Host()
    host.lib = lib
    host._check = lambda rc, what: (_ for _ in ()).throw(EngineError("%s failed (%d)" % (what, rc))) if rc else None
    return Syntax(host, h)
