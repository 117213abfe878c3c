"""ctypes wrapper over oracle/_ref/libhevcrecon_oracle.so (the plain-C restatement, recon_ref.c).

TEST INFRASTRUCTURE: parity checker only.  Imported by tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hevc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libhevcrecon_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "restate"])


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    V = C.c_void_p
    lib.hevcref_create.restype = V
    lib.hevcref_destroy.argtypes = [V]
    lib.hevcref_sequence_init.argtypes = [V, C.POINTER(abi.SeqParams)]
    lib.hevcref_reconstruct.argtypes = [V, C.POINTER(abi.PicParams), C.POINTER(abi.CtuSyntax), C.c_int, C.c_int]
    lib.hevcref_download.argtypes = [V, C.c_int, C.c_int, V, V, V]
    lib.hevcref_upload.argtypes = [V, C.c_int, C.c_int, V, V, V]
    lib.hevcref_set_scaling_lists.argtypes = [V, V]
    lib.hevcref_set_pred_weights.argtypes = [V, V]
    lib.hevcref_set_slices.argtypes = [V, C.c_int, V]
    lib.hevcref_inverse_tu.argtypes = [V, C.POINTER(abi.PicParams), V, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, V]
    return lib


class RefModel:
    def __init__(self, seq_params):
        self.lib = _load()
        self.ctx = self.lib.hevcref_create()
        self.sp = seq_params
        rc = self.lib.hevcref_sequence_init(self.ctx, C.byref(seq_params))
        if rc:
            raise RuntimeError("hevcref_sequence_init failed (%d)" % rc)
        self.w, self.h = seq_params.width, seq_params.height

    def close(self):
        if self.ctx:
            self.lib.hevcref_destroy(self.ctx)
            self.ctx = None

    def reconstruct(self, pic, dst_id=None):
        sl = getattr(pic, "scaling", None)
        self.lib.hevcref_set_scaling_lists(self.ctx, C.addressof(sl) if sl is not None else None)
        wp = getattr(pic, "pred_weights", None)
        self.lib.hevcref_set_pred_weights(self.ctx, C.addressof(wp) if wp is not None else None)
        slices = getattr(pic, "slices", None)
        self.lib.hevcref_set_slices(self.ctx, len(slices) if slices is not None else 0, C.addressof(slices) if slices is not None else None)
        ctus = pic.ctu_structs()
        rc = self.lib.hevcref_reconstruct(self.ctx, C.byref(pic.pic_params), ctus, pic.nctu, pic.pic_id if dst_id is None else dst_id)
        if rc:
            raise RuntimeError("hevcref_reconstruct failed (%d)" % rc)

    def download(self, stage, pic_id=0):
        y = np.empty((self.h, self.w), np.int16)
        cb = np.empty((self.h // 2, self.w // 2), np.int16)
        cr = np.empty((self.h // 2, self.w // 2), np.int16)
        rc = self.lib.hevcref_download(self.ctx, stage, pic_id, y.ctypes.data, cb.ctypes.data, cr.ctypes.data)
        if rc:
            raise RuntimeError("hevcref_download failed (%d)" % rc)
        return [y, cb, cr]

    def upload(self, pic_id, poc, planes):
        p = [np.ascontiguousarray(x, dtype=np.int16) for x in planes]
        self.lib.hevcref_upload(self.ctx, pic_id, poc, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data)

    def inverse_tu(self, pic_params, coef, log2n, comp, qpy, use_dst, tskip):
        n = 1 << log2n
        c32 = np.ascontiguousarray(coef, dtype=np.int32).reshape(n * n)
        out = np.empty((n, n), np.int16)
        self.lib.hevcref_inverse_tu(self.ctx, C.byref(pic_params), c32.ctypes.data, log2n, comp, qpy, use_dst, tskip, out.ctypes.data)
        return out
