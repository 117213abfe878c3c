"""ctypes mirror of include/hevcrecon.h (the engine's C ABI).

Only plumbing lives here: struct layouts, library loading and the argument
types of every exported entry point.  The library is the hand-written HIP
engine built by hevc_amd/csrc/build.sh; there is no Python or CPU fallback:
if the shared object is missing, loading raises.
"""
import ctypes as C
import os

MAX_REFS = 16
PARTS_PER_CTU = 256
NUM_KERNELS = 11

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEVCRECON_LIB names another build of the library (diagnostic variants); the default is the in-tree one
LIB_PATH = os.environ.get("HEVCRECON_LIB") or os.path.join(_HERE, "libhevcrecon.so")


class SeqParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("bit_depth_luma", C.c_int32),
        ("bit_depth_chroma", C.c_int32),
        ("ctu_size", C.c_int32),
        ("max_cu_depth", C.c_int32),
        ("strong_intra_smoothing", C.c_int32),
        ("pcm_bit_depth_luma", C.c_int32),
        ("pcm_bit_depth_chroma", C.c_int32),
        ("pcm_loop_filter_disable", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


class PicParams(C.Structure):
    _fields_ = [
        ("poc", C.c_int32),
        ("slice_type", C.c_int32),
        ("deblocking_disable", C.c_int32),
        ("beta_offset_div2", C.c_int32),
        ("tc_offset_div2", C.c_int32),
        ("cb_qp_offset_pps", C.c_int32),
        ("cr_qp_offset_pps", C.c_int32),
        ("cb_qp_delta_slice", C.c_int32),
        ("cr_qp_delta_slice", C.c_int32),
        ("sao_luma", C.c_int32),
        ("sao_chroma", C.c_int32),
        ("num_ref", C.c_int32 * 2),
        ("ref_pic_id", (C.c_int32 * MAX_REFS) * 2),
        ("ref_poc", (C.c_int32 * MAX_REFS) * 2),
        ("constrained_intra_pred", C.c_int32),
        ("no_loop_filter_across_tiles", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


class SliceParams(C.Structure):
    _fields_ = [
        ("start_addr", C.c_uint32),
        ("deblocking_disable", C.c_int32),
        ("beta_offset_div2", C.c_int32),
        ("tc_offset_div2", C.c_int32),
        ("cb_qp_delta_slice", C.c_int32),
        ("cr_qp_delta_slice", C.c_int32),
        ("loop_filter_across_slices", C.c_int32),
        ("reserved", C.c_int32),
    ]


# size of the structure in dumps of format 1 (before constrained_intra_pred + reserved were appended)
PIC_PARAMS_SIZE_V1 = (13 + 2 * MAX_REFS * 2) * 4


class CtuSyntax(C.Structure):
    _fields_ = [
        ("depth", C.c_void_p),
        ("pred_mode", C.c_void_p),
        ("part_size", C.c_void_p),
        ("tr_idx", C.c_void_p),
        ("cbf", C.c_void_p * 3),
        ("qp", C.c_void_p),
        ("luma_dir", C.c_void_p),
        ("chroma_dir", C.c_void_p),
        ("transform_skip", C.c_void_p * 3),
        ("mv", C.c_void_p * 2),
        ("ref_idx", C.c_void_p * 2),
        ("transquant_bypass", C.c_void_p),
        ("ipcm", C.c_void_p),
        ("coeff", C.c_void_p * 3),
        ("sao_type", C.c_int32 * 3),
        ("sao_subtype", C.c_int32 * 3),
        ("sao_offset", (C.c_int32 * 4) * 3),
        ("pcm", C.c_void_p * 3),
        ("slice_start", C.c_void_p),
        ("coding_addr_plus1", C.c_int32),
        ("tile_id", C.c_int32),
    ]


class ScalingLists(C.Structure):
    _fields_ = [
        ("m4", (C.c_uint8 * 16) * 6),
        ("m8", (C.c_uint8 * 64) * 6),
        ("m16", (C.c_uint8 * 64) * 6),
        ("m32", (C.c_uint8 * 64) * 2),
        ("dc16", C.c_uint8 * 6),
        ("dc32", C.c_uint8 * 2),
    ]


class PredWeights(C.Structure):
    _fields_ = [
        ("log2_denom", ((C.c_int16 * 3) * MAX_REFS) * 2),
        ("weight", ((C.c_int16 * 3) * MAX_REFS) * 2),
        ("offset", ((C.c_int16 * 3) * MAX_REFS) * 2),
    ]


# name -> (restype, argtypes); every symbol include/hevcrecon.h declares
_P = C.c_void_p
_I16P = C.c_void_p
EXPORTS = {
    "hevcrecon_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "hevcrecon_destroy": (None, [_P]),
    "hevcrecon_set_upload_ring": (C.c_int, [_P, C.c_int64]),
    "hevcrecon_last_error": (C.c_char_p, [_P]),
    "hevcrecon_syntax_last_error": (C.c_char_p, [_P]),
    "hevcrecon_sequence_init": (C.c_int, [_P, C.POINTER(SeqParams)]),
    "hevcrecon_syntax_create": (C.c_int, [_P, C.POINTER(_P)]),
    "hevcrecon_syntax_create_detached": (C.c_int, [C.POINTER(SeqParams), C.POINTER(_P)]),
    "hevcrecon_syntax_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "hevcrecon_syntax_destroy": (None, [_P]),
    "hevcrecon_syntax_begin": (C.c_int, [_P]),
    "hevcrecon_syntax_set_scaling_lists": (C.c_int, [_P, C.c_void_p]),
    "hevcrecon_syntax_set_pred_weights": (C.c_int, [_P, C.c_void_p]),
    "hevcrecon_syntax_set_slices": (C.c_int, [_P, C.c_int, C.c_void_p]),
    "hevcrecon_syntax_add_ctu": (C.c_int, [_P, C.c_int, C.POINTER(CtuSyntax)]),
    "hevcrecon_syntax_finish": (C.c_int, [_P, C.POINTER(PicParams)]),
    "hevcrecon_syntax_upload": (C.c_int, [_P]),
    "hevcrecon_syntax_upload_batch": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "hevcrecon_reconstruct": (C.c_int, [_P, _P, C.c_int]),
    "hevcrecon_reconstruct_noalloc": (C.c_int, [_P, _P, C.c_int]),
    "hevcrecon_reconstruct_batch": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int)]),
    "hevcrecon_picture_download": (C.c_int, [_P, C.c_int, _I16P, _I16P, _I16P, C.c_int, C.c_int]),
    "hevcrecon_picture_download_file": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "hevcrecon_picture_md5_batch": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "hevcrecon_syntax_export": (C.c_int, [_P, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "hevcrecon_syntax_import": (C.c_int, [_P, C.c_void_p, C.c_int64]),
    "hevcrecon_hub_connect": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hevcrecon_hub_pic_id_base": (C.c_int, [_P]),
    "hevcrecon_hub_submit": (C.c_int, [_P, _P, C.c_int]),
    "hevcrecon_hub_copy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "hevcrecon_hub_poll": (C.c_int, [_P, C.POINTER(C.c_int), C.c_void_p]),
    "hevcrecon_hub_wait": (C.c_int, [_P, C.c_void_p]),
    "hevcrecon_hub_error": (C.c_char_p, [_P]),
    "hevcrecon_hub_disconnect": (None, [_P]),
    "hevcrecon_picture_md5_submit": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "hevcrecon_picture_md5_submit_in_place": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "hevcrecon_picture_md5_poll": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int), C.c_void_p]),
    "hevcrecon_picture_md5_wait": (C.c_int, [_P, C.c_int64, C.c_void_p]),
    "hevcrecon_debug_keep_stages": (C.c_int, [_P, C.c_int]),
    "hevcrecon_debug_download_stage": (C.c_int, [_P, C.c_int, _I16P, _I16P, _I16P, C.c_int, C.c_int]),
    "hevcrecon_picture_upload": (C.c_int, [_P, C.c_int, C.c_int, _I16P, _I16P, _I16P, C.c_int, C.c_int]),
    "hevcrecon_picture_copy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "hevcrecon_picture_release": (C.c_int, [_P, C.c_int]),
    "hevcrecon_synchronize": (C.c_int, [_P]),
    "hevcrecon_profile_enable": (C.c_int, [_P, C.c_int]),
    "hevcrecon_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "hevcrecon_kernel_name": (C.c_char_p, [C.c_int]),
    "hevcrecon_syntax_algorithmic_bytes": (C.c_int64, [_P]),
}

_lib = None


def load_library(path=None):
    """Load libhevcrecon.so and type every export.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "hevc_amd: %s not found - build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()')" % p
        )
    lib = C.CDLL(p)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
