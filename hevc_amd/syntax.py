"""Per-picture syntax containers and the on-disk dump format.

A dump (".syn", optionally xz-compressed) is what oracle/hm_bridge.cpp writes in
"dump" mode: the sequence parameters, then for every picture in decoding order
the slice-level parameters, every CTU's parsed syntax arrays exactly as the
reference decoder holds them after parsing (TLibCommon/TComDataCU.h:122-204 of
the reference), the SAO parameters, MD5 digests of the picture before
deblocking / after deblocking / final, and optionally those three pictures.

The arrays are data, not code: they are the input of the reconstruction path
and the fixtures under tests/golden are made of them.
"""
import ctypes as C
import hashlib
import lzma
import struct

import numpy as np

from . import abi

MAGIC = b"HVSYN2\0\0"
MAGIC_V1 = b"HVSYN1\0\0"  # older dumps: shorter picture parameters (abi.PIC_PARAMS_SIZE_V1)
PICT = 0x54434950
SCAL = 0x4C414353  # optional record before a picture: the scaling lists it de-quantises with
WPRD = 0x44525057  # optional record before a picture: its explicit prediction weights
TILE = 0x454C4954  # optional record before a picture: (1 + coding-order address, tile index) of every CTU (tiles)
SLST = 0x54534C53  # optional record before a picture: slice start address of every partition (several slices)
SLPR = 0x52504C53  # optional record before a picture: its table of per-slice parameters (second header word: entries)
PCMS = 0x534D4350  # optional record before a picture: PCM sample arrays of the CTUs that hold IPCM units
_PCM_CTU_BYTES = 4 + (16 + 4 + 4) * abi.PARTS_PER_CTU * 2
P = abi.PARTS_PER_CTU

# (name, dtype, elements per partition, count) in file order
_U8_FIELDS = [
    ("depth", np.uint8),
    ("pred_mode", np.int8),
    ("part_size", np.int8),
    ("tr_idx", np.uint8),
    ("cbf0", np.uint8),
    ("cbf1", np.uint8),
    ("cbf2", np.uint8),
    ("qp", np.int8),
    ("luma_dir", np.uint8),
    ("chroma_dir", np.uint8),
    ("tskip0", np.uint8),
    ("tskip1", np.uint8),
    ("tskip2", np.uint8),
]


class PictureSyntax:
    """Parsed syntax of one picture, as numpy arrays indexed [ctu, partition]."""

    def __init__(self):
        self.pic_params = abi.PicParams()
        self.pic_id = 0
        self.nctu = 0
        self.a = {}          # name -> array
        self.sao = None      # (nctu, 3, 6) int32: type, subtype, offset[4]
        self.scaling = None  # abi.ScalingLists when the picture uses scaling lists
        self.pred_weights = None  # abi.PredWeights when the picture predicts with explicit weights
        self.slice_start = None  # (nctu, 256) uint32 when the picture has several slices
        self.tiles = None        # (nctu, 2) int32 when the picture has tiles: 1 + coding-order address, tile index
        self.slices = None       # ctypes array of abi.SliceParams when the slices differ or do not filter across their boundaries
        self.pcm = {}        # ctu address -> (y, cb, cr) int16 arrays, for the CTUs that hold IPCM units
        self.stage_md5 = None  # (3, 3) hex strings: [stage][component]
        self.planes = None   # optional [stage][component] 2-D uint16
        self._keep = None

    @property
    def poc(self):
        return self.pic_params.poc

    def ctu_structs(self):
        """ctypes array of hevcrecon_ctu_syntax pointing into this object's arrays."""
        if self._keep is not None:
            return self._keep[0]
        a = self.a
        coeff32 = [np.ascontiguousarray(a["coeff%d" % c].astype(np.int32)) for c in range(3)]
        arr = (abi.CtuSyntax * self.nctu)()

        def ptr(x, i):
            return x[i].ctypes.data

        for i in range(self.nctu):
            s = arr[i]
            s.depth = ptr(a["depth"], i)
            s.pred_mode = ptr(a["pred_mode"], i)
            s.part_size = ptr(a["part_size"], i)
            s.tr_idx = ptr(a["tr_idx"], i)
            s.qp = ptr(a["qp"], i)
            s.luma_dir = ptr(a["luma_dir"], i)
            s.chroma_dir = ptr(a["chroma_dir"], i)
            s.transquant_bypass = ptr(a["tq_bypass"], i)
            s.ipcm = ptr(a["ipcm"], i)
            for c in range(3):
                s.cbf[c] = ptr(a["cbf%d" % c], i)
                s.transform_skip[c] = ptr(a["tskip%d" % c], i)
                s.coeff[c] = ptr(coeff32[c], i)
                s.pcm[c] = self.pcm[i][c].ctypes.data if i in self.pcm else None
                s.sao_type[c] = int(self.sao[i, c, 0])
                s.sao_subtype[c] = int(self.sao[i, c, 1])
                for k in range(4):
                    s.sao_offset[c][k] = int(self.sao[i, c, 2 + k])
            s.slice_start = self.slice_start[i].ctypes.data if self.slice_start is not None else None
            if self.tiles is not None:
                s.coding_addr_plus1, s.tile_id = int(self.tiles[i, 0]), int(self.tiles[i, 1])
            for l in range(2):
                s.mv[l] = ptr(a["mv%d" % l], i)
                s.ref_idx[l] = ptr(a["ref_idx%d" % l], i)
        self._keep = (arr, coeff32)
        return arr

    def drop_ctypes(self):
        self._keep = None


class Sequence:
    def __init__(self):
        self.seq_params = abi.SeqParams()
        self.pictures = []

    @property
    def width(self):
        return self.seq_params.width

    @property
    def height(self):
        return self.seq_params.height


def _open(path):
    return lzma.open(path, "rb") if str(path).endswith(".xz") else open(path, "rb")


def read_dump(path, max_pictures=None):
    """Read a syntax dump written by the bridge decoder (raw or .xz)."""
    seq = Sequence()
    with _open(path) as f:
        magic = f.read(8)
        if magic not in (MAGIC, MAGIC_V1):
            raise ValueError("%s: not a syntax dump" % path)
        pp_size = C.sizeof(abi.PicParams) if magic == MAGIC else abi.PIC_PARAMS_SIZE_V1
        raw = f.read(C.sizeof(abi.SeqParams))
        C.memmove(C.addressof(seq.seq_params), raw, len(raw))
        w, h = seq.seq_params.width, seq.seq_params.height
        while max_pictures is None or len(seq.pictures) < max_pictures:
            hdr = f.read(16)
            if len(hdr) < 16:
                break
            tag, nctu, pic_id, has_planes = struct.unpack("<4i", hdr)
            scaling = None
            pcm = {}
            weights = None
            slice_start = None
            tiles = None
            slices = None
            while tag in (SCAL, PCMS, WPRD, SLST, TILE, SLPR):
                if tag == SCAL:
                    # the 16 bytes just read are the record's tag + 12 bytes of padding
                    scaling = abi.ScalingLists.from_buffer_copy(f.read(C.sizeof(abi.ScalingLists)))
                elif tag == WPRD:
                    weights = abi.PredWeights.from_buffer_copy(f.read(C.sizeof(abi.PredWeights)))
                elif tag == SLST:
                    slice_start = np.frombuffer(f.read(nctu * P * 4), dtype=np.uint32).reshape(nctu, P).copy()
                elif tag == TILE:
                    tiles = np.frombuffer(f.read(nctu * 8), dtype=np.int32).reshape(nctu, 2).copy()
                elif tag == SLPR:
                    slices = (abi.SliceParams * nctu).from_buffer_copy(f.read(nctu * C.sizeof(abi.SliceParams)))
                else:
                    for _ in range(nctu):  # second header word: number of CTUs in the record
                        addr = struct.unpack("<i", f.read(4))[0]
                        pcm[addr] = tuple(np.frombuffer(f.read(n * P * 2), dtype=np.int16).copy() for n in (16, 4, 4))
                hdr = f.read(16)
                tag, nctu, pic_id, has_planes = struct.unpack("<4i", hdr)
            if tag != PICT:
                raise ValueError("%s: bad picture tag" % path)
            pic = PictureSyntax()
            pic.nctu, pic.pic_id = nctu, pic_id
            pic.scaling = scaling
            pic.pcm = pcm
            pic.pred_weights = weights
            pic.slice_start = slice_start
            pic.tiles = tiles
            pic.slices = slices
            raw = f.read(pp_size)
            C.memmove(C.addressof(pic.pic_params), raw, len(raw))

            def arr(dtype, per_ctu):
                n = nctu * per_ctu
                return np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype).reshape(nctu, per_ctu).copy()

            for name, dt in _U8_FIELDS:
                pic.a[name] = arr(dt, P)
            pic.a["mv0"] = arr(np.int16, 2 * P)
            pic.a["mv1"] = arr(np.int16, 2 * P)
            pic.a["ref_idx0"] = arr(np.int8, P)
            pic.a["ref_idx1"] = arr(np.int8, P)
            pic.a["tq_bypass"] = arr(np.uint8, P)
            pic.a["ipcm"] = arr(np.uint8, P)
            pic.a["coeff0"] = arr(np.int16, 16 * P)
            pic.a["coeff1"] = arr(np.int16, 4 * P)
            pic.a["coeff2"] = arr(np.int16, 4 * P)
            pic.sao = np.frombuffer(f.read(nctu * 18 * 4), dtype=np.int32).reshape(nctu, 3, 6).copy()
            md5 = f.read(3 * 3 * 16)
            pic.stage_md5 = [[md5[(s * 3 + c) * 16:(s * 3 + c + 1) * 16].hex() for c in range(3)] for s in range(3)]
            if has_planes:
                pic.planes = []
                for s in range(3):
                    comps = []
                    for c in range(3):
                        pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
                        comps.append(np.frombuffer(f.read(pw * ph * 2), dtype=np.uint16).reshape(ph, pw).copy())
                    pic.planes.append(comps)
            seq.pictures.append(pic)
    return seq


def strip_planes_and_compress(src, dst, preset=6):
    """Re-write a raw dump without the stage pictures, xz-compressed (fixture maker)."""
    seq_size = C.sizeof(abi.SeqParams)
    with open(src, "rb") as f, lzma.open(dst, "wb", preset=preset) as out:
        magic = f.read(8)
        pp_size = C.sizeof(abi.PicParams) if magic == MAGIC else abi.PIC_PARAMS_SIZE_V1
        f.seek(0)
        out.write(f.read(8 + seq_size))
        f.seek(8)
        sp = abi.SeqParams.from_buffer_copy(f.read(seq_size))
        plane_bytes = 3 * (sp.width * sp.height * 3 // 2) * 2
        while True:
            hdr = f.read(16)
            if len(hdr) < 16:
                break
            tag, nctu, pic_id, has_planes = struct.unpack("<4i", hdr)
            if tag == SCAL:  # optional scaling-list record: copied as it is
                out.write(hdr)
                out.write(f.read(C.sizeof(abi.ScalingLists)))
                continue
            if tag == TILE:
                out.write(hdr)
                out.write(f.read(nctu * 8))
                continue
            if tag == SLST:
                out.write(hdr)
                out.write(f.read(nctu * P * 4))
                continue
            if tag == WPRD:
                out.write(hdr)
                out.write(f.read(C.sizeof(abi.PredWeights)))
                continue
            if tag == SLPR:
                out.write(hdr)
                out.write(f.read(nctu * C.sizeof(abi.SliceParams)))
                continue
            if tag == PCMS:
                out.write(hdr)
                out.write(f.read(nctu * _PCM_CTU_BYTES))
                continue
            body = pp_size + nctu * (len(_U8_FIELDS) * P + 2 * 4 * P + 2 * P + 2 * P + (16 + 4 + 4) * P * 2 + 18 * 4) + 144
            out.write(struct.pack("<4i", tag, nctu, pic_id, 0))
            out.write(f.read(body))
            if has_planes:
                f.seek(plane_bytes, 1)


def plane_md5(plane, bit_depth):
    """MD5 of a sample plane with the byte convention of the reference's picture hash
    (TLibCommon/TComPicYuvMD5.cpp:44-59,188-213): 1 byte per sample up to 8 bits, else 2 bytes LE."""
    a = np.ascontiguousarray(plane)
    data = a.astype(np.uint8).tobytes() if bit_depth <= 8 else a.astype("<u2").tobytes()
    return hashlib.md5(data).hexdigest()
