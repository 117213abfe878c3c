// common.h — data layout shared by the host packer and the gfx950 kernels.
//
// Everything a picture needs lives in HBM in a few flat arrays, written once by the host packer
// (packer.cpp) and read by every kernel:
//   * Part4   rec4[(H/4)*(W/4)]  one 16-byte record per 4x4 luma block, RASTER order over the picture
//                                (the reference keeps these per CTU in z-order: TLibCommon/TComDataCU.h:122-204).
//                                It crosses the host link run-length coded (runs[] + ctu_run_off[]: consecutive
//                                units of a CTU, in z-order, with identical records) and is expanded on the device
//                                by k_expand, the first kernel of the first batch that uses the picture
//   * TuItem  tu[4][...]         one entry per coded transform block, grouped by size 4/8/16/32
//   * uint32  coeff[...]         coded coefficients, sparse: per block the number of non-zero coefficients, then one
//                                word (raster position | value << 16) for each
//   * IntraItem intra[...]       one entry per intra prediction block, grouped by CTU, inside a CTU by local
//                                dependency level; IntraCtu ictu[...] one entry per CTU that holds intra blocks,
//                                sorted by CTU wave (a CTU's wave is one more than the highest wave among the
//                                neighbouring CTUs whose intra blocks its own blocks read)
//   * McTile  mc_tile[3][...] / McBlk mc_blk[...], mc_blk8[...]  regions with uniform / mixed motion
//   * SaoCtu  sao[nCtu*3]        SAO parameters per CTU and component
//   * CtuInfo ctu_info[nCtu]     only for pictures with per-slice filter parameters or uncrossed slice / tile boundaries
// Sample planes are uint8 for 8-bit sequences and uint16 otherwise, with no margin: kernels clamp
// reference coordinates instead of reading the reference's 80-sample replicated border.
#ifndef HEVCRECON_COMMON_H
#define HEVCRECON_COMMON_H

#include <stdint.h>

#define HR_MAX_REFS 16
#define HR_MAX_BATCH 256

// Part4.flags
#define P4_DEPTH_MASK 0x03
#define P4_INTRA 0x04
#define P4_PARTSIZE_SHIFT 3 /* 3 bits */
#define P4_BYPASS 0x40 /* no in-loop filtering: lossless coding unit, or IPCM unit of a sequence with pcm_loop_filter_disable */
#define P4_IPCM 0x80
// Part4.tu
#define P4_TRIDX_MASK 0x03
#define P4_CBF_Y 0x04 /* luma cbf at the leaf transform depth */
#define P4_VALID 0x20 /* inside the picture */

struct __attribute__((aligned(16))) Part4 {
  uint8_t flags;
  uint8_t tu;
  int8_t qp;
  uint8_t pad;     /* in a run: z index of the run's first unit inside its CTU; 0 in rec4 */
  int8_t ref_idx[2];
  uint8_t pad2[2]; /* in a run: pad2[0] = units in the run - 1; 0 in rec4 */
  int16_t mv[2][2]; /* [list][hor, ver], quarter luma samples, exactly as parsed (unused lists too) */
};

// TuItem.flags
#define TU_TSKIP 0x01
#define TU_DST 0x02
#define TU_TO_RESI 0x04 /* intra block: write the residual to the residual planes, else add it to the picture */
#define TU_BYPASS 0x08 /* lossless coding unit: the coefficients ARE the residual */
#define TU_PCM 0x10    /* IPCM coding unit: the "coefficients" are the PCM samples, stored as they are (shifted up to the bit depth) */

struct TuItem {
  uint16_t x, y;     /* block position in its component plane */
  uint8_t comp;      /* 0 Y, 1 Cb, 2 Cr */
  uint8_t flags;
  int8_t qp;         /* luma QP of the coding unit */
  uint8_t pad;
  uint32_t coeff_off; /* index of the block's header word in coeff[] */
};

// Motion compensation work lists.  Square luma regions (a whole CTU, else its 32x32 quadrants, else their 16x16
// quadrants) whose units all carry the same (clipped) motion go to mc_tile[0..2] and are predicted as one
// block; what is left of a 16x16 region goes by 8x8 quadrants: in pictures of 8-bit samples a quadrant whose four units
// carry the same motion to mc_blk8 (predicted as one 8x8 block, the motion taken from its first unit's record), every
// other quadrant with an inter unit to mc_blk, predicted 4x4 unit by 4x4 unit from the motion records.
struct McTile {
  uint16_t x, y;     /* luma position of the region */
  int16_t mv[2][2];  /* [list][hor, ver], quarter luma samples, already clipped */
  int8_t ref[2];     /* reference index per list, -1: list not used (identical-motion shortcut applied) */
  uint8_t pad[2];
};
struct McBlk {
  uint16_t x, y; /* luma position of the 8x8 block */
};

// Capacity of the intra kernel's per-CTU tables.  A 64x64 CTU holds at most 256 luma + 2 x 64 chroma 4x4 blocks, and
// its longest dependency chain (all blocks 4x4) has 90 links; the packer refuses anything beyond.
#define HR_INTRA_MAX_ITEMS 384
#define HR_INTRA_MAX_LEVELS 96
// The intra kernel takes a CTU's blocks in rounds: up to HR_INTRA_ROUND_BLOCKS blocks of one local level, one wave
// each.  (90 levels and 384 blocks give at most 90 + 384 / 8 rounds.)
#define HR_INTRA_ROUND_BLOCKS 8
// A 32x32 block stands in the list this many times (1, 2 or 4), each entry predicting its share of the block's rows:
// the waves that take the entries each build the block's reference samples for themselves and share no more than
// that work, so the block's prediction -- the longest chain a round can hold -- is cut without a hand-over.
#ifndef HR_INTRA_SPLIT32
#define HR_INTRA_SPLIT32 4
#endif
#define HR_INTRA_MAX_ROUNDS 144
// A round as the kernel reads it: first block (relative to the CTU's first) and number of blocks
#define IR_BASE_MASK 0x1ff
#define IR_COUNT_SHIFT 9
#define IR_COUNT_MASK 0xf

#define IN_CORNER 0x01
#define IN_HAS_RESI 0x02

struct IntraItem {
  uint16_t x, y;       /* block position in its component plane */
  uint8_t log2n;       /* 2..5 */
  uint8_t comp;
  uint8_t mode;        /* 0 planar, 1 DC, 2..34 angular (chroma DM already resolved) */
  uint8_t flags;
  uint16_t avail_left; /* bit j: unit j of the left column, from the top; units [n/u, 2n/u) are below-left */
  uint16_t avail_above; /* bit i: unit i of the row above, from the left; units [n/u, 2n/u) are above-right */
  uint32_t prep;        /* derived from the fields above when the picture closes (IP_*), so that the kernel need not */
};

// IntraItem.prep.  The 4n+1 reference samples are numbered from the bottom of the left column (0) over the corner
// (2n) to the right end of the row above (4n).  When the available ones form one run [first, last], sample i is
// read from position clamp(i, first, last) (TLibCommon/TComPattern.cpp:424-435,546-588 substitute exactly so).
#define IP_FIRST_SHIFT 0  /* 8 bits: first available sample (also with holes: it stands in for everything before it) */
#define IP_LAST_SHIFT 8   /* 8 bits: last available sample */
#define IP_KIND_SHIFT 16  /* 2 bits: IP_KIND_* */
#define IP_KIND_NONE 0    /* nothing available: every sample is 1 << (bit depth - 1) */
#define IP_KIND_RUN 1     /* one run of available samples */
#define IP_KIND_HOLES 2   /* anything else: resolved per sample from the unit masks */
#define IP_FILTER 0x40000 /* bit 18: the block predicts from smoothed samples (TComPattern::getPredictorPtr) */
#define IP_ANGLE_SHIFT 19 /* 7 bits: intraPredAngle + 32 (angular modes; negative for horizontal modes means the same as for vertical) */
#define IP_ABS_SHIFT 26   /* 4 bits: |mode - 10| or |mode - 26|, whichever applies (index of the inverse-angle table) */
#define IP_PART_SHIFT 30  /* 2 bits: which part of a 32x32 block this entry predicts (see HR_INTRA_SPLIT32) */

/* IntraCtu.deps: neighbouring CTUs whose intra blocks this CTU's blocks read (it has to wait for them) */
#define IC_DEP_LEFT 1
#define IC_DEP_ABOVE_LEFT 2
#define IC_DEP_ABOVE 4
#define IC_DEP_ABOVE_RIGHT 8

struct IntraCtu {
  uint32_t item_start; /* first IntraItem of this CTU */
  uint32_t lvl_start;  /* first entry of this CTU in the packer's per-level block counts (host side: statistics) */
  uint16_t n_items;
  uint16_t n_lvl;      /* local dependency levels */
  uint8_t cx, cy;      /* CTU position in CTUs (pictures are at most 8192 samples = 128 CTUs wide and high) */
  uint8_t deps;        /* IC_DEP_* */
  uint8_t pad;
  uint32_t round_start; /* first entry of this CTU in rounds[] */
  uint16_t n_rounds;
  uint16_t pad2;
};

// De-quantisation matrices of a picture that uses scaling lists (same content as hevcrecon_scaling_lists).
struct ScalingTables {
  uint8_t m4[6][16];
  uint8_t m8[6][64];
  uint8_t m16[6][64];
  uint8_t m32[2][64];
  uint8_t dc16[6];
  uint8_t dc32[2];
};

// Explicit weighted prediction of a picture (same content as hevcrecon_pred_weights): [list][reference index][component]
struct WpTables {
  int16_t log2d[2][HR_MAX_REFS][3];
  int16_t w[2][HR_MAX_REFS][3];
  int16_t o[2][HR_MAX_REFS][3];
};

struct SaoCtu {
  int8_t type; /* -1 off, 0..3 edge classes, 4 band */
  int8_t band; /* band position */
  int8_t offset[4];
  int8_t pad[2];
};

// Per-CTU view of the slice and tile structure, present only for pictures whose slices differ in their parameters
// or whose slices / tiles are not filtered across (slices end at CTU boundaries).
#define CI_DEBLOCK_OFF 0x01 /* the CTU's slice has deblocking disabled */
#define CI_EDGE_LEFT 0x02   /* the CTU's left edge may be deblocked (neighbour in reach of the slice / tile rules) */
#define CI_EDGE_TOP 0x04    /* the same for its top edge */
struct CtuInfo {
  int8_t beta_off_div2, tc_off_div2; /* of the CTU's slice */
  uint8_t flags;                     /* CI_* */
  uint8_t sao_avail;                 /* SAO may classify against samples of the neighbouring CTU: bit 0 left, 1 right,
                                        2 above, 3 below, 4 above-left, 5 above-right, 6 below-left, 7 below-right */
};

struct SeqInfo {
  int32_t w, h, cw, ch;
  int32_t uw, uh; /* picture size in 4x4 units */
  int32_t ctus_w, ctus_h;
  int32_t bd_y, bd_c;
  int32_t strong_intra;
  int32_t stride_y, stride_c; /* plane strides in samples */
  int32_t pcm_bd_y, pcm_bd_c; /* bit depth of IPCM samples */
  int32_t pcm_lf_disable;     /* IPCM units are left alone by deblocking and SAO */
};

// Per-picture job descriptor, read by every kernel through a device array indexed by the
// picture's position in the batch.
struct PicJob {
  void *rec[3];                       /* working picture: prediction, reconstruction, deblocking in place */
  void *out[3];                       /* final picture (DPB entry), written by SAO */
  const void *ref[2][HR_MAX_REFS][3]; /* reference pictures (final planes) */
  int32_t ref_poc[2][HR_MAX_REFS];
  const Part4 *rec4;
  const Part4 *runs;           /* rec4 run-length coded, as uploaded */
  const uint32_t *ctu_run_off; /* first run of every CTU, + the total */
  int32_t need_expand;         /* rec4 has not been expanded from the runs since the last upload */
  const McTile *mc_tile[3]; /* 64x64, 32x32, 16x16 */
  const McBlk *mc_blk, *mc_blk8; /* 8x8 blocks predicted unit by unit / as one block */
  int32_t n_mc_tile[3], n_mc_blk, n_mc_blk8;
  const TuItem *tu[4];
  int32_t n_tu[4];
  const uint32_t *coeff; /* sparse: per block a count, then (position | value << 16) words */
  int16_t *resi[3]; /* residual of intra blocks, picture-shaped planes (same strides as the sample planes);
                       only positions covered by a coded intra block are ever written or read */
  const IntraItem *intra;
  const IntraCtu *ictu;
  const uint16_t *rounds;     /* per CTU: its rounds (IR_*), level by level */
  const int32_t *wave_off;    /* n_waves + 1 entries into ictu[] */
  int32_t n_waves;
  uint32_t *ctu_done;         /* one word per CTU (raster), zero when the batch starts: set by the intra kernel once the
                                 CTU's intra blocks are in the picture */
  const SaoCtu *sao;
  const CtuInfo *ctu_info;      /* NULL: one set of filter parameters, every boundary inside the picture is crossed */
  const ScalingTables *scaling; /* NULL: flat de-quantisation */
  const WpTables *wp;           /* NULL: default (un)weighted prediction */
  int32_t slice_type; /* 0 B, 1 P, 2 I */
  int32_t deblock_disable, beta_off_div2, tc_off_div2;
  int32_t cb_off_pps, cr_off_pps; /* deblocking uses the PPS offsets only */
  int32_t cb_off, cr_off;         /* PPS offset + slice delta, for de-quantisation */
  int32_t sao_luma, sao_chroma;
  int32_t has_bypass; /* some coding unit is exempt from in-loop filtering: SAO has to leave its samples alone */
  int32_t has_pcm;    /* some coding unit is IPCM: its samples are in the picture before the intra kernel runs */
  int32_t *error_flag;
};

// Bits the kernels raise in PicJob::error_flag[0]
#define HR_ERR_EXPAND 1          /* k_expand: a 4x4 unit inside the picture is covered by no run */
#define HR_ERR_INTRA_CAPACITY 2  /* k_intra: a CTU with more blocks or levels than HR_INTRA_MAX_ITEMS / _LEVELS */
#define HR_ERR_INTRA_TIMEOUT 4   /* k_intra: the wait for a neighbouring CTU ran out */

#define HR_MD5_MAX_PICTURES 4096  // pictures per hevcrecon_picture_md5_batch call

// One plane to digest: k_md5 hashes rows x row_bytes bytes (row_bytes a multiple of 4), rows stride_bytes apart.
struct Md5Plane {
  const void *base;
  uint32_t row_bytes, rows, stride_bytes, pad;
};

// Kernel classes for the per-kernel timers (hevcrecon_profile_read).
// K_INTER is also the launch kind of all four inter kernels; its timer slot holds the 64x64 tile kernel.
// K_OTHER holds k_expand.
enum { K_INTER = 0, K_TRANSFORM, K_INTRA, K_DEBLOCK_V, K_DEBLOCK_H, K_SAO, K_COPY, K_OTHER, K_INTER_T32, K_INTER_T16, K_INTER_BLK, K_COUNT };

#endif
