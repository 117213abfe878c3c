/*
 * recon_ref.c — plain-C restatement of the HM 9.1 picture-reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP engine: tests,
 * the smoke check and the CPU-baseline leg of the benchmark may call it; the product
 * library never does.  It is single-threaded, scalar and follows the reference's own
 * processing order (CTU by CTU, CU by CU in z-order, then deblocking, then SAO), which is
 * deliberately different from the engine's schedule.
 *
 * Parity pinning: the final pictures it produces are checked against the per-picture MD5
 * digests carried in the reference encoder's SEI messages / printed by the reference
 * decoder, and its pre-deblocking and post-deblocking pictures against MD5s dumped from
 * the reference decoder (tests/golden, tests/test_oracle_golden.py).
 *
 * Citations are to the reference tree, relative to source/Lib.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/hevcrecon.h"

#define MAXPICS 4096
#define NPART 256
#define MAXCTUS 16384 /* 8192 x 8192 luma samples */
#define MAXSLICES 1024

typedef struct {
  int16_t *p[3];
  int poc;
} Pic;

typedef struct hevcref_ctx {
  hevcrecon_seq_params sp;
  int w, h, cw, ch, ctus_w, ctus_h, nctu;
  Pic pics[MAXPICS];
  int16_t *stage[2][3]; /* copies of the picture before / after deblocking */
  int16_t t4[4][4], t8[8][8], t16[16][16], t32[32][32];
  /* per-call state */
  const hevcrecon_pic_params *pp;
  const hevcrecon_ctu_syntax *ctus;
  int use_sl;                 /* de-quantise with the scaling lists below */
  hevcrecon_scaling_lists sl;
  int ctu_order[MAXCTUS];     /* raster address -> position in coding order (tile scan) */
  int use_wp;                 /* predict with the explicit weights below */
  hevcrecon_pred_weights wp;
  int n_slices;               /* slices differ in their parameters: table below (0: everything from pp) */
  hevcrecon_slice_params slices[MAXSLICES];
  const hevcrecon_slice_params *cur_slice; /* slice of the coding unit / CTU being worked on (NULL without a table) */
  int16_t *rec[3];
  uint8_t zscan_x[NPART], zscan_y[NPART]; /* partition z index -> x,y in 4-sample units */
  uint8_t raster_to_z[16][16];             /* [y][x] -> z */
} hevcref_ctx;

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- tables ----------- */

/* round(64*sqrt(2)*cos(pi*j/64)) as standardised for the HEVC core transform; j = 0..32.
 * Rows of the NxN matrices are sampled from it (TLibCommon/TComRom.cpp:230-304 holds the
 * expanded matrices). */
static const int8_t kCos[33] = {0 /*unused*/, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static int dct_coef(int n, int k, int i) {
  if (k == 0) return 64;
  int m = (k * (2 * i + 1) * (32 / n)) & 127;
  if (m > 64) m = 128 - m;
  return m > 32 ? -kCos[64 - m] : kCos[m];
}
/* TLibCommon/TComRom.cpp:316-322 */
static const int8_t kDst4[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
/* TLibCommon/TComRom.cpp:225-228 */
static const uint8_t kInvQuant[6] = {40, 45, 51, 57, 64, 72};
/* TLibCommon/TComRom.cpp:306-312 */
static const uint8_t kChromaScale[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
                                         20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35,
                                         36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};
/* TLibCommon/TComInterpolationFilter.cpp:55-73 */
static const int8_t kLumaTaps[4][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static const int8_t kChromaTaps[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                         {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
/* TLibCommon/TComLoopFilter.cpp:63-71 */
static const uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  1,  1,  1,  1,  1,  1,  1,  1,  1,
                                2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8,  9,  10, 11, 13, 14, 16, 18, 20, 22, 24};
static const uint8_t kBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,
                                  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
                                  34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};

/* ---------------------------------------------------------------- context ---------- */

hevcref_ctx *hevcref_create(void) {
  hevcref_ctx *c = (hevcref_ctx *)calloc(1, sizeof(*c));
  for (int k = 0; k < 32; k++)
    for (int i = 0; i < 32; i++) {
      c->t32[k][i] = dct_coef(32, k, i);
      if (k < 16 && i < 16) c->t16[k][i] = dct_coef(16, k, i);
      if (k < 8 && i < 8) c->t8[k][i] = dct_coef(8, k, i);
      if (k < 4 && i < 4) c->t4[k][i] = dct_coef(4, k, i);
    }
  for (int z = 0; z < NPART; z++) {
    int x = 0, y = 0;
    for (int b = 0; b < 4; b++) {
      x |= ((z >> (2 * b)) & 1) << b;
      y |= ((z >> (2 * b + 1)) & 1) << b;
    }
    c->zscan_x[z] = x;
    c->zscan_y[z] = y;
    c->raster_to_z[y][x] = z;
  }
  return c;
}

static void free_pic(Pic *p) {
  for (int i = 0; i < 3; i++) {
    free(p->p[i]);
    p->p[i] = NULL;
  }
}

void hevcref_destroy(hevcref_ctx *c) {
  if (!c) return;
  for (int i = 0; i < MAXPICS; i++) free_pic(&c->pics[i]);
  for (int s = 0; s < 2; s++)
    for (int i = 0; i < 3; i++) free(c->stage[s][i]);
  free(c);
}

int hevcref_sequence_init(hevcref_ctx *c, const hevcrecon_seq_params *sp) {
  if (sp->ctu_size != 64 || (sp->ctu_size >> sp->max_cu_depth) != 4) return 1;
  if ((sp->width & 7) || (sp->height & 7)) return 1;
  if (memcmp(&c->sp, sp, sizeof(*sp)) == 0) return 0;
  for (int i = 0; i < MAXPICS; i++) free_pic(&c->pics[i]);
  c->sp = *sp;
  c->w = sp->width;
  c->h = sp->height;
  c->cw = c->w / 2;
  c->ch = c->h / 2;
  c->ctus_w = (c->w + 63) / 64;
  c->ctus_h = (c->h + 63) / 64;
  c->nctu = c->ctus_w * c->ctus_h;
  for (int s = 0; s < 2; s++)
    for (int i = 0; i < 3; i++) {
      free(c->stage[s][i]);
      c->stage[s][i] = (int16_t *)calloc((size_t)c->w * c->h, sizeof(int16_t));
    }
  return 0;
}

static Pic *get_pic(hevcref_ctx *c, int id) {
  if (id < 0 || id >= MAXPICS) return NULL;
  Pic *p = &c->pics[id];
  if (!p->p[0]) {
    p->p[0] = (int16_t *)calloc((size_t)c->w * c->h, sizeof(int16_t));
    p->p[1] = (int16_t *)calloc((size_t)c->cw * c->ch, sizeof(int16_t));
    p->p[2] = (int16_t *)calloc((size_t)c->cw * c->ch, sizeof(int16_t));
  }
  return p;
}

static inline int plane_w(const hevcref_ctx *c, int comp) { return comp ? c->cw : c->w; }
static inline int plane_h(const hevcref_ctx *c, int comp) { return comp ? c->ch : c->h; }
static inline int bit_depth(const hevcref_ctx *c, int comp) { return comp ? c->sp.bit_depth_chroma : c->sp.bit_depth_luma; }
/* Units the in-loop filters leave alone: lossless ones (TComDataCU::isLosslessCoded, TLibCommon/TComDataCU.cpp:1703-1706;
 * the CU flag is only ever set when the PPS enables it) and IPCM ones when the SPS switches their filtering off
 * (bPCMFilter, TLibCommon/TComLoopFilter.cpp:562,633-641; TLibCommon/TComSampleAdaptiveOffset.cpp:1396,1435). */
static inline int unfiltered(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int z) {
  if (ctu->transquant_bypass && ctu->transquant_bypass[z]) return 1;
  return c->sp.pcm_loop_filter_disable && ctu->ipcm && ctu->ipcm[z];
}

/* z index of the partition that holds luma sample (x,y) of CTU-relative coordinates */
static inline int zidx(const hevcref_ctx *c, int x, int y) { return c->raster_to_z[y >> 2][x >> 2]; }

typedef struct {
  const hevcrecon_ctu_syntax *ctu;
  int z;
} PartRef;

/* Partition holding luma sample (x,y) of the picture. */
static PartRef part_at(const hevcref_ctx *c, int x, int y) {
  PartRef r;
  r.ctu = &c->ctus[(y >> 6) * c->ctus_w + (x >> 6)];
  r.z = zidx(c, x & 63, y & 63);
  return r;
}

/* The table entry of the slice partition z of `ctu` belongs to (pcCU->getSlice() of the reference); NULL when the
 * picture came without a table. */
static const hevcrecon_slice_params *slice_of(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int z) {
  if (!c->n_slices) return NULL;
  uint32_t start = ctu->slice_start ? ctu->slice_start[z] : 0;
  for (int i = 0; i < c->n_slices; i++)
    if (c->slices[i].start_addr == start) return &c->slices[i];
  return &c->slices[0];
}

/* ---------------------------------------------------------------- transform -------- */

/* TComTrQuant::setQPforQuant, TLibCommon/TComTrQuant.cpp:192-214; the chroma offset is the PPS's plus the slice's
 * (TLibDecoder/TDecCu.cpp:616-625,772-781) */
static int scaled_qp(const hevcref_ctx *c, int qpy, int comp) {
  if (comp == 0) return qpy + 6 * (c->sp.bit_depth_luma - 8);
  int bdo = 6 * (c->sp.bit_depth_chroma - 8);
  int dcb = c->cur_slice ? c->cur_slice->cb_qp_delta_slice : c->pp->cb_qp_delta_slice;
  int dcr = c->cur_slice ? c->cur_slice->cr_qp_delta_slice : c->pp->cr_qp_delta_slice;
  int off = comp == 1 ? c->pp->cb_qp_offset_pps + dcb : c->pp->cr_qp_offset_pps + dcr;
  int q = clip3(-bdo, 57, qpy + off);
  return q < 0 ? q + bdo : kChromaScale[q] + bdo;
}

/* xDeQuant (flat scaling), TLibCommon/TComTrQuant.cpp:1119-1186; then xIT / xITransformSkip /
 * bypass as selected by invtransformNxN, TLibCommon/TComTrQuant.cpp:1267-1290.
 * coef: n*n int32 raster.  out: n*n residual, raster with stride n. */
static void inverse_tu(const hevcref_ctx *c, const int32_t *coef, int log2n, int comp, int intra, int qpy, int use_dst, int tskip,
                       int bypass, int16_t *out) {
  int n = 1 << log2n, nn = n * n;
  if (bypass) {
    for (int i = 0; i < nn; i++) out[i] = (int16_t)coef[i];
    return;
  }
  int bd = bit_depth(c, comp);
  int q = scaled_qp(c, qpy, comp);
  int per = q / 6, rem = q % 6;
  int shift = bd + log2n - 9; /* QUANT_IQUANT_SHIFT - QUANT_SHIFT - (MAX_TR_DYNAMIC_RANGE - bd - log2n) */
  int add = 1 << (shift - 1);
  int scale = kInvQuant[rem] << per;
  int16_t dq[32 * 32];
  if (c->use_sl) {
    /* scaling-list branch of xDeQuant, TLibCommon/TComTrQuant.cpp:1142-1173, with the de-quantisation
     * coefficients of xSetScalingListDec / processScalingListDec (:2486-2590): invQuantScales[rem] * matrix entry,
     * an entry of the 8x8 base matrix covering ratio x ratio coefficients, DC replaced for 16x16 and 32x32 */
    int type = (intra ? 0 : 3) + comp;
    int mw = n < 8 ? n : 8, ratio = n / mw;
    const uint8_t *m = log2n == 2 ? c->sl.m4[type] : log2n == 3 ? c->sl.m8[type] : log2n == 4 ? c->sl.m16[type] : c->sl.m32[type ? 1 : 0];
    int dc = log2n == 4 ? c->sl.dc16[type] : log2n == 5 ? c->sl.dc32[type ? 1 : 0] : m[0];
    int sh = shift + 4;
    int bit_range = imin(15, 12 + log2n + bd - per), limit = 1 << bit_range;
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        int mk = (ratio > 1 && i == 0 && j == 0) ? dc : m[mw * (j / ratio) + i / ratio];
        int dcoef = kInvQuant[rem] * mk;
        int q0 = coef[j * n + i], r;
        if (sh > per) {
          r = (clip3(-32768, 32767, q0) * dcoef + (1 << (sh - per - 1))) >> (sh - per);
        } else {
          r = (clip3(-limit, limit - 1, q0) * dcoef) << (per - sh);
        }
        dq[j * n + i] = (int16_t)clip3(-32768, 32767, r);
      }
  } else {
    for (int i = 0; i < nn; i++) {
      int v = clip3(-32768, 32767, coef[i]);
      dq[i] = (int16_t)clip3(-32768, 32767, (v * scale + add) >> shift);
    }
  }
  if (tskip) { /* TLibCommon/TComTrQuant.cpp:1454-1486 */
    int s = 15 - bd - log2n;
    if (s > 0) {
      for (int i = 0; i < nn; i++) out[i] = (int16_t)((dq[i] + (1 << (s - 1))) >> s);
    } else {
      for (int i = 0; i < nn; i++) out[i] = (int16_t)(dq[i] << (-s));
    }
    return;
  }
  /* xITrMxN, TLibCommon/TComTrQuant.cpp:829-863: two passes of the transposing 1-D inverse, shifts 7 and
   * 12-(bd-8), each followed by a clip to 16 bits (partialButterflyInverse*, :464-822; fastInverseDst :445). */
  int16_t tmp[32 * 32];
  int s2 = 12 - (bd - 8);
  for (int pass = 0; pass < 2; pass++) {
    const int16_t *src = pass ? tmp : dq;
    int16_t *dst = pass ? out : tmp;
    int sh = pass ? s2 : 7;
    int rnd = 1 << (sh - 1);
    for (int j = 0; j < n; j++)
      for (int i = 0; i < n; i++) {
        int sum = 0;
        for (int k = 0; k < n; k++) {
          int t = use_dst ? kDst4[k][i]
                          : (n == 4 ? c->t4[k][i] : n == 8 ? c->t8[k][i] : n == 16 ? c->t16[k][i] : c->t32[k][i]);
          sum += t * src[k * n + j];
        }
        dst[j * n + i] = (int16_t)clip3(-32768, 32767, (sum + rnd) >> sh);
      }
  }
}

/* ---------------------------------------------------------------- inter ------------ */

static inline int ref_sample(const int16_t *p, int w, int h, int x, int y) {
  return p[clip3(0, h - 1, y) * w + clip3(0, w - 1, x)];
}

/* One reference block -> dst (w x h, stride dstride).  bi = keep the 14-bit intermediate.
 * TComPrediction::xPredInterLumaBlk / xPredInterChromaBlk (TLibCommon/TComPrediction.cpp:557-645) over
 * TComInterpolationFilter::filter / filterCopy (TLibCommon/TComInterpolationFilter.cpp:92-246).
 * The reference reads its 80-sample replicated margin; clamped coordinates are the same samples
 * because clipMv (below) keeps every tap inside that margin. */
static void mc_block(const hevcref_ctx *c, const int16_t *ref, int comp, int bx, int by, int w, int h, int mvx, int mvy,
                     int bi, int16_t *dst, int dstride) {
  int pw = plane_w(c, comp), ph = plane_h(c, comp), bd = bit_depth(c, comp);
  int ntaps = comp ? 4 : 8, fshift = comp ? 3 : 2, fmask = comp ? 7 : 3;
  int xf = mvx & fmask, yf = mvy & fmask;
  int x0 = bx + (mvx >> fshift), y0 = by + (mvy >> fshift);
  const int8_t *tx = comp ? kChromaTaps[xf] : kLumaTaps[xf];
  const int8_t *ty = comp ? kChromaTaps[yf] : kLumaTaps[yf];
  int head = 14 - bd, half = ntaps / 2 - 1;
  int maxv = (1 << bd) - 1;
  if (!xf && !yf) { /* filterCopy */
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int v = ref_sample(ref, pw, ph, x0 + x, y0 + y);
        dst[y * dstride + x] = bi ? (int16_t)((v << head) - 8192) : (int16_t)v;
      }
    return;
  }
  if (!yf || !xf) { /* single pass, isFirst = true */
    const int8_t *t = yf ? ty : tx;
    int shift = bi ? 6 - head : 6;
    int offset = bi ? -(8192 << shift) : 32;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        int sum = 0;
        for (int k = 0; k < ntaps; k++)
          sum += t[k] * (yf ? ref_sample(ref, pw, ph, x0 + x, y0 + y + k - half) : ref_sample(ref, pw, ph, x0 + x + k - half, y0 + y));
        int16_t val = (int16_t)((sum + offset) >> shift);
        if (!bi) val = (int16_t)clip3(0, maxv, val);
        dst[y * dstride + x] = val;
      }
    return;
  }
  /* horizontal into a 16-bit temporary of h + ntaps - 1 rows, then vertical */
  int16_t tmp[(64 + 7) * 64];
  int s1 = 6 - head;
  for (int y = 0; y < h + ntaps - 1; y++)
    for (int x = 0; x < w; x++) {
      int sum = 0;
      for (int k = 0; k < ntaps; k++) sum += tx[k] * ref_sample(ref, pw, ph, x0 + x + k - half, y0 + y - half);
      tmp[y * w + x] = (int16_t)((sum - (8192 << s1)) >> s1);
    }
  int shift = bi ? 6 : 6 + head;
  int offset = bi ? 0 : (1 << (shift - 1)) + (8192 << 6);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int sum = 0;
      for (int k = 0; k < ntaps; k++) sum += ty[k] * tmp[(y + k) * w + x];
      int16_t val = (int16_t)((sum + offset) >> shift);
      if (!bi) val = (int16_t)clip3(0, maxv, val);
      dst[y * dstride + x] = val;
    }
}

/* Inter prediction of one PU into pred[3] (CU-sized buffers, luma stride 64, chroma 32).
 * TComPrediction::motionCompensation, xPredInterUni, xPredInterBi, xCheckIdenticalMotion,
 * xWeightedAverage (TLibCommon/TComPrediction.cpp:396-543,647-661); TComYuv::addAvg
 * (TLibCommon/TComYuv.cpp:520-584); TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3207-3219). */
static void predict_pu(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int cux, int cuy, int px, int py, int pw,
                       int ph, int z, int16_t *pred[3]) {
  int r0 = ctu->ref_idx[0][z], r1 = ctu->ref_idx[1][z];
  int mv[2][2];
  for (int l = 0; l < 2; l++) {
    int hmax = (c->w + 8 - cux - 1) << 2, hmin = (-64 - 8 - cux + 1) << 2;
    int vmax = (c->h + 8 - cuy - 1) << 2, vmin = (-64 - 8 - cuy + 1) << 2;
    mv[l][0] = imin(hmax, imax(hmin, ctu->mv[l][2 * z]));
    mv[l][1] = imin(vmax, imax(vmin, ctu->mv[l][2 * z + 1]));
  }
  int use[2] = {r0 >= 0, r1 >= 0};
  /* TComPrediction::xCheckIdenticalMotion, TLibCommon/TComPrediction.cpp:396-411 (B slices without weighted bi-prediction) */
  if (use[0] && use[1] && c->pp->slice_type == 0 && !c->use_wp && c->pp->ref_poc[0][r0] == c->pp->ref_poc[1][r1] &&
      ctu->mv[0][2 * z] == ctu->mv[1][2 * z] && ctu->mv[0][2 * z + 1] == ctu->mv[1][2 * z + 1])
    use[1] = 0; /* identical motion: plain uni-prediction from list 0 */
  int bi = use[0] && use[1];
  int16_t t0[64 * 64], t1[64 * 64];
  for (int comp = 0; comp < 3; comp++) {
    int sh = comp ? 1 : 0;
    int bw = pw >> sh, bh = ph >> sh, stride = 64 >> sh;
    int16_t *dst = pred[comp] + ((py - cuy) >> sh) * stride + ((px - cux) >> sh);
    int bd = bit_depth(c, comp);
    if (c->use_wp) {
      /* TComPrediction::xPredInterBi with weights, TLibCommon/TComPrediction.cpp:513-538: every list in use is
       * predicted to the 14-bit intermediate, then TComWeightPrediction::xWeightedPredictionUni / -Bi
       * (TLibCommon/TComWeightPrediction.cpp:43-59,79-257) with the parameters of getWpScaling (:271-334) */
      const hevcrecon_pred_weights *wp = &c->wp;
      int maxv = (1 << bd) - 1, shift_num = 14 - bd;
      for (int l = 0; l < 2; l++) {
        if (!use[l]) continue;
        const Pic *rp = &c->pics[c->pp->ref_pic_id[l][l ? r1 : r0]];
        mc_block(c, rp->p[comp], comp, px >> sh, py >> sh, bw, bh, mv[l][0], mv[l][1], 1, l ? t1 : t0, bw);
      }
      if (bi) {
        int w0 = wp->weight[0][r0][comp], w1 = wp->weight[1][r1][comp];
        int offset = wp->offset[0][r0][comp] * (1 << (bd - 8)) + wp->offset[1][r1][comp] * (1 << (bd - 8));
        int shift = wp->log2_denom[0][r0][comp] + 1 + shift_num, round = 1 << (shift - 1);
        for (int y = 0; y < bh; y++)
          for (int x = 0; x < bw; x++)
            dst[y * stride + x] = (int16_t)clip3(
                0, maxv, (w0 * (t0[y * bw + x] + 8192) + w1 * (t1[y * bw + x] + 8192) + round + offset * (1 << (shift - 1))) >> shift);
      } else {
        int l = use[0] ? 0 : 1, ri = l ? r1 : r0;
        const int16_t *t = l ? t1 : t0;
        int w = wp->weight[l][ri][comp], offset = wp->offset[l][ri][comp] * (1 << (bd - 8));
        int shift = wp->log2_denom[l][ri][comp] + shift_num, round = shift ? 1 << (shift - 1) : 0;
        for (int y = 0; y < bh; y++)
          for (int x = 0; x < bw; x++)
            dst[y * stride + x] = (int16_t)clip3(0, maxv, ((w * (t[y * bw + x] + 8192) + round) >> shift) + offset);
      }
      continue;
    }
    for (int l = 0; l < 2; l++) {
      if (!use[l]) continue;
      int ri = l ? r1 : r0;
      const Pic *rp = &c->pics[c->pp->ref_pic_id[l][ri]];
      mc_block(c, rp->p[comp], comp, px >> sh, py >> sh, bw, bh, mv[l][0], mv[l][1], bi, bi ? (l ? t1 : t0) : dst,
               bi ? bw : stride);
    }
    if (bi) {
      int shift = 15 - bd, offset = (1 << (shift - 1)) + 16384;
      for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++)
          dst[y * stride + x] = (int16_t)clip3(0, (1 << bd) - 1, (t0[y * bw + x] + t1[y * bw + x] + offset) >> shift);
    }
  }
}

/* TComTrQuant::invRecurTransformNxN, TLibCommon/TComTrQuant.cpp:1292-1335 (inter residual quadtree).
 * (x,y): luma position of this node relative to the CU; size: luma node size. */
static void inter_residual(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int zcu, int cusize, int comp, int z,
                           int x, int y, int size, int trdepth, int16_t *resi, int rstride) {
  if (!((ctu->cbf[comp][z] >> trdepth) & 1)) return;
  if (trdepth < ctu->tr_idx[z]) {
    int hs = size >> 1, np = (size >> 2) * (size >> 2) / 4;
    for (int q = 0; q < 4; q++)
      inter_residual(c, ctu, zcu, cusize, comp, z + q * np, x + (q & 1) * hs, y + (q >> 1) * hs, hs, trdepth + 1, resi, rstride);
    return;
  }
  int log2 = 0;
  while ((1 << log2) < size) log2++;
  int tx = x, ty = y, tlog2 = log2;
  if (comp) {
    if (log2 == 2) { /* one 4x4 chroma block per 8x8 luma, carried by the first quadrant */
      if (z & 3) return;
    } else {
      tlog2 = log2 - 1;
    }
    tx >>= 1;
    ty >>= 1;
  }
  int16_t blk[32 * 32];
  const int32_t *coef = ctu->coeff[comp] + (comp ? 4 * z : 16 * z);
  int bypass = ctu->transquant_bypass ? ctu->transquant_bypass[z] : 0;
  inverse_tu(c, coef, tlog2, comp, 0, ctu->qp[zcu], 0, ctu->transform_skip[comp][z], bypass, blk);
  int n = 1 << tlog2;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) resi[(ty + j) * rstride + tx + i] = blk[j * n + i];
}

/* ---------------------------------------------------------------- intra ------------ */

/* Position of a 4x4 unit in coding order: CTUs in raster order or tile scan, z-order inside a CTU. */
static inline int coding_order(const hevcref_ctx *c, int ux, int uy) {
  return c->ctu_order[(uy >> 4) * c->ctus_w + (ux >> 4)] * NPART + c->raster_to_z[uy & 15][ux & 15];
}

/* Reference-sample array ref[0..4n]: index 2n is the corner, 2n-1-j is the left column going down,
 * 2n+1+i is the row above going right.  Availability, substitution and the luma smoothing follow
 * TComPattern::initAdiPattern / initAdiPatternChroma / fillReferenceSamples
 * (TLibCommon/TComPattern.cpp:213-602) with the neighbour rules of
 * TComDataCU::getPULeft/Above/AboveLeft/AboveRightAdi/BelowLeftAdi (TLibCommon/TComDataCU.cpp:1141-1589)
 * for one slice, one tile, unconstrained intra: a 4x4 unit is available iff it lies inside the
 * picture and precedes the current block in coding order. */
static void build_intra_refs(const hevcref_ctx *c, int comp, int x0, int y0, int n, int slice_start, int tile_id, int *ref) {
  int sh = comp ? 1 : 0, us = 4 >> sh; /* unit size in samples of this plane */
  int pw = plane_w(c, comp), ph = plane_h(c, comp);
  const int16_t *p = c->rec[comp];
  int nu = n / us, total = 4 * nu + 1;
  int avail[65];
  int cur = coding_order(c, (x0 << sh) >> 2, (y0 << sh) >> 2);
  int navail = 0;
  for (int u = 0; u < total; u++) {
    int sx, sy; /* first sample of the unit */
    if (u < 2 * nu) {
      sx = x0 - 1;
      sy = y0 + 2 * n - (u + 1) * us;
    } else if (u == 2 * nu) {
      sx = x0 - 1;
      sy = y0 - 1;
    } else {
      sx = x0 + (u - 2 * nu - 1) * us;
      sy = y0 - 1;
    }
    int ok = sx >= 0 && sy >= 0 && sx < pw && sy < ph;
    if (ok) {
      int o = coding_order(c, (sx << sh) >> 2, (sy << sh) >> 2);
      ok = o < cur;
      /* neighbours in another CTU must belong to the block's slice (bEnforceSliceRestriction in TComDataCU::getPULeft,
       * getPUAbove, getPUAboveLeft, getPUAboveRightAdi, getPUBelowLeftAdi, TLibCommon/TComDataCU.cpp:1141-1589; no
       * such check for neighbours inside the CTU) */
      if (ok && (o >> 8) != (cur >> 8)) {
        const hevcrecon_ctu_syntax *nc = &c->ctus[(((sy << sh) >> 6)) * c->ctus_w + ((sx << sh) >> 6)];
        ok = o >= slice_start && nc->tile_id == tile_id; /* ... and to its tile (bEnforceTileRestriction) */
      }
    }
    if (ok && c->pp->constrained_intra_pred) { /* TComPattern::is*Available with constrained intra prediction */
      PartRef nb = part_at(c, sx << sh, sy << sh);
      ok = nb.ctu->pred_mode[nb.z] == 1;
    }
    avail[u] = ok;
    navail += ok;
  }
  int dc = 1 << (bit_depth(c, comp) - 1);
  if (navail == 0) {
    for (int i = 0; i <= 4 * n; i++) ref[i] = dc;
    return;
  }
  /* gather */
  for (int i = 0; i <= 4 * n; i++) {
    int u, sx, sy;
    if (i < 2 * n) {
      u = i / us;
      sx = x0 - 1;
      sy = y0 + 2 * n - 1 - i;
    } else if (i == 2 * n) {
      u = 2 * nu;
      sx = x0 - 1;
      sy = y0 - 1;
    } else {
      u = 2 * nu + 1 + (i - 2 * n - 1) / us;
      sx = x0 + (i - 2 * n - 1);
      sy = y0 - 1;
    }
    ref[i] = avail[u] ? p[sy * pw + sx] : -1;
  }
  if (navail == total) return;
  /* substitution, TLibCommon/TComPattern.cpp:546-588 */
  int first_s[65], last_s[65]; /* first / last sample index of each unit inside ref[] */
  for (int u = 0; u < total; u++) {
    if (u < 2 * nu) {
      first_s[u] = u * us;
      last_s[u] = u * us + us - 1;
    } else if (u == 2 * nu) {
      first_s[u] = last_s[u] = 2 * n;
    } else {
      first_s[u] = 2 * n + 1 + (u - 2 * nu - 1) * us;
      last_s[u] = first_s[u] + us - 1;
    }
  }
  int u = 0;
  if (!avail[0]) {
    int nx = 1;
    while (nx < total && !avail[nx]) nx++;
    int v = ref[first_s[nx]];
    for (; u < nx; u++)
      for (int i = first_s[u]; i <= last_s[u]; i++) ref[i] = v;
  }
  for (; u < total; u++) {
    if (avail[u]) continue;
    int v = ref[last_s[u - 1]];
    for (int i = first_s[u]; i <= last_s[u]; i++) ref[i] = v;
  }
}

/* Luma reference smoothing, TLibCommon/TComPattern.cpp:263-356. */
static void filter_intra_refs(const hevcref_ctx *c, int n, const int *ref, int *out) {
  int len = 4 * n + 1;
  if (c->sp.strong_intra_smoothing && n >= 32) {
    int bl = ref[0], tl = ref[2 * n], tr = ref[4 * n];
    int thr = 1 << (c->sp.bit_depth_luma - 5);
    if (iabs(bl + tl - 2 * ref[n]) < thr && iabs(tl + tr - 2 * ref[3 * n]) < thr) {
      int shift = 0;
      while ((1 << shift) < 2 * n) shift++;
      out[0] = ref[0];
      out[2 * n] = ref[2 * n];
      out[4 * n] = ref[4 * n];
      for (int i = 1; i < 2 * n; i++) out[i] = ((2 * n - i) * bl + i * tl + n) >> shift;
      for (int i = 1; i < 2 * n; i++) out[2 * n + i] = ((2 * n - i) * tl + i * tr + n) >> shift;
      return;
    }
  }
  out[0] = ref[0];
  out[len - 1] = ref[len - 1];
  for (int i = 1; i < len - 1; i++) out[i] = (ref[i - 1] + 2 * ref[i] + ref[i + 1] + 2) >> 2;
}

/* TComPrediction::predIntraLumaAng / predIntraChromaAng / xPredIntraAng / xPredIntraPlanar /
 * xDCPredFiltering (TLibCommon/TComPrediction.cpp:122-390,691-763).  pred: n x n raster. */
static void intra_predict(const hevcref_ctx *c, int comp, int n, int mode, const int *ref, int16_t *pred) {
  const int *corner = ref + 2 * n; /* corner[-1-j] = left[j], corner[1+i] = above[i] */
#define LEFT(j) corner[-1 - (j)]
#define ABOVE(i) corner[1 + (i)]
  int bd = bit_depth(c, comp);
  int log2 = 0;
  while ((1 << log2) < n) log2++;
  int edge = comp == 0 && n <= 16;
  if (mode == 0) { /* planar */
    int bl = LEFT(n), tr = ABOVE(n);
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) {
        int hor = (LEFT(y) << log2) + n + (x + 1) * (tr - LEFT(y));
        int ver = (ABOVE(x) << log2) + (y + 1) * (bl - ABOVE(x));
        pred[y * n + x] = (int16_t)((hor + ver) >> (log2 + 1));
      }
    return;
  }
  if (mode == 1) { /* DC */
    int sum = 0;
    for (int i = 0; i < n; i++) sum += ABOVE(i) + LEFT(i);
    int dcv = (sum + n) / (2 * n);
    for (int i = 0; i < n * n; i++) pred[i] = (int16_t)dcv;
    if (edge) {
      pred[0] = (int16_t)((ABOVE(0) + LEFT(0) + 2 * dcv + 2) >> 2);
      for (int x = 1; x < n; x++) pred[x] = (int16_t)((ABOVE(x) + 3 * dcv + 2) >> 2);
      for (int y = 1; y < n; y++) pred[y * n] = (int16_t)((LEFT(y) + 3 * dcv + 2) >> 2);
    }
    return;
  }
  static const int ang_table[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
  static const int inv_ang_table[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
  int hor = mode < 18;
  int a = hor ? -(mode - 10) : mode - 26;
  int absang = ang_table[iabs(a)], inv = inv_ang_table[iabs(a)];
  int angle = a < 0 ? -absang : absang;
  int main_buf[3 * 32 + 2], *rmain = main_buf + 32; /* rmain[0] = corner */
  int side_buf[2 * 32 + 2];
  for (int k = 0; k <= 2 * n; k++) {
    rmain[k] = hor ? corner[-k] : corner[k];
    side_buf[k] = hor ? corner[k] : corner[-k];
  }
  if (angle < 0) {
    int sum = 128;
    for (int k = -1; k > ((n * angle) >> 5); k--) {
      sum += inv;
      rmain[k] = side_buf[sum >> 8];
    }
  }
  int16_t tmp[32 * 32];
  if (angle == 0) {
    for (int y = 0; y < n; y++)
      for (int x = 0; x < n; x++) tmp[y * n + x] = (int16_t)rmain[x + 1];
    if (edge)
      for (int y = 0; y < n; y++) tmp[y * n] = (int16_t)clip3(0, (1 << bd) - 1, tmp[y * n] + ((side_buf[y + 1] - side_buf[0]) >> 1));
  } else {
    int pos = 0;
    for (int y = 0; y < n; y++) {
      pos += angle;
      int di = pos >> 5, df = pos & 31;
      for (int x = 0; x < n; x++) {
        int idx = x + di + 1;
        tmp[y * n + x] = df ? (int16_t)(((32 - df) * rmain[idx] + df * rmain[idx + 1] + 16) >> 5) : (int16_t)rmain[idx];
      }
    }
  }
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) pred[y * n + x] = hor ? tmp[x * n + y] : tmp[y * n + x];
#undef LEFT
#undef ABOVE
}

/* TDecCu::xIntraRecLumaBlk / xIntraRecChromaBlk, TLibDecoder/TDecCu.cpp:489-648.
 * (x0,y0): block position in the component plane, picture coordinates. */
static void intra_block(hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int zcu, int z, int comp, int x0, int y0,
                        int log2, int mode) {
  int n = 1 << log2;
  int ref[4 * 32 + 1], fref[4 * 32 + 1];
  build_intra_refs(c, comp, x0, y0, n, ctu->slice_start ? (int)ctu->slice_start[z] : 0, ctu->tile_id, ref);
  const int *use = ref;
  if (comp == 0 && mode != 1) { /* TComPattern::getPredictorPtr, TLibCommon/TComPattern.cpp:627-651 */
    static const int thr[4] = {10, 7, 1, 0};
    int diff = imin(iabs(mode - 10), iabs(mode - 26));
    if (diff > thr[log2 - 2]) {
      filter_intra_refs(c, n, ref, fref);
      use = fref;
    }
  }
  int16_t pred[32 * 32], resi[32 * 32];
  intra_predict(c, comp, n, mode, use, pred);
  const int32_t *coef = ctu->coeff[comp] + (comp ? 4 * z : 16 * z);
  int bypass = ctu->transquant_bypass ? ctu->transquant_bypass[z] : 0;
  inverse_tu(c, coef, log2, comp, 1, ctu->qp[zcu], comp == 0 && log2 == 2, ctu->transform_skip[comp][z], bypass, resi);
  int pw = plane_w(c, comp), maxv = (1 << bit_depth(c, comp)) - 1;
  for (int y = 0; y < n; y++)
    for (int x = 0; x < n; x++) c->rec[comp][(y0 + y) * pw + x0 + x] = (int16_t)clip3(0, maxv, pred[y * n + x] + resi[y * n + x]);
}

/* TDecCu::xIntraLumaRecQT / xIntraChromaRecQT, TLibDecoder/TDecCu.cpp:686-743. */
static void intra_qt(hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int cusize, int z,
                     int size, int trdepth, int chroma) {
  if (trdepth < ctu->tr_idx[z]) {
    int np = (size >> 2) * (size >> 2) / 4;
    for (int q = 0; q < 4; q++) intra_qt(c, ctu, ctux, ctuy, zcu, cusize, z + q * np, size >> 1, trdepth + 1, chroma);
    return;
  }
  int x = ctux + 4 * c->zscan_x[z], y = ctuy + 4 * c->zscan_y[z];
  int log2 = 0;
  while ((1 << log2) < size) log2++;
  if (!chroma) {
    intra_block(c, ctu, zcu, z, 0, x, y, log2, ctu->luma_dir[z]);
    return;
  }
  int clog2 = log2 - 1;
  if (log2 == 2) { /* 4x4 luma: chroma handled once for the enclosing 8x8, at its first quadrant */
    if (z & 3) return;
    clog2 = 2;
  }
  int mode = ctu->chroma_dir[zcu];
  if (mode == 36) mode = ctu->luma_dir[zcu];
  intra_block(c, ctu, zcu, z, 1, x >> 1, y >> 1, clog2, mode);
  intra_block(c, ctu, zcu, z, 2, x >> 1, y >> 1, clog2, mode);
}

/* ---------------------------------------------------------------- CU walk ---------- */

/* TDecCu::xDecompressCU, TLibDecoder/TDecCu.cpp:404-466 */
static void decompress_cu(hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int depth) {
  int size = 64 >> depth;
  int x = ctux + 4 * c->zscan_x[z], y = ctuy + 4 * c->zscan_y[z];
  int boundary = x + size > c->w || y + size > c->h;
  if ((depth < ctu->depth[z] && depth < c->sp.max_cu_depth - 1) || boundary) {
    int np = NPART >> (2 * (depth + 1));
    for (int q = 0; q < 4; q++) {
      int zq = z + q * np;
      int qx = ctux + 4 * c->zscan_x[zq], qy = ctuy + 4 * c->zscan_y[zq];
      if (qx < c->w && qy < c->h) decompress_cu(c, ctu, ctux, ctuy, zq, depth + 1);
    }
    return;
  }
  c->cur_slice = slice_of(c, ctu, z);
  if (ctu->pred_mode[z] == 1) {
    if (ctu->ipcm && ctu->ipcm[z]) {
      /* TDecCu::xReconPCM / xDecodePCMTexture, TLibDecoder/TDecCu.cpp:798-871: the CU's PCM samples (offset
       * 16*z luma, 4*z chroma, raster inside the CU) shifted up to the sample bit depth */
      for (int comp = 0; comp < 3; comp++) {
        int sh = comp ? 1 : 0, n = size >> sh, pw = plane_w(c, comp);
        int up = bit_depth(c, comp) - (comp ? c->sp.pcm_bit_depth_chroma : c->sp.pcm_bit_depth_luma);
        const int16_t *src = ctu->pcm[comp] + (comp ? 4 * z : 16 * z);
        for (int j = 0; j < n; j++)
          for (int i = 0; i < n; i++) c->rec[comp][((y >> sh) + j) * pw + (x >> sh) + i] = (int16_t)(src[j * n + i] << up);
      }
      return;
    }
    int parts = ctu->part_size[z] == 3 ? 4 : 1; /* TDecCu::xReconIntraQT, TLibDecoder/TDecCu.cpp:652-674 */
    int init = parts == 4 ? 1 : 0;
    int np = (size >> 2) * (size >> 2) / 4;
    for (int chroma = 0; chroma < 2; chroma++)
      for (int pu = 0; pu < parts; pu++)
        intra_qt(c, ctu, ctux, ctuy, z, size, z + (parts == 4 ? pu * np : 0), size >> init, init, chroma);
    return;
  }
  /* TDecCu::xReconInter, TLibDecoder/TDecCu.cpp:468-486 */
  int16_t predbuf[3][64 * 64], resibuf[3][64 * 64];
  int16_t *pred[3] = {predbuf[0], predbuf[1], predbuf[2]};
  int np = NPART >> (2 * depth);
  int q = size >> 2, hsz = size >> 1;
  /* TComDataCU::getPartIndexAndSize, TLibCommon/TComDataCU.cpp:2206-2243: {x, y, w, h} per PU */
  int pu[4][4], npu = 1;
  switch (ctu->part_size[z]) {
    case 1: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = size; pu[0][3] = hsz; pu[1][0] = 0; pu[1][1] = hsz; pu[1][2] = size; pu[1][3] = hsz; break;
    case 2: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = hsz; pu[0][3] = size; pu[1][0] = hsz; pu[1][1] = 0; pu[1][2] = hsz; pu[1][3] = size; break;
    case 3:
      npu = 4;
      for (int i = 0; i < 4; i++) { pu[i][0] = (i & 1) * hsz; pu[i][1] = (i >> 1) * hsz; pu[i][2] = hsz; pu[i][3] = hsz; }
      break;
    case 4: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = size; pu[0][3] = q; pu[1][0] = 0; pu[1][1] = q; pu[1][2] = size; pu[1][3] = size - q; break;
    case 5: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = size; pu[0][3] = size - q; pu[1][0] = 0; pu[1][1] = size - q; pu[1][2] = size; pu[1][3] = q; break;
    case 6: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = q; pu[0][3] = size; pu[1][0] = q; pu[1][1] = 0; pu[1][2] = size - q; pu[1][3] = size; break;
    case 7: npu = 2; pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = size - q; pu[0][3] = size; pu[1][0] = size - q; pu[1][1] = 0; pu[1][2] = q; pu[1][3] = size; break;
    default: pu[0][0] = 0; pu[0][1] = 0; pu[0][2] = size; pu[0][3] = size; break;
  }
  (void)np;
  for (int i = 0; i < npu; i++) {
    int px = x + pu[i][0], py = y + pu[i][1];
    predict_pu(c, ctu, x, y, px, py, pu[i][2], pu[i][3], zidx(c, px - ctux, py - ctuy), pred);
  }
  int any = ctu->cbf[0][z] || ctu->cbf[1][z] || ctu->cbf[2][z];
  memset(resibuf, 0, sizeof(resibuf));
  if (any)
    for (int comp = 0; comp < 3; comp++) inter_residual(c, ctu, z, size, comp, z, 0, 0, size, 0, resibuf[comp], comp ? 32 : 64);
  for (int comp = 0; comp < 3; comp++) {
    int sh = comp ? 1 : 0, stride = 64 >> sh, pw = plane_w(c, comp), maxv = (1 << bit_depth(c, comp)) - 1;
    for (int j = 0; j < (size >> sh); j++)
      for (int i = 0; i < (size >> sh); i++) {
        int v = pred[comp][j * stride + i];
        if (any) v = clip3(0, maxv, v + resibuf[comp][j * stride + i]);
        c->rec[comp][((y >> sh) + j) * pw + (x >> sh) + i] = (int16_t)v;
      }
  }
}

/* ---------------------------------------------------------------- deblocking ------- */

/* TComLoopFilter::xGetBoundaryStrengthSingle, TLibCommon/TComLoopFilter.cpp:421-543.
 * q at (x,y); p is the partition left of / above it.  tu_edge: the edge is a transform-block edge. */
static int boundary_strength(const hevcref_ctx *c, int x, int y, int dir, int tu_edge) {
  PartRef q = part_at(c, x, y), p = dir == 0 ? part_at(c, x - 1, y) : part_at(c, x, y - 1);
  if (p.ctu->pred_mode[p.z] == 1 || q.ctu->pred_mode[q.z] == 1) return 2;
  if (tu_edge && (((q.ctu->cbf[0][q.z] >> q.ctu->tr_idx[q.z]) & 1) || ((p.ctu->cbf[0][p.z] >> p.ctu->tr_idx[p.z]) & 1))) return 1;
  const hevcrecon_pic_params *pp = c->pp;
  const int none = -0x7fffffff;
  int rp[2], rq[2], mp[2][2], mq[2][2];
  for (int l = 0; l < 2; l++) {
    int ip = p.ctu->ref_idx[l][p.z], iq = q.ctu->ref_idx[l][q.z];
    rp[l] = ip < 0 ? none : pp->ref_poc[l][ip];
    rq[l] = iq < 0 ? none : pp->ref_poc[l][iq];
    for (int k = 0; k < 2; k++) {
      mp[l][k] = p.ctu->mv[l][2 * p.z + k];
      mq[l][k] = q.ctu->mv[l][2 * q.z + k];
    }
  }
#define FAR(a, b) (iabs((a)[0] - (b)[0]) >= 4 || iabs((a)[1] - (b)[1]) >= 4)
  if (pp->slice_type == 0) {
    if ((rp[0] == rq[0] && rp[1] == rq[1]) || (rp[0] == rq[1] && rp[1] == rq[0])) {
      if (rp[0] != rp[1]) {
        if (rp[0] == rq[0]) return FAR(mp[0], mq[0]) || FAR(mp[1], mq[1]);
        return FAR(mp[0], mq[1]) || FAR(mp[1], mq[0]);
      }
      return (FAR(mp[0], mq[1]) || FAR(mp[1], mq[0])) && (FAR(mp[0], mq[0]) || FAR(mp[1], mq[1]));
    }
    return 1;
  }
  return rp[0] != rq[0] || FAR(mp[0], mq[0]);
#undef FAR
}

static int chroma_qp_deblock(int q) { return q < 0 ? q : (q > 57 ? q - 6 : kChromaScale[q]); }

/* TComLoopFilter::xEdgeFilterLuma + xPelFilterLuma, TLibCommon/TComLoopFilter.cpp:546-660,809-868:
 * one 4-sample segment.  s points at q0 of the first line; step = distance across the edge,
 * line = distance between the 4 lines.  keep_p / keep_q: that side belongs to a lossless coding unit and is
 * put back after filtering (bPartPNoFilter / bPartQNoFilter, :633-642,856-867). */
static void filter_luma_segment(const hevcref_ctx *c, int16_t *s, int step, int line, int bs, int qp, int keep_p, int keep_q) {
  int scale = 1 << (c->sp.bit_depth_luma - 8), maxv = (1 << c->sp.bit_depth_luma) - 1;
  int tc_off = c->cur_slice ? c->cur_slice->tc_offset_div2 : c->pp->tc_offset_div2;
  int beta_off = c->cur_slice ? c->cur_slice->beta_offset_div2 : c->pp->beta_offset_div2;
  int tc = kTc[clip3(0, 53, qp + 2 * (bs - 1) + 2 * tc_off)] * scale;
  int beta = kBeta[clip3(0, 51, qp + 2 * beta_off)] * scale;
  int side = (beta + (beta >> 1)) >> 3, thr_cut = tc * 10;
#define P(i, k) s[(k) * line - ((i) + 1) * step]
#define Q(i, k) s[(k) * line + (i) * step]
  int dp0 = iabs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dq0 = iabs(Q(0, 0) - 2 * Q(1, 0) + Q(2, 0));
  int dp3 = iabs(P(2, 3) - 2 * P(1, 3) + P(0, 3)), dq3 = iabs(Q(0, 3) - 2 * Q(1, 3) + Q(2, 3));
  int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if (d >= beta) return;
  int fp = dp < side, fq = dq < side;
  int sw = 1;
  for (int k = 0; k < 4; k += 3) {
    int dd = 2 * (k ? d3 : d0);
    int strong = iabs(P(3, k) - P(0, k)) + iabs(Q(3, k) - Q(0, k));
    sw = sw && strong < (beta >> 3) && dd < (beta >> 2) && iabs(P(0, k) - Q(0, k)) < ((tc * 5 + 1) >> 1);
  }
  for (int k = 0; k < 4; k++) {
    int m0 = P(3, k), m1 = P(2, k), m2 = P(1, k), m3 = P(0, k), m4 = Q(0, k), m5 = Q(1, k), m6 = Q(2, k), m7 = Q(3, k);
    if (sw) {
      P(0, k) = (int16_t)clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
      Q(0, k) = (int16_t)clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
      P(1, k) = (int16_t)clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
      Q(1, k) = (int16_t)clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
      P(2, k) = (int16_t)clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
      Q(2, k) = (int16_t)clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    } else {
      int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
      if (iabs(delta) < thr_cut) {
        delta = clip3(-tc, tc, delta);
        P(0, k) = (int16_t)clip3(0, maxv, m3 + delta);
        Q(0, k) = (int16_t)clip3(0, maxv, m4 - delta);
        int tc2 = tc >> 1;
        if (fp) P(1, k) = (int16_t)clip3(0, maxv, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
        if (fq) Q(1, k) = (int16_t)clip3(0, maxv, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
      }
    }
    if (keep_p) P(0, k) = (int16_t)m3, P(1, k) = (int16_t)m2, P(2, k) = (int16_t)m1;
    if (keep_q) Q(0, k) = (int16_t)m4, Q(1, k) = (int16_t)m5, Q(2, k) = (int16_t)m6;
  }
#undef P
#undef Q
}

/* TComLoopFilter::xEdgeFilterChroma + xPelFilterChroma, TLibCommon/TComLoopFilter.cpp:663-792,879-900:
 * the two chroma lines belonging to one 4-luma-sample unit. */
static void filter_chroma_segment(const hevcref_ctx *c, int16_t *s, int step, int line, int qp_avg, int pps_off, int keep_p, int keep_q) {
  int scale = 1 << (c->sp.bit_depth_chroma - 8), maxv = (1 << c->sp.bit_depth_chroma) - 1;
  int qp = chroma_qp_deblock(qp_avg + pps_off);
  int tc = kTc[clip3(0, 53, qp + 2 + 2 * (c->cur_slice ? c->cur_slice->tc_offset_div2 : c->pp->tc_offset_div2))] * scale;
  for (int k = 0; k < 2; k++) {
    int16_t *m = s + k * line;
    int m2 = m[-2 * step], m3 = m[-step], m4 = m[0], m5 = m[step];
    int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    if (!keep_p) m[-step] = (int16_t)clip3(0, maxv, m3 + delta);
    if (!keep_q) m[0] = (int16_t)clip3(0, maxv, m4 - delta);
  }
}

/* TComDataCU::getPULeft / getPUAbove (TLibCommon/TComDataCU.cpp:1141-1238) as the loop filter calls them
 * (TLibCommon/TComLoopFilter.cpp:383,405): a neighbour inside the same CTU always exists; one in the CTU to the left /
 * above is dropped when the slice of partition z does not filter across slice boundaries and the neighbour precedes
 * that slice's start, or when tiles are not crossed and the neighbouring CTU belongs to another tile. */
static int edge_neighbour_usable(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int dir) {
  int ux = c->zscan_x[z], uy = c->zscan_y[z];
  if ((dir == 0 ? ux : uy) != 0) return 1;
  int a = (ctuy >> 6) * c->ctus_w + (ctux >> 6);
  int na = dir == 0 ? a - 1 : a - c->ctus_w;
  int nz = dir == 0 ? c->raster_to_z[uy][15] : c->raster_to_z[15][ux];
  const hevcrecon_slice_params *sl = slice_of(c, ctu, z);
  if (sl && !sl->loop_filter_across_slices) {
    uint32_t start = ctu->slice_start ? ctu->slice_start[z] : 0;
    if ((uint32_t)(c->ctu_order[na] * NPART + nz) < start) return 0;
  }
  if (c->pp->no_loop_filter_across_tiles && c->ctus[na].tile_id != ctu->tile_id) return 0;
  return 1;
}

/* TComLoopFilter::xDeblockCU for one leaf CU and one direction, TLibCommon/TComLoopFilter.cpp:186-359. */
static void deblock_cu(hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int depth, int dir) {
  int size = 64 >> depth;
  int skip_cu_edge = 0;
  if (depth < ctu->depth[z]) {
    int np = NPART >> (2 * (depth + 1));
    for (int q = 0; q < 4; q++) {
      int zq = z + q * np;
      int qx = ctux + 4 * c->zscan_x[zq], qy = ctuy + 4 * c->zscan_y[zq];
      if (qx < c->w && qy < c->h) deblock_cu(c, ctu, ctux, ctuy, zq, depth + 1, dir);
    }
    return;
  }
  if (ctu->part_size[z] == 15) return;
  /* the parameters are those of the slice the CTU's object points at (pcCU->getSlice(), :371,569-570,769) */
  c->cur_slice = slice_of(c, ctu, z);
  if (c->cur_slice ? c->cur_slice->deblocking_disable : c->pp->deblocking_disable) return;
  int x0 = ctux + 4 * c->zscan_x[z], y0 = ctuy + 4 * c->zscan_y[z];
  /* xSetLoopfilterParam (:363-419): the coding unit's left / top edge is filtered when the neighbour of its first
   * partition exists for getPULeft / getPUAbove with the slice and tile restrictions the flags ask for */
  if ((dir == 0 ? x0 : y0) > 0 && !edge_neighbour_usable(c, ctu, ctux, ctuy, z, dir)) skip_cu_edge = 1;
  int tusize = size >> ctu->tr_idx[z]; /* transform edges; refined per partition below */
  (void)tusize;
  int ps = ctu->part_size[z];
  /* offsets (in luma samples, relative to the CU) of prediction-block edges in this direction */
  int pu_edge = -1;
  if (dir == 0) {
    if (ps == 2 || ps == 3) pu_edge = size / 2;
    if (ps == 6) pu_edge = size / 4;
    if (ps == 7) pu_edge = size - size / 4;
  } else {
    if (ps == 1 || ps == 3) pu_edge = size / 2;
    if (ps == 4) pu_edge = size / 4;
    if (ps == 5) pu_edge = size - size / 4;
  }
  for (int e = 0; e < size; e += 8) {    /* edges on the 8-sample grid */
    for (int u = 0; u < size; u += 4) {  /* 4-sample units along the edge */
      int x = dir == 0 ? x0 + e : x0 + u, y = dir == 0 ? y0 + u : y0 + e;
      if ((dir == 0 ? x : y) == 0) continue; /* picture border */
      if (e == 0 && skip_cu_edge) continue;
      PartRef q = part_at(c, x, y);
      int tsz = (64 >> q.ctu->depth[q.z]) >> q.ctu->tr_idx[q.z];
      int tu_edge = ((dir == 0 ? x : y) & (tsz - 1)) == 0;
      if (!tu_edge && e != pu_edge) continue;
      int bs = boundary_strength(c, x, y, dir, tu_edge);
      if (!bs) continue;
      PartRef p = dir == 0 ? part_at(c, x - 1, y) : part_at(c, x, y - 1);
      int qp_avg = (p.ctu->qp[p.z] + q.ctu->qp[q.z] + 1) >> 1;
      int16_t *s = c->rec[0] + y * c->w + x;
      int keep_p = unfiltered(c, p.ctu, p.z), keep_q = unfiltered(c, q.ctu, q.z);
      filter_luma_segment(c, s, dir == 0 ? 1 : c->w, dir == 0 ? c->w : 1, bs, qp_avg, keep_p, keep_q);
      if (bs == 2 && ((dir == 0 ? x : y) & 15) == 0) {
        int cx = x >> 1, cy = y >> 1;
        filter_chroma_segment(c, c->rec[1] + cy * c->cw + cx, dir == 0 ? 1 : c->cw, dir == 0 ? c->cw : 1, qp_avg, c->pp->cb_qp_offset_pps, keep_p, keep_q);
        filter_chroma_segment(c, c->rec[2] + cy * c->cw + cx, dir == 0 ? 1 : c->cw, dir == 0 ? c->cw : 1, qp_avg, c->pp->cr_qp_offset_pps, keep_p, keep_q);
      }
    }
  }
}

/* ---------------------------------------------------------------- SAO -------------- */

static inline int sgn(int v) { return (v > 0) - (v < 0); }

/* TComSampleAdaptiveOffset::processSaoUnitAll / processSaoCuOrg,
 * TLibCommon/TComSampleAdaptiveOffset.cpp:814-1039,1097-1261.  The reference filters in place and
 * keeps line buffers so that every classification sees deblocked samples; reading from an untouched
 * copy (src) is the same thing. */
/* May SAO classify a sample of CTU `a` against a sample of the CTU (ox, oy) CTUs away?  The neighbouring block exists,
 * and: same slice, or the later of the two slices filters across its boundaries; and, when tiles are not crossed, no
 * tile boundary on the sides of `a` that the neighbour lies beyond.  TComDataCU::setNDBFilterBlockBorderAvailability,
 * TLibCommon/TComDataCU.cpp:3813-4166 (one block per CTU: slices end at CTU boundaries), with the tile sides found
 * by TComPic::createNonDBFilterInfo, TLibCommon/TComPic.cpp:301-327. */
static int sao_block_available(const hevcref_ctx *c, int a, int ox, int oy) {
  int cx = a % c->ctus_w + ox, cy = a / c->ctus_w + oy;
  if (cx < 0 || cy < 0 || cx >= c->ctus_w || cy >= c->ctus_h) return 0;
  int n = cy * c->ctus_w + cx;
  const hevcrecon_ctu_syntax *me = &c->ctus[a], *nb = &c->ctus[n];
  uint32_t s_me = me->slice_start ? me->slice_start[0] : 0, s_nb = nb->slice_start ? nb->slice_start[0] : 0;
  if (s_me != s_nb) {
    const hevcrecon_slice_params *later = s_nb > s_me ? slice_of(c, nb, 0) : slice_of(c, me, 0);
    if (later && !later->loop_filter_across_slices) return 0;
  }
  if (c->pp->no_loop_filter_across_tiles) {
    if (ox && c->ctus[a + ox].tile_id != me->tile_id) return 0;
    if (oy && c->ctus[a + oy * c->ctus_w].tile_id != me->tile_id) return 0;
  }
  return 1;
}

/* TComSampleAdaptiveOffset::processSaoUnitAll / processSaoCu (TLibCommon/TComSampleAdaptiveOffset.cpp:546-579,
 * 1097-1261) with processSaoCuOrg (:814-1039) or, when slices or tiles are not crossed, processSaoBlock (:592-809):
 * in both a sample is classified only when its two neighbours along the class direction lie in the CTU itself or in a
 * neighbouring block that may be used (processSaoCuOrg: any block inside the picture).  The reference filters in place
 * and keeps line buffers / a copy so that every classification sees deblocked samples; reading from an untouched
 * copy (src) is the same thing. */
static void sao_ctu(const hevcref_ctx *c, const hevcrecon_ctu_syntax *ctu, int comp, int ctux, int ctuy, const int16_t *src,
                    int16_t *dst) {
  int type = ctu->sao_type[comp];
  if (type < 0) return;
  int sh = comp ? 1 : 0, pw = plane_w(c, comp), ph = plane_h(c, comp), bd = bit_depth(c, comp);
  int x0 = ctux >> sh, y0 = ctuy >> sh, x1 = imin(x0 + (64 >> sh), pw), y1 = imin(y0 + (64 >> sh), ph);
  int inc = imax(bd - 10, 0), maxv = (1 << bd) - 1;
  if (type == 4) {
    int band[32];
    memset(band, 0, sizeof(band));
    for (int i = 0; i < 4; i++) band[(ctu->sao_subtype[comp] + i) & 31] = ctu->sao_offset[comp][i] << inc;
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) dst[y * pw + x] = (int16_t)clip3(0, maxv, src[y * pw + x] + band[src[y * pw + x] >> (bd - 5)]);
    return;
  }
  static const int dx[4] = {1, 0, 1, -1}, dy[4] = {0, 1, 1, 1};
  static const int eo_map[5] = {1, 2, 0, 3, 4};
  int off[5] = {0};
  for (int i = 0; i < 4; i++) off[i + 1] = ctu->sao_offset[comp][i] << inc;
  int a = (ctuy >> 6) * c->ctus_w + (ctux >> 6);
  int avail[3][3];
  for (int oy = -1; oy <= 1; oy++)
    for (int ox = -1; ox <= 1; ox++) avail[oy + 1][ox + 1] = (ox == 0 && oy == 0) || sao_block_available(c, a, ox, oy);
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++) {
      int ax = x - dx[type], ay = y - dy[type], bx = x + dx[type], by = y + dy[type];
      if (!avail[(ay >= y1) - (ay < y0) + 1][(ax >= x1) - (ax < x0) + 1]) continue;
      if (!avail[(by >= y1) - (by < y0) + 1][(bx >= x1) - (bx < x0) + 1]) continue;
      int v = src[y * pw + x];
      int e = sgn(v - src[ay * pw + ax]) + sgn(v - src[by * pw + bx]) + 2;
      dst[y * pw + x] = (int16_t)clip3(0, maxv, v + off[eo_map[e]]);
    }
}

/* ---------------------------------------------------------------- entry points ----- */

/* Scaling lists for the pictures reconstructed from now on (NULL: flat de-quantisation). */
void hevcref_set_scaling_lists(hevcref_ctx *c, const hevcrecon_scaling_lists *sl) {
  c->use_sl = sl != NULL;
  if (sl) c->sl = *sl;
}

/* Slice table for the pictures reconstructed from now on (n = 0: one set of parameters per picture). */
int hevcref_set_slices(hevcref_ctx *c, int n, const hevcrecon_slice_params *sl) {
  if (n < 0 || n > MAXSLICES) return 1;
  c->n_slices = n;
  if (n) memcpy(c->slices, sl, sizeof(*sl) * n);
  return 0;
}

/* Explicit prediction weights for the pictures reconstructed from now on (NULL: default prediction). */
void hevcref_set_pred_weights(hevcref_ctx *c, const hevcrecon_pred_weights *w) {
  c->use_wp = w != NULL;
  if (w) c->wp = *w;
}

int hevcref_reconstruct(hevcref_ctx *c, const hevcrecon_pic_params *pp, const hevcrecon_ctu_syntax *ctus, int nctu, int dst_id) {
  if (nctu != c->nctu) return 1;
  Pic *dst = get_pic(c, dst_id);
  if (!dst) return 1;
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < pp->num_ref[l]; i++) {
      int id = pp->ref_pic_id[l][i];
      if (id < 0 || id >= MAXPICS || !c->pics[id].p[0]) return 2;
    }
  c->pp = pp;
  c->ctus = ctus;
  for (int i = 0; i < 3; i++) c->rec[i] = c->stage[1][i];
  /* CTUs in coding order: raster, or tile scan when the caller numbers them */
  if (nctu > MAXCTUS) return 1;
  int *by_order = (int *)malloc(sizeof(int) * nctu);
  for (int a = 0; a < nctu; a++) by_order[a] = -1;
  for (int a = 0; a < nctu; a++) {
    int o = ctus[a].coding_addr_plus1 ? ctus[a].coding_addr_plus1 - 1 : a;
    if (o < 0 || o >= nctu || by_order[o] >= 0) {
      free(by_order);
      return 1;
    }
    by_order[o] = a;
    c->ctu_order[a] = o;
  }
  for (int o = 0; o < nctu; o++) {
    int a = by_order[o];
    decompress_cu(c, &ctus[a], (a % c->ctus_w) * 64, (a / c->ctus_w) * 64, 0, 0);
  }
  free(by_order);
  memcpy(c->stage[0][0], c->rec[0], sizeof(int16_t) * c->w * c->h);
  memcpy(c->stage[0][1], c->rec[1], sizeof(int16_t) * c->cw * c->ch);
  memcpy(c->stage[0][2], c->rec[2], sizeof(int16_t) * c->cw * c->ch);
  /* TComLoopFilter::loopFilterPic, TLibCommon/TComLoopFilter.cpp:142-174: all vertical edges, then all horizontal */
  for (int dir = 0; dir < 2; dir++)
    for (int a = 0; a < nctu; a++) deblock_cu(c, &ctus[a], (a % c->ctus_w) * 64, (a / c->ctus_w) * 64, 0, 0, dir);
  /* SAO: TDecGop::filterPicture, TLibDecoder/TDecGop.cpp:184-203 */
  memcpy(dst->p[0], c->rec[0], sizeof(int16_t) * c->w * c->h);
  memcpy(dst->p[1], c->rec[1], sizeof(int16_t) * c->cw * c->ch);
  memcpy(dst->p[2], c->rec[2], sizeof(int16_t) * c->cw * c->ch);
  for (int comp = 0; comp < 3; comp++) {
    if (!(comp ? pp->sao_chroma : pp->sao_luma)) continue;
    for (int a = 0; a < nctu; a++) sao_ctu(c, &ctus[a], comp, (a % c->ctus_w) * 64, (a / c->ctus_w) * 64, c->rec[comp], dst->p[comp]);
  }
  /* TComSampleAdaptiveOffset::PCMLFDisableProcess / xPCMRestoration, TLibCommon/TComSampleAdaptiveOffset.cpp:1385-1515,
   * called after SAO (TLibDecoder/TDecGop.cpp:200): lossless and unfiltered IPCM units get their reconstruction back (the
   * reference keeps it in the CU's PCM buffer, TDecCu::xFillPCMBuffer, TLibDecoder/TDecCu.cpp:879-915) */
  if (pp->sao_luma || pp->sao_chroma)
    for (int y = 0; y < c->h; y += 4)
      for (int x = 0; x < c->w; x += 4) {
        PartRef r = part_at(c, x, y);
        if (!unfiltered(c, r.ctu, r.z)) continue;
        for (int j = 0; j < 4; j++)
          for (int i = 0; i < 4; i++) dst->p[0][(y + j) * c->w + x + i] = c->stage[0][0][(y + j) * c->w + x + i];
        for (int comp = 1; comp < 3; comp++)
          for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) dst->p[comp][(y / 2 + j) * c->cw + x / 2 + i] = c->stage[0][comp][(y / 2 + j) * c->cw + x / 2 + i];
      }
  dst->poc = pp->poc;
  return 0;
}

/* stage 0: before deblocking, 1: after deblocking, 2: final picture pic_id */
int hevcref_download(hevcref_ctx *c, int stage, int pic_id, int16_t *y, int16_t *cb, int16_t *cr) {
  int16_t *const *src;
  if (stage == 2) {
    if (pic_id < 0 || pic_id >= MAXPICS || !c->pics[pic_id].p[0]) return 1;
    src = c->pics[pic_id].p;
  } else {
    src = c->stage[stage];
  }
  memcpy(y, src[0], sizeof(int16_t) * c->w * c->h);
  memcpy(cb, src[1], sizeof(int16_t) * c->cw * c->ch);
  memcpy(cr, src[2], sizeof(int16_t) * c->cw * c->ch);
  return 0;
}

int hevcref_upload(hevcref_ctx *c, int pic_id, int poc, const int16_t *y, const int16_t *cb, const int16_t *cr) {
  Pic *p = get_pic(c, pic_id);
  if (!p) return 1;
  memcpy(p->p[0], y, sizeof(int16_t) * c->w * c->h);
  memcpy(p->p[1], cb, sizeof(int16_t) * c->cw * c->ch);
  memcpy(p->p[2], cr, sizeof(int16_t) * c->cw * c->ch);
  p->poc = poc;
  return 0;
}

/* Stand-alone kernels for unit-level parity tests. */
void hevcref_inverse_tu(hevcref_ctx *c, const hevcrecon_pic_params *pp, const int32_t *coef, int log2n, int comp, int qpy,
                        int use_dst, int tskip, int16_t *out) {
  c->pp = pp;
  c->cur_slice = NULL;
  inverse_tu(c, coef, log2n, comp, use_dst, qpy, use_dst, tskip, 0, out);
}
