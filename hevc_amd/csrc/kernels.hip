// kernels.hip — the gfx950 kernels of the picture-reconstruction engine.
//
// All arithmetic is integer and must match HM 9.1 bit for bit; each kernel cites the reference code
// whose results it reproduces (paths relative to source/Lib of the reference).  None of it is a
// translation: the reference walks CTU by CTU on one thread, here every kind of work runs over the
// whole picture (and over a batch of pictures, blockIdx.y) at once:
//
//   k_inter_tile one workgroup per 64x64 / 32x32 / 16x16 region of uniform motion: separable
//                interpolation through LDS, byte and 16-bit dot products
//   k_inter_blk  one lane per 4x4 luma block (+ its 2x2 chroma blocks) of the regions with mixed motion:
//                motion compensation straight from the per-4x4 motion records
//   k_transform  one lane per column/row of a transform block: de-quantisation + 2-D inverse
//                transform, result added to the picture (inter) or kept as residual (intra)
//   k_intra      one workgroup per CTU at a time (CTU staged in LDS, blocks level by level inside), one launch per batch:
//                resident workgroups draw CTUs in dependency order and wait for their neighbours on device flags
//   k_deblock    one lane per 4-sample edge segment; vertical edges, then horizontal edges
//   k_sao        one lane per 16 samples, out of place (deblocked picture -> final picture)
#ifdef HEVCRECON_EMULATE
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdlib>
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// Wave-level helpers.  hr_uniform: a value known to be equal in all lanes, moved to a scalar
// register.  hr_wave_sync: orders LDS traffic between the lanes of ONE wave (a wave executes its LDS
// instructions in order, so this only has to stop the compiler from moving accesses across it).
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ int hr_uniform(int v) { return v; }
__device__ __forceinline__ unsigned hr_uniform(unsigned v) { return v; }
__device__ __forceinline__ void hr_wave_sync() { emu::current_block().wave_barrier[emu::wave()].wait(); }
__device__ __forceinline__ int hr_readlane(int v, int lane) { return __shfl(v, lane); }
__device__ __forceinline__ unsigned hr_brev(unsigned v) {
  unsigned r = 0;
  for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
#else
__device__ __forceinline__ int hr_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned hr_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// value of lane `lane` (the same lane index in every lane of the wave), as a scalar
__device__ __forceinline__ int hr_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ void hr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ unsigned hr_brev(unsigned v) { return __brev(v); }
#endif

template <typename pel_t>
__device__ __forceinline__ const pel_t *plane_c(const void *p) { return (const pel_t *)p; }

// Pointers that a kernel loads from memory (picture planes out of the job descriptor) are generic to the compiler,
// and generic accesses are slower to issue and to wait for than global ones: hot loops say what they are.
#ifdef HEVCRECON_EMULATE
#define HR_GLOBAL
#else
#define HR_GLOBAL __attribute__((address_space(1)))
#endif

// Hand-over of data between workgroups of ONE launch (the intra kernel's CTUs).  The L1 of a compute unit is never
// refreshed by another unit's stores and plain stores may sit in the L2 of their chiplet, so:
//   producer  write-through stores (hr_store_through) -> every storing wave waits for its stores (hr_drain_stores)
//             -> workgroup barrier -> one lane raises the flag (hr_flag_raise)
//   consumer  one lane polls the flag (hr_flag_read, L1 bypassed) -> hr_acquire (drops this unit's L1 lines)
//             -> workgroup barrier -> plain loads
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ unsigned hr_ticket_take(unsigned *p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
__device__ __forceinline__ unsigned hr_flag_read(const unsigned *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
__device__ __forceinline__ void hr_flag_raise(unsigned *p) { __atomic_store_n(p, 1u, __ATOMIC_RELEASE); }
__device__ __forceinline__ void hr_store_through(unsigned long long *p, unsigned long long v) { *p = v; }
__device__ __forceinline__ void hr_store_through16(void *p, uint4 v) { *(uint4 *)p = v; }
__device__ __forceinline__ void hr_drain_stores() {}
__device__ __forceinline__ void hr_acquire() {}
__device__ __forceinline__ void hr_nap() {}
__device__ __forceinline__ unsigned long long hr_clock100() { return 0; }
#else
// Every shared word is accessed through a GLOBAL (address space 1) pointer at agent scope: pointers that kernels load
// from memory are generic, and generic (flat) accesses neither bypass the L1 reliably nor order like global ones.
typedef __attribute__((address_space(1))) unsigned hr_gu32;
typedef __attribute__((address_space(1))) unsigned long long hr_gu64;
__device__ __forceinline__ unsigned hr_ticket_take(unsigned *p) {
  return __hip_atomic_fetch_add((hr_gu32 *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned hr_flag_read(const unsigned *p) {
  return __hip_atomic_load((const hr_gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hr_flag_raise(unsigned *p) { __hip_atomic_store((hr_gu32 *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hr_store_through(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store((hr_gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes written through to memory in one fabric write (an 8-byte write-through store costs as much).  The
// compiler does not see the store: its data registers are only free again once the instruction has read them
// (hence the s_nop inside the string), and the wave counts it in vmcnt like any other store (hr_drain_stores).
__device__ __forceinline__ void hr_store_through16(void *p, uint4 v) {
  typedef unsigned hr_v4u __attribute__((ext_vector_type(4)));
  const hr_v4u d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void hr_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void hr_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// constant 100 MHz clock, for bounding waits by time rather than by iterations
__device__ __forceinline__ unsigned long long hr_clock100() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void hr_nap() { __builtin_amdgcn_s_sleep(4); }  // ~0.1 us between polls: a CTU on the critical chain starts that soon after its neighbour
#endif

// Core transform matrix of the 32-point inverse DCT: entry (k, i) is round(64*sqrt(2)*cos(pi*k*(2i+1)/64)) as
// standardised (row 0 is 64); the 16/8/4-point matrices are its rows 2k/4k/8k truncated to their first
// 16/8/4 entries.  Built at compile time from the 32 distinct magnitudes.  (The reference holds the
// expanded matrices in TLibCommon/TComRom.cpp:230-304.)
struct T32Table {
  int8_t v[32][32];
};
constexpr T32Table make_t32() {
  const int mag[33] = {0, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                       61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};  // index j: angle pi*j/64
  T32Table t = {};
  for (int k = 0; k < 32; k++)
    for (int i = 0; i < 32; i++) {
      int val = 64;
      if (k > 0) {
        int m = (k * (2 * i + 1)) & 127;
        if (m > 64) m = 128 - m;
        val = m > 32 ? -mag[64 - m] : mag[m];
      }
      t.v[k][i] = (int8_t)val;
    }
  return t;
}
// The same matrices as 16-bit pairs for the 2-way dot product: entry [log2 N - 2][k / 2][i] holds rows k and k + 1
// of the N-point matrix at column i.
struct TPairTable {
  unsigned v[4][16][32];
};
constexpr TPairTable make_tpairs() {
  const T32Table t = make_t32();
  TPairTable p = {};
  for (int l = 0; l < 4; l++) {
    const int n = 4 << l, r = 32 / n;
    for (int kp = 0; kp < n / 2; kp++)
      for (int i = 0; i < n; i++) {
        const int lo = t.v[2 * kp * r][i], hi = t.v[(2 * kp + 1) * r][i];
        p.v[l][kp][i] = ((unsigned)lo & 0xffffu) | (((unsigned)hi & 0xffffu) << 16);
      }
  }
  return p;
}
__constant__ TPairTable kTPtab = make_tpairs();
#define kTP kTPtab.v
// 4-point inverse DST used for intra luma 4x4 (TLibCommon/TComRom.cpp:316-322), rows (0,1) and (2,3) paired likewise.
#define HR_P16(a, b) (((unsigned)(a)&0xffffu) | (((unsigned)(b)&0xffffu) << 16))
__constant__ unsigned kDst4P[2][4] = {{HR_P16(29, 74), HR_P16(55, 74), HR_P16(74, 0), HR_P16(84, -74)},
                                      {HR_P16(84, 55), HR_P16(-29, -84), HR_P16(-74, 74), HR_P16(55, -29)}};
// chroma QP mapping (TLibCommon/TComRom.cpp:306-312)
__constant__ uint8_t kChromaScale[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
                                         20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35,
                                         36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};
// deblocking tables (TLibCommon/TComLoopFilter.cpp:63-71)
__constant__ uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  1,  1,  1,  1,  1,  1,  1,  1,  1,
                                2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9,  10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t kBeta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,
                                  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
                                  34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};

// ------------------------------------------------------------------------------------------------
// k_inter: motion-compensated prediction
// ------------------------------------------------------------------------------------------------
// Interpolation taps (TLibCommon/TComInterpolationFilter.cpp:55-73), packed one signed byte per tap so a
// lane can pick its row with selects instead of a divergent table read.
__device__ __forceinline__ void luma_taps(int frac, int c[8]) {
  // {0,0,0,64,0,0,0,0} {-1,4,-10,58,17,-5,1,0} {-1,4,-11,40,40,-11,4,-1} {0,1,-5,17,58,-10,4,-1}
  const int t1[8] = {-1, 4, -10, 58, 17, -5, 1, 0};
  const int t2[8] = {-1, 4, -11, 40, 40, -11, 4, -1};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int id = (k == 3) ? 64 : 0;
    c[k] = frac == 0 ? id : frac == 1 ? t1[k] : frac == 2 ? t2[k] : t1[7 - k];
  }
}
__device__ __forceinline__ void chroma_taps(int frac, int c[4]) {
  // rows 0..4 of the table; rows 5..7 are rows 3..1 mirrored
  const int t[5][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}};
  int f = frac <= 4 ? frac : 8 - frac;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int kk = frac <= 4 ? k : 3 - k;
    int v = t[0][0];
    v = f == 0 ? (kk == 0 ? t[0][0] : kk == 1 ? t[0][1] : kk == 2 ? t[0][2] : t[0][3]) : v;
    v = f == 1 ? (kk == 0 ? t[1][0] : kk == 1 ? t[1][1] : kk == 2 ? t[1][2] : t[1][3]) : v;
    v = f == 2 ? (kk == 0 ? t[2][0] : kk == 1 ? t[2][1] : kk == 2 ? t[2][2] : t[2][3]) : v;
    v = f == 3 ? (kk == 0 ? t[3][0] : kk == 1 ? t[3][1] : kk == 2 ? t[3][2] : t[3][3]) : v;
    v = f == 4 ? (kk == 0 ? t[4][0] : kk == 1 ? t[4][1] : kk == 2 ? t[4][2] : t[4][3]) : v;
    c[k] = v;
  }
}

// Separable interpolation of a BW x BH block with NT taps produces the 14-bit intermediate of the
// reference's bi-prediction path: horizontal pass (sum - (8192 << s1)) >> s1, s1 = bd - 8, then the
// vertical pass sum >> 6 (TLibCommon/TComInterpolationFilter.cpp:165-246 with isLast = false).
// Running a zero fraction through the identity filter {..64..} reproduces the reference's shortcut
// paths exactly (filterCopy and the single-pass cases), and the uni-prediction result follows from
// the same intermediate: clip((v + 8192 + (1 << (13 - bd))) >> (14 - bd)).
// Reference samples outside the picture are read with clamped coordinates, which equals the
// reference's replicated 80-sample border because clipped vectors never reach beyond it.

// ---- byte-packed fast path for 8-bit pictures -------------------------------------------------------
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ unsigned hr_alignbyte(unsigned hi, unsigned lo, int sh) {
  return (unsigned)((((unsigned long long)hi << 32) | lo) >> (8 * (sh & 3)));
}
__device__ __forceinline__ int hr_sdot4(unsigned a, unsigned b, int c) {
  for (int k = 0; k < 4; k++) c += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
  return c;
}
#else
__device__ __forceinline__ unsigned hr_alignbyte(unsigned hi, unsigned lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ int hr_sdot4(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
#endif
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
  return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)(d & 255) << 24);
}

// That arithmetic for 8-bit samples, organised for the hardware: a row of the reference
// window is fetched as aligned dwords and shifted into place (v_alignbyte), samples are biased to signed
// bytes (x ^ 0x80 = x - 128) and the horizontal filter is two 4-way byte dot products.  Because the taps sum
// to 64, sum(c_k * (p_k - 128)) = sum(c_k * p_k) - 8192, which IS the reference's 14-bit intermediate at
// bit depth 8 (offset 8192, shift 0).  Rows whose vertical tap is zero are skipped when the vertical
// fraction is zero.  Blocks whose window leaves the picture gather clamped bytes instead.
struct __attribute__((packed)) HrBytes12 {
  unsigned v[3];
};
struct __attribute__((packed)) HrBytes8 {
  unsigned v[2];
};
// 12 / 8 bytes from any address of a picture plane (one unaligned global load)
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ void hr_load12(const void *p, unsigned w[3]) {
  const HrBytes12 q = *(const HrBytes12 *)p;
  for (int k = 0; k < 3; k++) w[k] = q.v[k];
}
__device__ __forceinline__ void hr_load8(const void *p, unsigned w[2]) {
  const HrBytes8 q = *(const HrBytes8 *)p;
  for (int k = 0; k < 2; k++) w[k] = q.v[k];
}
#else
typedef unsigned hr_u32x4 __attribute__((ext_vector_type(4)));
typedef hr_u32x4 hr_u32x4_unaligned __attribute__((aligned(1)));
typedef unsigned hr_u32x3 __attribute__((ext_vector_type(3)));
typedef hr_u32x3 hr_u32x3_unaligned __attribute__((aligned(1)));
typedef unsigned hr_u32x2 __attribute__((ext_vector_type(2)));
typedef hr_u32x2 hr_u32x2_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ void hr_load12(const HR_GLOBAL void *p, unsigned w[3]) {
  const hr_u32x3 q = *(const HR_GLOBAL hr_u32x3_unaligned *)p;
  w[0] = q.x, w[1] = q.y, w[2] = q.z;
}
__device__ __forceinline__ void hr_load8(const HR_GLOBAL void *p, unsigned w[2]) {
  const hr_u32x2 q = *(const HR_GLOBAL hr_u32x2_unaligned *)p;
  w[0] = q.x, w[1] = q.y;
}
#endif

// Window of a BW x BH block for an NT-tap filter: ROWS rows of NW dwords, samples biased by 128.
template <int NT, int BW, int BH>
__device__ __forceinline__ void interp_u8_load(const HR_GLOBAL uint8_t *ref, int stride, int pw, int ph, int x0, int y0, bool v_identity,
                                               unsigned w[BH + NT - 1][(BW + NT - 1 + 3) / 4]) {
  constexpr int HALF = NT / 2 - 1;
  constexpr int NPIX = BW + NT - 1;
  constexpr int NW = (NPIX + 3) / 4;
  constexpr int ROWS = BH + NT - 1;
  const int xs = x0 - HALF, ys = y0 - HALF;
  const bool inside = xs >= 0 && xs + NPIX - 1 < pw && ys >= 0 && ys + ROWS - 1 < ph;
  // With a zero vertical fraction only the BH middle rows matter: the others re-read a middle row (same
  // cache line, no extra traffic) and are skipped by the filter.
  if (inside) {
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const int rr = v_identity ? clip3i(HALF, HALF + BH - 1, r) : r;
      const HR_GLOBAL uint8_t *rp = ref + (size_t)(ys + rr) * stride + xs;  // unaligned loads
      if constexpr (NW == 3) {
        hr_load12(rp, w[r]);
#pragma unroll
        for (int k = 0; k < 3; k++) w[r][k] ^= 0x80808080u;
      } else {
        hr_load8(rp, w[r]);
#pragma unroll
        for (int k = 0; k < 2; k++) w[r][k] ^= 0x80808080u;
      }
    }
  } else {
    int xb = xs;  // (see tile_row_load: keeps the clamping arithmetic of this rare path inside the branch)
#ifndef HEVCRECON_EMULATE
    asm volatile("" : "+v"(xb));
#endif
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const HR_GLOBAL uint8_t *row = ref + (size_t)clip3i(0, ph - 1, ys + r) * stride;
#pragma unroll
      for (int k = 0; k < NW; k++) w[r][k] = 0;
#pragma unroll
      for (int i = 0; i < NPIX; i++) w[r][i >> 2] |= (unsigned)row[clip3i(0, pw - 1, xb + i)] << (8 * (i & 3));
#pragma unroll
      for (int k = 0; k < NW; k++) w[r][k] ^= 0x80808080u;
    }
  }
}

// The arithmetic above for 8-bit samples, organised for the hardware: samples biased to signed
// bytes (x ^ 0x80 = x - 128), the horizontal filter is two 4-way byte dot products.  Because the taps sum
// to 64, sum(c_k * (p_k - 128)) = sum(c_k * p_k) - 8192, which IS the reference's 14-bit intermediate at
// bit depth 8 (offset 8192, shift 0).  Rows whose vertical tap is zero are skipped when the vertical
// fraction is zero.
template <int NT, int BW, int BH>
__device__ __forceinline__ void interp_u8_filter(const unsigned w[BH + NT - 1][(BW + NT - 1 + 3) / 4], unsigned tlo, unsigned thi,
                                                 const int *cy, bool v_identity, int out[BH][BW]) {
  constexpr int HALF = NT / 2 - 1;
  constexpr int NW = (BW + NT - 1 + 3) / 4;
  constexpr int ROWS = BH + NT - 1;
  int acc[BH][BW];
#pragma unroll
  for (int j = 0; j < BH; j++)
#pragma unroll
    for (int i = 0; i < BW; i++) acc[j][i] = 0;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    if (v_identity && (r < HALF || r > HALF + BH - 1)) continue;
#pragma unroll
    for (int i = 0; i < BW; i++) {
      const unsigned lo = i ? hr_alignbyte(w[r][1], w[r][0], i) : w[r][0];
      int h = hr_sdot4(lo, tlo, 0);
      if (NT == 8) {
        const unsigned hi = i ? hr_alignbyte(w[r][NW - 1], w[r][1], i) : w[r][1];
        h = hr_sdot4(hi, thi, h);
      }
#pragma unroll
      for (int j = 0; j < BH; j++) {
        const int k = r - j;
        if (k >= 0 && k < NT) acc[j][i] += __mul24(cy[k], h);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BH; j++)
#pragma unroll
    for (int i = 0; i < BW; i++) out[j][i] = acc[j][i] >> 6;
}

struct __attribute__((packed)) HrBytes16 {
  unsigned v[4];
};
// 16 bytes from any address of a picture plane (one unaligned global_load_dwordx4)
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ void hr_load16(const void *p, unsigned w[4]) {
  const HrBytes16 q = *(const HrBytes16 *)p;
  for (int k = 0; k < 4; k++) w[k] = q.v[k];
}
__device__ __forceinline__ void hr_store16(void *p, unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned *q = (unsigned *)p;
  q[0] = a, q[1] = b, q[2] = c, q[3] = d;
}
#else
__device__ __forceinline__ void hr_load16(const HR_GLOBAL void *p, unsigned w[4]) {
  const hr_u32x4 q = *(const HR_GLOBAL hr_u32x4_unaligned *)p;
  w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
}
// 16 bytes to a 16-byte aligned address
__device__ __forceinline__ void hr_store16(HR_GLOBAL void *p, unsigned a, unsigned b, unsigned c, unsigned d) {
  hr_u32x4 q;
  q.x = a, q.y = b, q.z = c, q.w = d;
  *(HR_GLOBAL hr_u32x4 *)p = q;
}
#endif

// 2-way 16-bit dot product with 32-bit accumulation (v_dot2_i32_i16)
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ int hr_sdot2(unsigned a, unsigned b, int c) {
  return c + (int)(int16_t)(a & 0xffff) * (int)(int16_t)(b & 0xffff) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16);
}
#else
typedef short hr_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int hr_sdot2(unsigned a, unsigned b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(hr_short2, a), __builtin_bit_cast(hr_short2, b), c, false);
}
#endif

// Packed tap tables (TLibCommon/TComInterpolationFilter.cpp:55-73), read with wave-uniform indices:
// four signed bytes per dword for the horizontal byte dot products, 16-bit pairs for the vertical ones
// ("even": taps (0,1)(2,3)...; "odd": the same taps shifted by one row, (-,0)(1,2)...(7,-)).
#define HR_B4(a, b, c, d) (((unsigned)(a)&255u) | (((unsigned)(b)&255u) << 8) | (((unsigned)(c)&255u) << 16) | (((unsigned)(d)&255u) << 24))
#define HR_P2(a, b) (((unsigned)(a)&0xffffu) | ((unsigned)(b) << 16))
#define HR_L8(f, a, b, c, d, e, g, h, i)                                                                 \
  { HR_B4(a, b, c, d), HR_B4(e, g, h, i), HR_P2(a, b), HR_P2(c, d), HR_P2(e, g), HR_P2(h, i), HR_P2(0, a), HR_P2(b, c), \
    HR_P2(d, e), HR_P2(g, h), HR_P2(i, 0), 0 }
#define HR_C4(a, b, c, d) { HR_B4(a, b, c, d), HR_P2(a, b), HR_P2(c, d), HR_P2(0, a), HR_P2(b, c), HR_P2(d, 0), 0, 0 }
__constant__ unsigned kLumaTab[4][12] = {HR_L8(0, 0, 0, 0, 64, 0, 0, 0, 0), HR_L8(1, -1, 4, -10, 58, 17, -5, 1, 0),
                                         HR_L8(2, -1, 4, -11, 40, 40, -11, 4, -1), HR_L8(3, 0, 1, -5, 17, 58, -10, 4, -1)};
__constant__ unsigned kChromaTab[8][8] = {HR_C4(0, 64, 0, 0),    HR_C4(-2, 58, 10, -2), HR_C4(-4, 54, 16, -2), HR_C4(-6, 46, 28, -4),
                                          HR_C4(-4, 36, 36, -4), HR_C4(-4, 28, 46, -6), HR_C4(-2, 16, 54, -4), HR_C4(-2, 10, 58, -2)};


// The same for 16-bit planes: a window row is NW dwords of two samples each, the horizontal filter runs as 2-way
// 16-bit dot products: an even output uses the tap pairs (0,1)(2,3).., an odd output the pairs shifted by one sample
// ((-,0)(1,2)..(7,-)), so no sample is ever re-aligned.  pairs[]: the even pairs, then the odd pairs (the layout of
// kLumaTab[f] + 2 and kChromaTab[f] + 1).
template <int NT, int BW, int BH>
__device__ __forceinline__ void interp_u16_load(const HR_GLOBAL uint16_t *ref, int stride, int pw, int ph, int x0, int y0, bool v_identity,
                                                unsigned w[BH + NT - 1][(BW + NT) / 2]) {
  constexpr int HALF = NT / 2 - 1;
  constexpr int NPIX = BW + NT - 1;
  constexpr int NW = (BW + NT) / 2;
  constexpr int ROWS = BH + NT - 1;
  const int xs = x0 - HALF, ys = y0 - HALF;
  const bool inside = xs >= 0 && xs + 2 * NW - 1 < pw && ys >= 0 && ys + ROWS - 1 < ph;
  if (inside) {
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const int rr = v_identity ? clip3i(HALF, HALF + BH - 1, r) : r;
      const HR_GLOBAL uint16_t *rp = ref + (size_t)(ys + rr) * stride + xs;  // unaligned loads
      if constexpr (NW == 6) {
        hr_load16(rp, w[r]);
        hr_load8(rp + 8, w[r] + 4);
      } else {
        hr_load12(rp, w[r]);
      }
    }
  } else {
    int xb = xs;  // (see tile_row_load: keeps the clamping arithmetic of this rare path inside the branch)
#ifndef HEVCRECON_EMULATE
    asm volatile("" : "+v"(xb));
#endif
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      const HR_GLOBAL uint16_t *row = ref + (size_t)clip3i(0, ph - 1, ys + r) * stride;
#pragma unroll
      for (int k = 0; k < NW; k++) w[r][k] = 0;
#pragma unroll
      for (int i = 0; i < NPIX; i++) w[r][i >> 1] |= (unsigned)row[clip3i(0, pw - 1, xb + i)] << (16 * (i & 1));
    }
  }
}

template <int NT, int BW, int BH>
__device__ __forceinline__ void interp_u16_filter(const unsigned w[BH + NT - 1][(BW + NT) / 2], const unsigned *pairs, const int *cy,
                                                  bool v_identity, int bd, int out[BH][BW]) {
  constexpr int HALF = NT / 2 - 1;
  constexpr int ROWS = BH + NT - 1;
  constexpr int NE = NT / 2, NO = NT / 2 + 1;  // dot products of an even / odd output
  const int s1 = bd - 8, init = -(8192 << s1);
  unsigned te[NE], to[NO];
#pragma unroll
  for (int k = 0; k < NE; k++) te[k] = pairs[k];
#pragma unroll
  for (int k = 0; k < NO; k++) to[k] = pairs[NE + k];
  int acc[BH][BW];
#pragma unroll
  for (int j = 0; j < BH; j++)
#pragma unroll
    for (int i = 0; i < BW; i++) acc[j][i] = 0;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    if (v_identity && (r < HALF || r > HALF + BH - 1)) continue;
#pragma unroll
    for (int i = 0; i < BW; i++) {
      const int d = i >> 1;
      int sum = init;
      if (i & 1) {
#pragma unroll
        for (int k = 0; k < NO; k++) sum = hr_sdot2(w[r][d + k], to[k], sum);
      } else {
#pragma unroll
        for (int k = 0; k < NE; k++) sum = hr_sdot2(w[r][d + k], te[k], sum);
      }
      const int h = sum >> s1;
#pragma unroll
      for (int j = 0; j < BH; j++) {
        const int k = r - j;
        if (k >= 0 && k < NT) acc[j][i] += __mul24(cy[k], h);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BH; j++)
#pragma unroll
    for (int i = 0; i < BW; i++) out[j][i] = acc[j][i] >> 6;
}

// Final rounding of a prediction, one component: out = clip(((w0*P0 + w1*P1 + add) >> sh) + post), P the 14-bit
// intermediates (offset by -8192) of the lists in use.
//   default, one list:  clip((P + 8192 + rnd) >> (14 - bd))                       (TComInterpolationFilter, isLast)
//   default, two lists: TComYuv::addAvg (TLibCommon/TComYuv.cpp:520-584)
//   explicit weights:   TComWeightPrediction::addWeightUni / addWeightBi with the parameters of getWpScaling
//                       (TLibCommon/TComWeightPrediction.cpp:43-59,79-257,271-334)
struct PredRound {
  int w[2], add, sh, post;
};
__device__ __forceinline__ PredRound pred_round(const WpTables *wp, int r0, int r1, int comp, int bd) {
  const int head = 14 - bd;
  const bool bi = r0 >= 0 && r1 >= 0;
  PredRound R;
  if (!wp) {
    R.w[0] = R.w[1] = 1;
    R.sh = bi ? head + 1 : head;
    R.add = bi ? 16384 + (1 << head) : 8192 + (head ? 1 << (head - 1) : 0);
    R.post = 0;
  } else if (bi) {
    R.w[0] = wp->w[0][r0][comp];
    R.w[1] = wp->w[1][r1][comp];
    R.sh = wp->log2d[0][r0][comp] + 1 + head;
    const int off = (wp->o[0][r0][comp] + wp->o[1][r1][comp]) * (1 << (bd - 8));
    R.add = (R.w[0] + R.w[1]) * 8192 + (1 << (R.sh - 1)) + off * (1 << (R.sh - 1));
    R.post = 0;
  } else {
    const int l = r0 >= 0 ? 0 : 1, r = l ? r1 : r0;
    const int w = wp->w[l][r][comp];
    R.w[0] = l ? 0 : w;  // (not R.w[l]: an array indexed by a run-time value lives in scratch memory)
    R.w[1] = l ? w : 0;
    R.sh = wp->log2d[l][r][comp] + head;
    R.add = w * 8192 + (R.sh ? 1 << (R.sh - 1) : 0);
    R.post = wp->o[l][r][comp] * (1 << (bd - 8));
  }
  return R;
}

// Motion-compensated prediction of one 4x4 luma unit (and its 2x2 chroma units) straight from its motion record.
// WP: the batch holds pictures with explicit weighted prediction (J.wp set for those).
template <typename pel_t, bool WP>
__device__ __forceinline__ void inter_unit(const PicJob &J, const SeqInfo &s, int ux, int uy) {
  const Part4 p = J.rec4[(size_t)uy * s.uw + ux];
  if (p.flags & P4_INTRA) return;
  const int x = ux * 4, y = uy * 4;
  // TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3207-3219): the range depends on the coding unit's origin.
  const int cusize = 64 >> (p.flags & P4_DEPTH_MASK);
  const int cux = x & ~(cusize - 1), cuy = y & ~(cusize - 1);
  const int hmax = (s.w + 8 - cux - 1) * 4, hmin = (-64 - 8 - cux + 1) * 4;
  const int vmax = (s.h + 8 - cuy - 1) * 4, vmin = (-64 - 8 - cuy + 1) * 4;
  bool use0 = p.ref_idx[0] >= 0, use1 = p.ref_idx[1] >= 0;
  // TComPrediction::xCheckIdenticalMotion (TLibCommon/TComPrediction.cpp:396-411); not with weighted bi-prediction
  const WpTables *wp = WP ? J.wp : nullptr;
  if (use0 && use1 && J.slice_type == 0 && !wp && J.ref_poc[0][p.ref_idx[0]] == J.ref_poc[1][p.ref_idx[1]] && p.mv[0][0] == p.mv[1][0] &&
      p.mv[0][1] == p.mv[1][1])
    use1 = false;
  const int r0 = use0 ? p.ref_idx[0] : -1, r1 = use1 ? p.ref_idx[1] : -1;
  const PredRound Ry = pred_round(wp, r0, r1, 0, s.bd_y);
  const PredRound Rc[2] = {pred_round(wp, r0, r1, 1, s.bd_c), pred_round(wp, r0, r1, 2, s.bd_c)};
  int ly[4][4], lc[2][2][2];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) ly[j][i] = 0;
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 2; i++) lc[c][j][i] = 0;
  if constexpr (sizeof(pel_t) == 1) {
    // 8-bit pictures: the windows of both lists and all three planes are fetched before anything is filtered,
    // so a lane pays one trip to memory instead of six dependent ones
    unsigned wy[2][11][3], wc[2][2][5][2];
    int mvx[2], mvy[2];
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (!(l ? use1 : use0)) continue;
      const int ri = p.ref_idx[l];
      mvx[l] = min(hmax, max(hmin, (int)p.mv[l][0]));
      mvy[l] = min(vmax, max(vmin, (int)p.mv[l][1]));
      interp_u8_load<8, 4, 4>((const HR_GLOBAL uint8_t *)J.ref[l][ri][0], s.stride_y, s.w, s.h, x + (mvx[l] >> 2), y + (mvy[l] >> 2),
                              (mvy[l] & 3) == 0, wy[l]);
#pragma unroll
      for (int c = 0; c < 2; c++)
        interp_u8_load<4, 2, 2>((const HR_GLOBAL uint8_t *)J.ref[l][ri][1 + c], s.stride_c, s.cw, s.ch, (x >> 1) + (mvx[l] >> 3),
                                (y >> 1) + (mvy[l] >> 3), (mvy[l] & 7) == 0, wc[l][c]);
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if (!(l ? use1 : use0)) continue;
      {
        int tx[8], ty[8], o[4][4];
        luma_taps(mvx[l] & 3, tx);
        luma_taps(mvy[l] & 3, ty);
        interp_u8_filter<8, 4, 4>(wy[l], pack4(tx[0], tx[1], tx[2], tx[3]), pack4(tx[4], tx[5], tx[6], tx[7]), ty, (mvy[l] & 3) == 0, o);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) ly[j][i] += WP ? __mul24(Ry.w[l], o[j][i]) : o[j][i];
      }
      {
        int tx[4], ty[4];
        chroma_taps(mvx[l] & 7, tx);
        chroma_taps(mvy[l] & 7, ty);
#pragma unroll
        for (int c = 0; c < 2; c++) {
          int o[2][2];
          interp_u8_filter<4, 2, 2>(wc[l][c], pack4(tx[0], tx[1], tx[2], tx[3]), 0u, ty, (mvy[l] & 7) == 0, o);
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) lc[c][j][i] += WP ? __mul24(Rc[c].w[l], o[j][i]) : o[j][i];
        }
      }
    }
  } else {
    // 16-bit pictures: one list at a time (a list's windows are 96 registers), its three windows fetched before
    // anything is filtered
    for (int l = 0; l < 2; l++) {
      if (!(l ? use1 : use0)) continue;
      const int ri = p.ref_idx[l];
      const int mvx = min(hmax, max(hmin, (int)p.mv[l][0])), mvy = min(vmax, max(vmin, (int)p.mv[l][1]));
      unsigned wy[11][6], wc[2][5][3];
      interp_u16_load<8, 4, 4>((const HR_GLOBAL uint16_t *)J.ref[l][ri][0], s.stride_y, s.w, s.h, x + (mvx >> 2), y + (mvy >> 2),
                               (mvy & 3) == 0, wy);
#pragma unroll
      for (int c = 0; c < 2; c++)
        interp_u16_load<4, 2, 2>((const HR_GLOBAL uint16_t *)J.ref[l][ri][1 + c], s.stride_c, s.cw, s.ch, (x >> 1) + (mvx >> 3),
                                 (y >> 1) + (mvy >> 3), (mvy & 7) == 0, wc[c]);
      {
        int ty[8], o[4][4];
        luma_taps(mvy & 3, ty);
        interp_u16_filter<8, 4, 4>(wy, kLumaTab[mvx & 3] + 2, ty, (mvy & 3) == 0, s.bd_y, o);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) ly[j][i] += WP ? __mul24(Ry.w[l], o[j][i]) : o[j][i];
      }
      {
        int ty[4];
        chroma_taps(mvy & 7, ty);
#pragma unroll
        for (int c = 0; c < 2; c++) {
          int o[2][2];
          interp_u16_filter<4, 2, 2>(wc[c], kChromaTab[mvx & 7] + 1, ty, (mvy & 7) == 0, s.bd_c, o);
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) lc[c][j][i] += WP ? __mul24(Rc[c].w[l], o[j][i]) : o[j][i];
        }
      }
    }
  }
  {
    const int maxv = (1 << s.bd_y) - 1;
    const int sh = Ry.sh, off = Ry.add, post = Ry.post;
    pel_t *dst = (pel_t *)J.rec[0] + (size_t)y * s.stride_y + x;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] = clip3i(0, maxv, ((ly[j][i] + off) >> sh) + post);
      if constexpr (sizeof(pel_t) == 1) {
        *(unsigned *)dst = pack4(v[0], v[1], v[2], v[3]);  // x is a multiple of 4 and rows are 128-byte aligned
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i] = (pel_t)v[i];
      }
      dst += s.stride_y;
    }
  }
  {
    const int maxv = (1 << s.bd_c) - 1;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int sh = Rc[c].sh, off = Rc[c].add, post = Rc[c].post;
      pel_t *dst = (pel_t *)J.rec[1 + c] + (size_t)(y >> 1) * s.stride_c + (x >> 1);
#pragma unroll
      for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int i = 0; i < 2; i++) dst[i] = (pel_t)clip3i(0, maxv, ((lc[c][j][i] + off) >> sh) + post);
        dst += s.stride_c;
      }
    }
  }
}

// k_inter_blk: the 8x8 blocks of mixed motion (and, in 16-bit pictures, every 8x8 block outside the uniform regions).
// One wave per sixteen blocks, one lane per 4x4 unit.
template <typename pel_t, bool WP>
__global__ __launch_bounds__(256) void k_inter_blk(const PicJob *jobs, SeqInfo s) {
  const PicJob &J = jobs[blockIdx.y];
  const int r = blockIdx.x * 64 + (threadIdx.x >> 2);
  if (r >= J.n_mc_blk) return;
  const McBlk b = J.mc_blk[r];
  const int u = threadIdx.x & 3;
  const int ux = (b.x >> 2) + (u & 1), uy = (b.y >> 2) + (u >> 1);
  if (ux >= s.uw || uy >= s.uh) return;
  inter_unit<pel_t, WP>(J, s, ux, uy);
}

// ---- k_inter_tile: regions with uniform motion -----------------------------------------------------------
// One workgroup predicts a TS x TS luma region (TS = 64, 32, 16) and its two chroma regions as single blocks.
//   phase H  work items of two window rows x eight outputs: the lane fetches 16 bytes of each row straight
//            from the reference picture (unaligned loads; the lanes of a row are contiguous, so they
//            coalesce; all of a lane's loads are issued before the first is used), filters horizontally and
//            stores the 16-bit intermediates of the two rows packed in one dword, P[pair][x];
//   phase V  one lane per column and four rows: the vertical filter as 2-way 16-bit dot products over the
//            packed row pairs (odd rows use the tap pairs shifted by one), both lists accumulated in
//            registers, then the uni- or bi-prediction rounding; the bytes go to an LDS tile;
//   store    the tile is written out as dwords.
// Arithmetic as in interp_u8: samples biased by 128 so the horizontal sums ARE the reference's 14-bit
// intermediates; TComInterpolationFilter.cpp:165-246, TComPrediction.cpp:413-507, TComYuv.cpp:520-584.
// Final rounding of the tile kernels: clip((v + off) >> sh) to a byte.  The shifted value is hidden from the
// optimiser on purpose: hipcc (ROCm 7.2) otherwise fuses "shift, clamp to 0..255, pack two bytes" into
// v_ashr_pk_u8_i32 and then ORs the other two bytes into the upper half of its result register, which that
// instruction does not clear on gfx950 (the upper bytes of a negative input came out as 0xFF).
__device__ __forceinline__ int tile_px(int v, int off, int sh, int maxv = 255, int post = 0) {
  int x = ((v + off) >> sh) + post;
#ifndef HEVCRECON_EMULATE
  asm volatile("" : "+v"(x));
#endif
  return clip3i(0, maxv, x);
}
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
#else
// low halves of two registers in one instruction (v_perm_b32: bytes 0,1 of lo, then bytes 0,1 of hi)
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }
#endif

// 16 bytes of a picture row starting at sample x0; without `fast` every sample position is clamped into the row
__device__ __forceinline__ void tile_row_load(const HR_GLOBAL uint8_t *row, int x0, int pw, bool fast, unsigned w[4]) {
  if (fast) {
    hr_load16(row + x0, w);
  } else {
    // (the volatile no-op pins the clamping arithmetic of this rare path inside the branch; left to itself the
    // compiler computes all 16 clamped positions ahead of the branch, for every lane of every item)
    int xb = x0;
#ifndef HEVCRECON_EMULATE
    asm volatile("" : "+v"(xb));
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i >> 2] |= (unsigned)row[clip3i(0, pw - 1, xb + i)] << (8 * (i & 3));
  }
}

// 16 samples (two 16-byte loads) of a row of a 16-bit picture, two samples per dword
__device__ __forceinline__ void tile_row_load(const HR_GLOBAL uint16_t *row, int x0, int pw, bool fast, unsigned w[8]) {
  if (fast) {
    hr_load16(row + x0, w);
    hr_load16(row + x0 + 8, w + 4);
  } else {
    int xb = x0;
#ifndef HEVCRECON_EMULATE
    asm volatile("" : "+v"(xb));
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i >> 1] |= (unsigned)row[clip3i(0, pw - 1, xb + i)] << (16 * (i & 1));
  }
}

// first product of a dot-product chain.  (Not inline assembly with a literal zero accumulator: results of
// the dot instructions need wait states before another VALU instruction reads them on this hardware, which the
// compiler only inserts for instructions it has emitted itself.)
__device__ __forceinline__ int hr_sdot4z(unsigned a, unsigned b) { return hr_sdot4(a, b, 0); }
__device__ __forceinline__ int hr_sdot2z(unsigned a, unsigned b) { return hr_sdot2(a, b, 0); }

// A region's description is the same for every lane of its workgroup: held in scalar registers, so that everything
// derived from it (tap tables, reference planes, which paths run) is scalar work and scalar branches as well.
struct McTileU {
  int x, y, mvx[2], mvy[2], ref[2];
};
__device__ __forceinline__ McTileU load_tile_uniform(const McTile *p) {
  const uint4 q = *(const uint4 *)p;  // 16 bytes: x, y | mv[0] | mv[1] | ref[0], ref[1], pad
  const unsigned w0 = hr_uniform(q.x), w1 = hr_uniform(q.y), w2 = hr_uniform(q.z), w3 = hr_uniform(q.w);
  McTileU t;
  t.x = (int)(w0 & 0xffffu);
  t.y = (int)(w0 >> 16);
  t.mvx[0] = (int)(int16_t)(w1 & 0xffffu);
  t.mvy[0] = (int)(int16_t)(w1 >> 16);
  t.mvx[1] = (int)(int16_t)(w2 & 0xffffu);
  t.mvy[1] = (int)(int16_t)(w2 >> 16);
  t.ref[0] = (int)(int8_t)(w3 & 0xffu);
  t.ref[1] = (int)(int8_t)((w3 >> 8) & 0xffu);
  return t;
}

template <typename pel_t, int TS, int NTHR, bool WP>
__global__ __launch_bounds__(NTHR, (sizeof(pel_t) == 1 ? 8 : 4)) void k_inter_tile(const PicJob *jobs, SeqInfo s) {
  constexpr bool W16 = sizeof(pel_t) == 2;  // 16-bit planes: 16-bit dot products in the horizontal pass too
  constexpr int WD = W16 ? 8 : 4;           // dwords fetched per window row and work item (16 samples)
  constexpr int CS = TS / 2;                                     // chroma region size
  constexpr int NPY = (TS + 8) / 2, OY = TS / 8, TY = NPY * OY;  // luma: row pairs, 8-output groups per row, items
  constexpr int NPC = (CS + 4) / 2, OC = CS / 8, TC1 = NPC * OC, TC = 2 * TC1;  // chroma, both components
  // With SEG the chroma items of a list start at a multiple of 64, so that a wave never mixes luma and chroma
  // and everything that depends on the class stays in scalar registers.
  constexpr bool SEG = TS >= 64;
  constexpr int LYP = SEG ? (TY + 63) / 64 * 64 : TY;
  constexpr int HIT = (LYP + TC + NTHR - 1) / NTHR;  // H items per lane and list
  constexpr int LVL = TS == 64 ? 0 : TS == 32 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) unsigned P_y[2][NPY * TS];
  __shared__ __attribute__((aligned(16))) unsigned P_c[2][2 * NPC * CS];
  const PicJob &J = jobs[blockIdx.y];
  // XCD-contiguous order: workgroups go to the 8 XCDs round-robin; consecutive regions share cache lines
  const int n = J.n_mc_tile[LVL], per_xcd = (n + 7) >> 3;
  const int idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || idx >= n) return;
  const McTileU t = load_tile_uniform(&J.mc_tile[LVL][idx]);
  const int tid = threadIdx.x;
  const int rx = t.x, ry = t.y;
  // a list with a zero vertical fraction needs no vertical filter: its rows outside the block are never read
  const bool vid_y[2] = {(t.mvy[0] & 3) == 0, (t.mvy[1] & 3) == 0};
  const bool vid_c[2] = {(t.mvy[0] & 7) == 0, (t.mvy[1] & 7) == 0};

  // ---- phase H ----
  // item -> (luma/chroma, component, row pair p, output group o) of list l
#define MT_DECODE                                                                                \
  const int item = tid + it * NTHR;                                                              \
  const bool chroma = (SEG ? hr_uniform(item & ~63) : item) >= LYP;                              \
  const int k = item - (chroma ? LYP : 0);                                                       \
  const int comp = chroma && k >= TC1 ? 1 : 0, kk = k - comp * TC1;                              \
  const int p = chroma ? kk / OC : kk / OY, o = chroma ? kk % OC : kk % OY;                      \
  const bool live = t.ref[l] >= 0 && k < (chroma ? TC : TY) &&                                   \
                    !(chroma ? vid_c[l] && p > CS / 2 : vid_y[l] && (p < 1 || p > TS / 2 + 1));
  unsigned w[2][HIT][2][WD];
#pragma unroll
  for (int l = 0; l < 2; l++) {
    const int ri = t.ref[l] >= 0 ? t.ref[l] : 0;
    const int mvx = t.mvx[l], mvy = t.mvy[l];
    const int wx0 = rx + (mvx >> 2) - 3, wy0 = ry + (mvy >> 2) - 3;
    const int cwx0 = (rx >> 1) + (mvx >> 3) - 1, cwy0 = (ry >> 1) + (mvy >> 3) - 1;
    const bool fast_y = wx0 >= 0 && wx0 + TS + 6 < s.w, fast_c = cwx0 >= 0 && cwx0 + CS + 2 < s.cw;
#pragma unroll
    for (int it = 0; it < HIT; it++) {
      MT_DECODE
      if (live) {
        const HR_GLOBAL pel_t *base = (const HR_GLOBAL pel_t *)(chroma ? (comp ? J.ref[l][ri][2] : J.ref[l][ri][1]) : J.ref[l][ri][0]);
        const int stride = chroma ? s.stride_c : s.stride_y, pw = chroma ? s.cw : s.w, ph = chroma ? s.ch : s.h;
        const int x0 = (chroma ? cwx0 : wx0) + 8 * o, y0 = chroma ? cwy0 : wy0;
        const int last = chroma ? CS + 2 : TS + 6;
        const bool fast = chroma ? fast_c : fast_y;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          const int r = min(2 * p + rr, last);
          tile_row_load(base + (size_t)clip3i(0, ph - 1, y0 + r) * stride, x0, pw, fast, w[l][it][rr]);
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < 2; l++) {
    const unsigned *lt = kLumaTab[t.mvx[l] & 3], *ct = kChromaTab[t.mvx[l] & 7];
    const unsigned c_lo = lt[0], c_hi = lt[1], c4 = ct[0];
#pragma unroll
    for (int it = 0; it < HIT; it++) {
      MT_DECODE
      if (live) {
        int h[2][8];
        if constexpr (W16) {
          // 16-bit samples: (sum - (8192 << s1)) >> s1 with s1 = bit depth - 8; even outputs use the tap pairs
          // (0,1)(2,3).., odd outputs the pairs shifted by one sample, so no sample ever has to be re-aligned
          const int s1 = (chroma ? s.bd_c : s.bd_y) - 8, init = -(8192 << s1);
#pragma unroll
          for (int rr = 0; rr < 2; rr++) {
            const unsigned *x = w[l][it][rr];
            if (!chroma) {
#pragma unroll
              for (int i = 0; i < 8; i++) {
                const int d = i >> 1;
                const int sum = (i & 1) ? hr_sdot2(x[d + 4], lt[10], hr_sdot2(x[d + 3], lt[9], hr_sdot2(x[d + 2], lt[8], hr_sdot2(x[d + 1], lt[7], hr_sdot2(x[d], lt[6], init)))))
                                        : hr_sdot2(x[d + 3], lt[5], hr_sdot2(x[d + 2], lt[4], hr_sdot2(x[d + 1], lt[3], hr_sdot2(x[d], lt[2], init))));
                h[rr][i] = sum >> s1;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 8; i++) {
                const int d = i >> 1;
                const int sum = (i & 1) ? hr_sdot2(x[d + 2], ct[5], hr_sdot2(x[d + 1], ct[4], hr_sdot2(x[d], ct[3], init)))
                                        : hr_sdot2(x[d + 1], ct[2], hr_sdot2(x[d], ct[1], init));
                h[rr][i] = sum >> s1;
              }
            }
          }
        } else if ((chroma ? t.mvx[l] & 7 : t.mvx[l] & 3) == 0) {
          // whole-sample horizontal position (TComInterpolationFilter::filterCopy, or the identity tap): the
          // intermediate is the sample itself, 64 * (p - 128); output i is window byte i + 3 (luma) / i + 1 (chroma)
#pragma unroll
          for (int rr = 0; rr < 2; rr++) {
            unsigned x[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = w[l][it][rr][i] ^ 0x80808080u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
              const int jl = i + 3, jc = i + 1;
              const int vl = (int)(int8_t)(x[jl >> 2] >> (8 * (jl & 3))), vc = (int)(int8_t)(x[jc >> 2] >> (8 * (jc & 3)));
              h[rr][i] = (chroma ? vc : vl) * 64;
            }
          }
        } else if (!chroma) {
#pragma unroll
          for (int rr = 0; rr < 2; rr++) {
            unsigned x[4], a[12];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = w[l][it][rr][i] ^ 0x80808080u;
#pragma unroll
            for (int i = 0; i < 12; i++) a[i] = (i & 3) ? hr_alignbyte(x[(i >> 2) + 1], x[i >> 2], i & 3) : x[i >> 2];
#pragma unroll
            for (int i = 0; i < 8; i++) h[rr][i] = hr_sdot4(a[i + 4], c_hi, hr_sdot4z(a[i], c_lo));
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 2; rr++) {
            unsigned x[3];
#pragma unroll
            for (int i = 0; i < 3; i++) x[i] = w[l][it][rr][i] ^ 0x80808080u;
#pragma unroll
            for (int i = 0; i < 8; i++) h[rr][i] = hr_sdot4z((i & 3) ? hr_alignbyte(x[(i >> 2) + 1], x[i >> 2], i & 3) : x[i >> 2], c4);
          }
        }
        uint4 *dst = chroma ? (uint4 *)&P_c[l][comp * (NPC * CS) + p * CS + 8 * o] : (uint4 *)&P_y[l][p * TS + 8 * o];
        dst[0] = make_uint4(pack16(h[0][0], h[1][0]), pack16(h[0][1], h[1][1]), pack16(h[0][2], h[1][2]), pack16(h[0][3], h[1][3]));
        dst[1] = make_uint4(pack16(h[0][4], h[1][4]), pack16(h[0][5], h[1][5]), pack16(h[0][6], h[1][6]), pack16(h[0][7], h[1][7]));
      }
    }
  }
#undef MT_DECODE
  __syncthreads();

  // ---- phase V: one lane per two rows x four columns, result stored as one dword per row ----
  // final rounding per component (default: one list clip((v + 8192 + rnd) >> (14 - bd)), two lists TComYuv::addAvg;
  // explicit weights when the picture carries them)
  const WpTables *wp = WP ? J.wp : nullptr;
  const PredRound Ry = pred_round(wp, t.ref[0], t.ref[1], 0, s.bd_y);
  const PredRound Rcb = pred_round(wp, t.ref[0], t.ref[1], 1, s.bd_c), Rcr = pred_round(wp, t.ref[0], t.ref[1], 2, s.bd_c);
  const int max_y = (1 << s.bd_y) - 1, max_c = (1 << s.bd_c) - 1;
  constexpr int LT = TS * TS / 8, CT = 2 * CS * CS / 8;  // luma and chroma work items
  for (int v0 = 0; v0 < LT + CT; v0 += NTHR) {
    const int v = v0 + tid;
    const bool chroma = (SEG ? hr_uniform(v & ~63) : v) >= LT;
    if (!chroma) {
      const int q = v % (TS / 4), rp = v / (TS / 4);
      int acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.ref[l] < 0) continue;
        const uint4 *src = (const uint4 *)&P_y[l][rp * TS + 4 * q];
        if (vid_y[l]) {
          // rows j+3 and j+4 of the window are the block's rows j and j+1
          const uint4 a = src[(TS / 4) * 1], b = src[(TS / 4) * 2];
          const unsigned qa[4] = {a.x, a.y, a.z, a.w}, qb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int a0 = (int)qa[c] >> 16, a1 = (int)(int16_t)(qb[c] & 0xffff);
            acc[0][c] += WP ? __mul24(Ry.w[l], a0) : a0;
            acc[1][c] += WP ? __mul24(Ry.w[l], a1) : a1;
          }
        } else {
          const unsigned *lt = kLumaTab[t.mvy[l] & 3];
          const unsigned e0 = lt[2], e1 = lt[3], e2 = lt[4], e3 = lt[5], o0 = lt[6], o1 = lt[7], o2 = lt[8], o3 = lt[9], o4 = lt[10];
          unsigned qq[5][4];
#pragma unroll
          for (int i = 0; i < 5; i++) {
            const uint4 a = src[(TS / 4) * i];
            qq[i][0] = a.x, qq[i][1] = a.y, qq[i][2] = a.z, qq[i][3] = a.w;
          }
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int a0 = hr_sdot2(qq[3][c], e3, hr_sdot2(qq[2][c], e2, hr_sdot2(qq[1][c], e1, hr_sdot2z(qq[0][c], e0)))) >> 6;
            const int a1 =
                hr_sdot2(qq[4][c], o4, hr_sdot2(qq[3][c], o3, hr_sdot2(qq[2][c], o2, hr_sdot2(qq[1][c], o1, hr_sdot2z(qq[0][c], o0))))) >> 6;
            acc[0][c] += WP ? __mul24(Ry.w[l], a0) : a0;
            acc[1][c] += WP ? __mul24(Ry.w[l], a1) : a1;
          }
        }
      }
      if (v < LT && rx + 4 * q < s.w) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int y = ry + 2 * rp + j;
          if (y < s.h) {
            int px[4];
#pragma unroll
            for (int c = 0; c < 4; c++) px[c] = tile_px(acc[j][c], Ry.add, Ry.sh, max_y, WP ? Ry.post : 0);
            HR_GLOBAL pel_t *o = (HR_GLOBAL pel_t *)J.rec[0] + (size_t)y * s.stride_y + rx + 4 * q;
            if constexpr (W16)
              *(HR_GLOBAL unsigned long long *)o = (unsigned long long)pack16(px[0], px[1]) | ((unsigned long long)pack16(px[2], px[3]) << 32);
            else
              *(HR_GLOBAL unsigned *)o = pack4(px[0], px[1], px[2], px[3]);
          }
        }
      }
    } else {
      const int vc = v - LT;
      const int comp = vc >= CT / 2 ? 1 : 0, r = vc - comp * (CT / 2);
      const int q = r % (CS / 4), rp = r / (CS / 4);
      int acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      const PredRound &Rc = comp ? Rcr : Rcb;
#pragma unroll
      for (int l = 0; l < 2; l++) {
        if (t.ref[l] < 0 || vc >= CT) continue;
        const uint4 *src = (const uint4 *)&P_c[l][comp * (NPC * CS) + rp * CS + 4 * q];
        if (vid_c[l]) {
          const uint4 a = src[0], b = src[CS / 4];
          const unsigned qa[4] = {a.x, a.y, a.z, a.w}, qb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int a0 = (int)qa[c] >> 16, a1 = (int)(int16_t)(qb[c] & 0xffff);
            acc[0][c] += WP ? __mul24(Rc.w[l], a0) : a0;
            acc[1][c] += WP ? __mul24(Rc.w[l], a1) : a1;
          }
        } else {
          const unsigned *ct = kChromaTab[t.mvy[l] & 7];
          const unsigned e0 = ct[1], e1 = ct[2], o0 = ct[3], o1 = ct[4], o2 = ct[5];
          unsigned qq[3][4];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const uint4 a = src[(CS / 4) * i];
            qq[i][0] = a.x, qq[i][1] = a.y, qq[i][2] = a.z, qq[i][3] = a.w;
          }
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int a0 = hr_sdot2(qq[1][c], e1, hr_sdot2z(qq[0][c], e0)) >> 6;
            const int a1 = hr_sdot2(qq[2][c], o2, hr_sdot2(qq[1][c], o1, hr_sdot2z(qq[0][c], o0))) >> 6;
            acc[0][c] += WP ? __mul24(Rc.w[l], a0) : a0;
            acc[1][c] += WP ? __mul24(Rc.w[l], a1) : a1;
          }
        }
      }
      const int cx = (rx >> 1) + 4 * q;
      if (vc < CT && cx < s.cw) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int y = (ry >> 1) + 2 * rp + j;
          if (y < s.ch) {
            int px[4];
#pragma unroll
            for (int c = 0; c < 4; c++) px[c] = tile_px(acc[j][c], Rc.add, Rc.sh, max_c, WP ? Rc.post : 0);
            HR_GLOBAL pel_t *o = (HR_GLOBAL pel_t *)J.rec[1 + comp] + (size_t)y * s.stride_c + cx;
            if constexpr (W16)
              *(HR_GLOBAL unsigned long long *)o = (unsigned long long)pack16(px[0], px[1]) | ((unsigned long long)pack16(px[2], px[3]) << 32);
            else
              *(HR_GLOBAL unsigned *)o = pack4(px[0], px[1], px[2], px[3]);
          }
        }
      }
    }
  }
}

// ---- k_inter_blk8: 8x8 blocks of one motion inside regions of mixed motion (8-bit pictures) ----------------------
// The two phases of k_inter_tile at the smallest size that still shares work between lanes; a workgroup of four waves
// takes 16 blocks, and the kinds of work are dealt out by wave so that no wave runs two of them one after the other:
//   phase H  waves 0, 1: eight lanes per block, each a pair of the 16 luma window rows (15 are needed; 8 outputs);
//            waves 2, 3: eight lanes per block, each a pair of the 8 window rows of Cb / Cr (7 needed; 4 outputs);
//            16 bytes per row straight from the reference picture;
//   phase V  waves 0, 1: eight lanes per block, two rows x four columns of luma each; wave 2: four lanes per block,
//            two rows x four columns of a chroma block.
// The motion comes from the record of the block's first unit, resolved as in inter_unit (clipMv, identical motion) by
// one lane per block and handed to the others through LDS.  A wave serves several blocks with different motions, so
// nothing is wave-uniform here: the tap tables are read from LDS by fraction, and a zero fraction runs through the
// identity taps (same result, no branch).
struct BlkMotion {
  int x, y, z, w;
};
template <bool WP>
__global__ __launch_bounds__(256) void k_inter_blk8(const PicJob *jobs, SeqInfo s) {
  __shared__ __attribute__((aligned(16))) unsigned P_y[16][2][8 * 8];  // [block][list][row pair * 8 + column]: rows 2p, 2p+1 in one dword
  __shared__ __attribute__((aligned(16))) unsigned P_c[16][2][2 * 4 * 4];  // [block][list][component * 16 + row pair * 4 + column]
  __shared__ __attribute__((aligned(16))) int mot[16][8];  // per block: x, y, mvx[0], mvy[0], mvx[1], mvy[1], ref[0], ref[1] (-1: list not used)
  __shared__ unsigned tab_l[4 * 12], tab_c[8 * 8];
  const PicJob &J = jobs[blockIdx.y];
  if ((int)blockIdx.x * 16 >= J.n_mc_blk8) return;
  const int tid = threadIdx.x;
  const WpTables *wp = WP ? J.wp : nullptr;
  if (tid < 16) {
    const int r = (int)blockIdx.x * 16 + tid;
    int m[8] = {0, 0, 0, 0, 0, 0, -1, -1};
    if (r < J.n_mc_blk8) {
      const McBlk b = J.mc_blk8[r];
      const int x = b.x, y = b.y;
      const Part4 p = J.rec4[(size_t)(y >> 2) * s.uw + (x >> 2)];
      // TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3207-3219): the range depends on the coding unit's origin.
      const int cusize = 64 >> (p.flags & P4_DEPTH_MASK);
      const int cux = x & ~(cusize - 1), cuy = y & ~(cusize - 1);
      const int hmax = (s.w + 8 - cux - 1) * 4, hmin = (-64 - 8 - cux + 1) * 4;
      const int vmax = (s.h + 8 - cuy - 1) * 4, vmin = (-64 - 8 - cuy + 1) * 4;
      bool use0 = p.ref_idx[0] >= 0 && !(p.flags & P4_INTRA), use1 = p.ref_idx[1] >= 0 && !(p.flags & P4_INTRA);
      // TComPrediction::xCheckIdenticalMotion (TLibCommon/TComPrediction.cpp:396-411); not with weighted bi-prediction
      if (use0 && use1 && J.slice_type == 0 && !wp && J.ref_poc[0][p.ref_idx[0]] == J.ref_poc[1][p.ref_idx[1]] && p.mv[0][0] == p.mv[1][0] &&
          p.mv[0][1] == p.mv[1][1])
        use1 = false;
      m[0] = x, m[1] = y;
      m[2] = min(hmax, max(hmin, (int)p.mv[0][0])), m[3] = min(vmax, max(vmin, (int)p.mv[0][1]));
      m[4] = min(hmax, max(hmin, (int)p.mv[1][0])), m[5] = min(vmax, max(vmin, (int)p.mv[1][1]));
      m[6] = use0 ? p.ref_idx[0] : -1, m[7] = use1 ? p.ref_idx[1] : -1;
    }
    *(uint4 *)&mot[tid][0] = make_uint4((unsigned)m[0], (unsigned)m[1], (unsigned)m[2], (unsigned)m[3]);
    *(uint4 *)&mot[tid][4] = make_uint4((unsigned)m[4], (unsigned)m[5], (unsigned)m[6], (unsigned)m[7]);
  }
  if (tid >= 64 && tid < 112) tab_l[tid - 64] = kLumaTab[(tid - 64) / 12][(tid - 64) % 12];
  if (tid >= 128 && tid < 192) tab_c[tid - 128] = kChromaTab[(tid - 128) >> 3][(tid - 128) & 7];
  __syncthreads();

  // ---- phase H ----
  {
    const bool luma = tid < 128;
    const int k = luma ? tid : tid - 128, g = k >> 3;
    const uint4 u0 = *(const uint4 *)&mot[g][0], u1 = *(const uint4 *)&mot[g][4];
    const BlkMotion m0 = {(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w}, m1 = {(int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
    const int x = m0.x, y = m0.y;
#pragma unroll
    for (int l = 0; l < 2; l++) {
      const int ri = l ? m1.w : m1.z, mvx = l ? m1.x : m0.z, mvy = l ? m1.y : m0.w;
      if (ri < 0) continue;
      unsigned w[2][4];
      if (luma) {
        const int rp = k & 7;
        const int wx0 = x + (mvx >> 2) - 3, wy0 = y + (mvy >> 2) - 3;
        const bool fast = wx0 >= 0 && wx0 + 15 < s.w;
        const HR_GLOBAL uint8_t *base = (const HR_GLOBAL uint8_t *)J.ref[l][ri][0];
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
          tile_row_load(base + (size_t)clip3i(0, s.h - 1, wy0 + min(2 * rp + rr, 14)) * s.stride_y, wx0, s.w, fast, w[rr]);
        const unsigned c_lo = tab_l[(mvx & 3) * 12], c_hi = tab_l[(mvx & 3) * 12 + 1];
        int h[2][8];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          unsigned xs[4], a[12];
#pragma unroll
          for (int i = 0; i < 4; i++) xs[i] = w[rr][i] ^ 0x80808080u;
#pragma unroll
          for (int i = 0; i < 12; i++) a[i] = (i & 3) ? hr_alignbyte(xs[(i >> 2) + 1], xs[i >> 2], i & 3) : xs[i >> 2];
#pragma unroll
          for (int i = 0; i < 8; i++) h[rr][i] = hr_sdot4(a[i + 4], c_hi, hr_sdot4z(a[i], c_lo));
        }
        uint4 *dst = (uint4 *)&P_y[g][l][rp * 8];
        dst[0] = make_uint4(pack16(h[0][0], h[1][0]), pack16(h[0][1], h[1][1]), pack16(h[0][2], h[1][2]), pack16(h[0][3], h[1][3]));
        dst[1] = make_uint4(pack16(h[0][4], h[1][4]), pack16(h[0][5], h[1][5]), pack16(h[0][6], h[1][6]), pack16(h[0][7], h[1][7]));
      } else {
        const int comp = (k >> 2) & 1, rp = k & 3;
        const int cwx0 = (x >> 1) + (mvx >> 3) - 1, cwy0 = (y >> 1) + (mvy >> 3) - 1;
        const bool fast = cwx0 >= 0 && cwx0 + 15 < s.cw;
        const HR_GLOBAL uint8_t *base = (const HR_GLOBAL uint8_t *)(comp ? J.ref[l][ri][2] : J.ref[l][ri][1]);
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
          tile_row_load(base + (size_t)clip3i(0, s.ch - 1, cwy0 + min(2 * rp + rr, 6)) * s.stride_c, cwx0, s.cw, fast, w[rr]);
        const unsigned c4 = tab_c[(mvx & 7) * 8];
        int h[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          const unsigned x0 = w[rr][0] ^ 0x80808080u, x1 = w[rr][1] ^ 0x80808080u;
#pragma unroll
          for (int i = 0; i < 4; i++) h[rr][i] = hr_sdot4z(i ? hr_alignbyte(x1, x0, i) : x0, c4);
        }
        *(uint4 *)&P_c[g][l][comp * 16 + rp * 4] = make_uint4(pack16(h[0][0], h[1][0]), pack16(h[0][1], h[1][1]), pack16(h[0][2], h[1][2]), pack16(h[0][3], h[1][3]));
      }
    }
  }
  __syncthreads();
  if (tid >= 192) return;

  // ---- phase V: two rows x four columns per lane, then the uni- or bi-prediction rounding ----
  int acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if (tid < 128) {
    const int g = tid >> 3, q = tid & 1, rp = (tid >> 1) & 3;
    const uint4 u0 = *(const uint4 *)&mot[g][0], u1 = *(const uint4 *)&mot[g][4];
    const BlkMotion m0 = {(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w}, m1 = {(int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
    const int x = m0.x, y = m0.y, r0 = m1.z, r1 = m1.w;
    if (r0 < 0 && r1 < 0) return;  // no block here (the list ended inside the workgroup)
    const PredRound R = pred_round(wp, r0, r1, 0, s.bd_y);
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if ((l ? r1 : r0) < 0) continue;
      const unsigned *lt = &tab_l[((l ? m1.y : m0.w) & 3) * 12];
      const unsigned e0 = lt[2], e1 = lt[3], e2 = lt[4], e3 = lt[5], o0 = lt[6], o1 = lt[7], o2 = lt[8], o3 = lt[9], o4 = lt[10];
      const uint4 *src = (const uint4 *)&P_y[g][l][rp * 8 + 4 * q];
      unsigned qq[5][4];
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const uint4 a = src[2 * i];  // row pair rp + i
        qq[i][0] = a.x, qq[i][1] = a.y, qq[i][2] = a.z, qq[i][3] = a.w;
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int a0 = hr_sdot2(qq[3][c], e3, hr_sdot2(qq[2][c], e2, hr_sdot2(qq[1][c], e1, hr_sdot2z(qq[0][c], e0)))) >> 6;
        const int a1 = hr_sdot2(qq[4][c], o4, hr_sdot2(qq[3][c], o3, hr_sdot2(qq[2][c], o2, hr_sdot2(qq[1][c], o1, hr_sdot2z(qq[0][c], o0))))) >> 6;
        const int wl = l ? R.w[1] : R.w[0];
        acc[0][c] += WP ? __mul24(wl, a0) : a0;
        acc[1][c] += WP ? __mul24(wl, a1) : a1;
      }
    }
    const int maxv = (1 << s.bd_y) - 1;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int px[4];
#pragma unroll
      for (int c = 0; c < 4; c++) px[c] = tile_px(acc[j][c], R.add, R.sh, maxv, WP ? R.post : 0);
      *(HR_GLOBAL unsigned *)((HR_GLOBAL uint8_t *)J.rec[0] + (size_t)(y + 2 * rp + j) * s.stride_y + x + 4 * q) = pack4(px[0], px[1], px[2], px[3]);
    }
  } else {
    const int k = tid - 128, g = k >> 2, comp = (k >> 1) & 1, rp = k & 1;
    const uint4 u0 = *(const uint4 *)&mot[g][0], u1 = *(const uint4 *)&mot[g][4];
    const BlkMotion m0 = {(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w}, m1 = {(int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
    const int x = m0.x, y = m0.y, r0 = m1.z, r1 = m1.w;
    if (r0 < 0 && r1 < 0) return;
    const PredRound R = pred_round(wp, r0, r1, 1 + comp, s.bd_c);
#pragma unroll
    for (int l = 0; l < 2; l++) {
      if ((l ? r1 : r0) < 0) continue;
      const unsigned *ct = &tab_c[((l ? m1.y : m0.w) & 7) * 8];
      const unsigned e0 = ct[1], e1 = ct[2], o0 = ct[3], o1 = ct[4], o2 = ct[5];
      const uint4 *src = (const uint4 *)&P_c[g][l][comp * 16 + rp * 4];
      unsigned qq[3][4];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const uint4 a = src[i];  // row pair rp + i
        qq[i][0] = a.x, qq[i][1] = a.y, qq[i][2] = a.z, qq[i][3] = a.w;
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int a0 = hr_sdot2(qq[1][c], e1, hr_sdot2z(qq[0][c], e0)) >> 6;
        const int a1 = hr_sdot2(qq[2][c], o2, hr_sdot2(qq[1][c], o1, hr_sdot2z(qq[0][c], o0))) >> 6;
        const int wl = l ? R.w[1] : R.w[0];
        acc[0][c] += WP ? __mul24(wl, a0) : a0;
        acc[1][c] += WP ? __mul24(wl, a1) : a1;
      }
    }
    const int maxv = (1 << s.bd_c) - 1;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int px[4];
#pragma unroll
      for (int c = 0; c < 4; c++) px[c] = tile_px(acc[j][c], R.add, R.sh, maxv, WP ? R.post : 0);
      *(HR_GLOBAL unsigned *)((HR_GLOBAL uint8_t *)J.rec[1 + comp] + (size_t)((y >> 1) + 2 * rp + j) * s.stride_c + (x >> 1)) = pack4(px[0], px[1], px[2], px[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_transform: de-quantisation + inverse transform
// ------------------------------------------------------------------------------------------------
// N lanes per transform block, 256/N blocks per workgroup.  Pass 1: lane c owns column c of the
// coefficients and produces column c of the intermediate (N sums over the rows); pass 2: lane y owns
// row y of the intermediate and produces row y of the residual.  This is the plain matrix form of the
// reference's partial butterflies (TLibCommon/TComTrQuant.cpp:464-863: same sums, shifts 7 and
// 12-(bd-8), 16-bit clip after each pass); the matrix entry is wave-uniform in both passes, and a
// coefficient row that is zero in the whole wave is skipped.
// De-quantisation: TComTrQuant::xDeQuant flat path (TLibCommon/TComTrQuant.cpp:1174-1185) with the QP of
// TComTrQuant::setQPforQuant (:192-214); transform skip: xITransformSkip (:1454-1486).
// `group`: which 256 / N transform blocks of the picture's list of this size the workgroup takes; `lds`: TPB * N * (N + 1) words.
template <typename pel_t, int LOG2N>
__device__ __forceinline__ void transform_blocks(const PicJob &J, const SeqInfo &s, int group, int *lds) {
  constexpr int N = 1 << LOG2N;
  constexpr int TPB = 256 / N;  // transform blocks per workgroup
  int(*tmp)[N][N + 1] = (int(*)[N][N + 1])lds;
  const int t_local = threadIdx.x / N, c = threadIdx.x % N;
  const int t_idx = group * TPB + t_local;
  const bool live = t_idx < J.n_tu[LOG2N - 2];
  TuItem it;
  it.x = it.y = 0;
  it.comp = it.flags = 0;
  it.qp = 0;
  it.coeff_off = 0;
  if (live) it = J.tu[LOG2N - 2][t_idx];
  const int comp = it.comp;
  const int bd = comp ? s.bd_c : s.bd_y;
  // QP
  int q;
  if (comp == 0) {
    q = it.qp + 6 * (s.bd_y - 8);
  } else {
    const int bdo = 6 * (s.bd_c - 8);
    int qc = clip3i(-bdo, 57, it.qp + (comp == 1 ? J.cb_off : J.cr_off));
    q = qc < 0 ? qc + bdo : kChromaScale[qc] + bdo;
  }
  const int per = q / 6, rem = q - per * 6;
  const int iq = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  const int scale = iq << per;
  const int shift = bd + LOG2N - 9, add = 1 << (shift - 1);
  const uint32_t *coef = J.coeff + it.coeff_off;  // header word, then the block's non-zero coefficients
  const bool tskip = (it.flags & TU_TSKIP) != 0;
  // lossless coding unit (TComTrQuant::invtransformNxN, TLibCommon/TComTrQuant.cpp:1269-1279): residual = coefficients
  // IPCM coding unit (TDecCu::xDecodePCMTexture, TLibDecoder/TDecCu.cpp:798-837): the block holds samples, not coefficients
  const bool pcm = (it.flags & TU_PCM) != 0;
  const bool bypass = (it.flags & (TU_BYPASS | TU_PCM)) != 0;
  const bool dst4 = LOG2N == 2 && (it.flags & TU_DST);

  // ---- pass 1 (columns) ----
  int acc[N];
#pragma unroll
  for (int y = 0; y < N; y++) acc[y] = 0;
  // The block's non-zero coefficients are scattered into a cleared LDS tile (the N lanes of the block share the
  // list), then every lane picks up its column.  Afterwards lane c only ever writes column c again before the
  // barrier ahead of pass 2, so the tile can be reused for the intermediate without another barrier.
#pragma unroll
  for (int k = 0; k < N; k++) tmp[t_local][k][c] = 0;
  const int nnz = live ? (int)coef[0] : 0;
  __syncthreads();
  for (int e = c; e < nnz; e += N) {
    const unsigned w = coef[1 + e];
    const int pos = (int)(w & 0xffffu);
    tmp[t_local][pos >> LOG2N][pos & (N - 1)] = (int)(int16_t)(w >> 16);
  }
  __syncthreads();
  int dq[N];
#pragma unroll
  for (int k = 0; k < N; k++) dq[k] = tmp[t_local][k][c];
  // Scaling lists (TComTrQuant::xDeQuant, scaling-list branch, TLibCommon/TComTrQuant.cpp:1142-1173): the lane's
  // column meets at most 8 different matrix entries (one per matrix row; a 16x16 / 32x32 entry spans 2 / 4 rows)
  const ScalingTables *sl = J.scaling;
  int mrow[8] = {16, 16, 16, 16, 16, 16, 16, 16};
  int mdc = 16;
  if (sl && live) {
    const int type = ((it.flags & TU_TO_RESI) ? 0 : 3) + comp;  // intra blocks are the ones that keep their residual
    const uint8_t *m = LOG2N == 2 ? sl->m4[type] : LOG2N == 3 ? sl->m8[type] : LOG2N == 4 ? sl->m16[type] : sl->m32[type ? 1 : 0];
    constexpr int MW = N < 8 ? N : 8, RATIO = N / MW;
#pragma unroll
    for (int r = 0; r < MW; r++) mrow[r] = m[r * MW + c / RATIO];
    mdc = LOG2N == 4 ? sl->dc16[type] : LOG2N == 5 ? sl->dc32[type ? 1 : 0] : mrow[0];
  }
  const int sl_shift = shift + 4 - per;  // > 0: round and shift right; <= 0: shift left
  const int sl_lim = 1 << min(15, 12 + LOG2N + bd - per);
  // Rows are taken two at a time: both factors fit 16 bits (the de-quantised value is clipped to 16 bits, the
  // matrix entries to 8), so one 2-way 16-bit dot product does two multiply-adds exactly.
#pragma unroll
  for (int k2 = 0; k2 < N; k2 += 2) {
    int dd[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int k = k2 + h;
      int d;
      if (!sl) {
        d = clip3i(-32768, 32767, (dq[k] * scale + add) >> shift);
      } else {
        constexpr int RATIO = N / (N < 8 ? N : 8);
        const int mk = (RATIO > 1 && k == 0 && c == 0) ? mdc : mrow[k / RATIO];
        const int coef = iq * mk;
        if (sl_shift > 0)
          d = clip3i(-32768, 32767, (dq[k] * coef + (1 << (sl_shift - 1))) >> sl_shift);
        else
          d = clip3i(-32768, 32767, (clip3i(-sl_lim, sl_lim - 1, dq[k]) * coef) * (1 << (-sl_shift)));
      }
      if (tskip) tmp[t_local][k][c] = d;  // transform-skip blocks bypass both passes: keep the de-quantised block as it is
      if (bypass) tmp[t_local][k][c] = dq[k];
      dd[h] = (tskip || bypass) ? 0 : d;
    }
    if (!__any((dd[0] | dd[1]) != 0)) continue;  // wave-uniform
    const unsigned pk = pack16(dd[0], dd[1]);
#pragma unroll
    for (int y = 0; y < N; y++) {
      const unsigned m = kTP[LOG2N - 2][k2 / 2][y];
      const unsigned md = (LOG2N == 2) ? kDst4P[k2 / 2][y & 3] : 0u;
      acc[y] = hr_sdot2(pk, dst4 ? md : m, acc[y]);
    }
  }
  if (!tskip && !bypass) {
#pragma unroll
    for (int y = 0; y < N; y++) tmp[t_local][y][c] = clip3i(-32768, 32767, (acc[y] + 64) >> 7);
  }
  __syncthreads();
  // ---- pass 2 (rows): lane c now owns row y = c ----
  const int y = c;
  int res[N];
#pragma unroll
  for (int x = 0; x < N; x++) res[x] = 0;
  int row[N];  // the lane's row of the intermediate, read from LDS in one go
#pragma unroll
  for (int k = 0; k < N; k++) row[k] = tmp[t_local][y][k];
#pragma unroll
  for (int k2 = 0; k2 < N; k2 += 2) {
    const int v0 = (tskip || bypass) ? 0 : row[k2], v1 = (tskip || bypass) ? 0 : row[k2 + 1];
    if (!__any((v0 | v1) != 0)) continue;  // wave-uniform
    const unsigned pk = pack16(v0, v1);
#pragma unroll
    for (int x = 0; x < N; x++) {
      const unsigned m = kTP[LOG2N - 2][k2 / 2][x];
      const unsigned md = (LOG2N == 2) ? kDst4P[k2 / 2][x & 3] : 0u;
      res[x] = hr_sdot2(pk, dst4 ? md : m, res[x]);
    }
  }
  if (bypass) {
    const int up = pcm ? bd - (comp ? s.pcm_bd_c : s.pcm_bd_y) : 0;
#pragma unroll
    for (int x = 0; x < N; x++) res[x] = row[x] * (1 << up);
  } else if (tskip) {
    const int ts = 15 - bd - LOG2N;  // > 0 for every supported bit depth
#pragma unroll
    for (int x = 0; x < N; x++) res[x] = (row[x] + (1 << (ts - 1))) >> ts;
  } else {
    const int s2 = 12 - (bd - 8), r2 = 1 << (s2 - 1);
#pragma unroll
    for (int x = 0; x < N; x++) res[x] = clip3i(-32768, 32767, (res[x] + r2) >> s2);
  }
  if (!live) return;
  if (it.flags & TU_TO_RESI) {
    int16_t *o = J.resi[comp] + (size_t)(it.y + y) * (comp ? s.stride_c : s.stride_y) + it.x;
#pragma unroll
    for (int x = 0; x < N; x++) o[x] = (int16_t)res[x];
  } else if (pcm) {
    pel_t *o = (pel_t *)J.rec[comp] + (size_t)(it.y + y) * (comp ? s.stride_c : s.stride_y) + it.x;
#pragma unroll
    for (int x = 0; x < N; x++) o[x] = (pel_t)res[x];
  } else {
    // TDecCu::xReconInter / TComYuv::addClip (TLibDecoder/TDecCu.cpp:468-486, TLibCommon/TComYuv.cpp:401-459):
    // adding a zero residual is the identity, so only coded blocks need touching.
    const int stride = comp ? s.stride_c : s.stride_y, maxv = (1 << bd) - 1;
    pel_t *o = (pel_t *)J.rec[comp] + (size_t)(it.y + y) * stride + it.x;
#pragma unroll
    for (int x = 0; x < N; x++) o[x] = (pel_t)clip3i(0, maxv, (int)o[x] + res[x]);
  }
}

// ---- wave-level helpers of the intra kernel: neighbouring lanes' values and a sum over the wave, without LDS traffic
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ int hr_lane_prev(int v) { return __shfl(v, (emu::lane() + 63) & 63); }  // lane 0: unspecified
__device__ __forceinline__ int hr_lane_next(int v) { return __shfl(v, (emu::lane() + 1) & 63); }   // lane 63: unspecified
__device__ __forceinline__ int hr_wave_sum(int v) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#else
__device__ __forceinline__ int hr_lane_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int hr_lane_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int hr_wave_sum(int v) {
  // running sums inside each row of 16 lanes (row_shr:1,2,4,8, zeros shifted in), then the four row totals
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  return hr_readlane(v, 15) + hr_readlane(v, 31) + hr_readlane(v, 47) + hr_readlane(v, 63);
}
#endif

#define IT_WAVES HR_INTRA_ROUND_BLOCKS /* waves per workgroup (power of two) = blocks per round */
#define IT_NTHR (IT_WAVES * 64)
#define IT_X0 4   /* tile column of the CTU's first sample: the column left of it holds the left neighbours, and
                     groups of four samples start on 8-byte boundaries */
#define IT_YW 100 /* luma tile row: x = -1 .. 95 in columns 3 .. 99 */
#define IT_CW 52  /* chroma tile row: x = -1 .. 47 in columns 3 .. 51 */
#define IT_REFW 136 /* reference samples of a block: 4 * 32 + 1, rounded up */

// invAngle by |mode - 26| (or |mode - 10|), TLibCommon/TComPrediction.cpp:226-230
__constant__ int kIntraInvAngle[16] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256, 0, 0, 0, 0, 0, 0, 0};
// (the kernel keeps the table in one register, entry i in lane i, and picks from it with a lane read)

// One wave builds the reference samples r[0 .. 4n] of a block (r[2n] is the corner, r[2n-1-j] the left column going
// down, r[2n+1+i] the row above going right: TComPattern::initAdiPattern / fillReferenceSamples,
// TLibCommon/TComPattern.cpp:213-602): gathered from the CTU tile with unavailable samples substituted, smoothed when
// the block's mode asks for it (only the version the block predicts from is kept), stored to ref[].  Lane l holds
// samples l, l + 64 and l + 128.  Everything but `lane` is the same in all lanes.  Returns the DC value (mode 1).
// base_off: tile position of the corner sample.
__device__ __forceinline__ int intra_references(const uint16_t *tile, int tw, int base_off, int log2n, int comp, int mode, int flags, unsigned w2,
                                                unsigned w3, int bd, int strong_intra, uint16_t *ref, int lane) {
  const int n = 1 << log2n, len = 4 * n + 1;
  const int kind = (w3 >> IP_KIND_SHIFT) & 3;
  const int first_i = (w3 >> IP_FIRST_SHIFT) & 255, last_i = (w3 >> IP_LAST_SHIFT) & 255;
  int v[3];
#pragma unroll
  for (int k = 0; k < 3; k++) v[k] = 1 << (bd - 1);
  if (kind != IP_KIND_NONE) {
    int src[3];
    if (kind == IP_KIND_RUN) {
      // the available samples form one run: what lies before it takes its first sample, what lies after it its last
#pragma unroll
      for (int k = 0; k < 3; k++) src[k] = min(max(lane + 64 * k, first_i), last_i);
    } else {
      // holes (slice and tile corners, constrained intra prediction): per sample from the unit masks.  Availability
      // of the 4*nu+1 units in sample order: units 0..31 in m, the 33rd unit of a 32x32 luma block in m_top
      const int lus = comp ? 1 : 2, us = 1 << lus, nu = n >> lus;
      const unsigned long long m64 = (unsigned long long)(hr_brev(w2 & 0xffff) >> (32 - 2 * nu)) |
                                     ((flags & IN_CORNER) ? 1ull << (2 * nu) : 0ull) |
                                     ((unsigned long long)((w2 >> 16) & ((1u << (2 * nu)) - 1)) << (2 * nu + 1));
      const unsigned m = (unsigned)m64;
      const bool m_top = (m64 >> 32) != 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int i = min(lane + 64 * k, len - 1);
        const int u = i < 2 * n ? i >> lus : (i == 2 * n ? 2 * nu : 2 * nu + 1 + ((i - 2 * n - 1) >> lus));
        int sk = i;
        const bool avail = u < 32 ? ((m >> u) & 1) != 0 : m_top;
        if (!avail) {
          const unsigned below = u < 32 ? m & ((1u << u) - 1) : m;
          if (below) {
            const int ub = 31 - __clz((int)below);  // nearest available unit before this one: its last sample
            sk = ub < 2 * nu ? (ub << lus) + us - 1 : (ub == 2 * nu ? 2 * n : 2 * n + 1 + ((ub - 2 * nu - 1) << lus) + us - 1);
          } else {
            sk = first_i;
          }
        }
        src[k] = sk;
      }
    }
    // sample number -> tile position: the left column runs upwards to the corner, the row above runs right from it;
    // all reads are issued before the first is used
    int off[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int d = src[k] - 2 * n;
      off[k] = base_off + max(d, __mul24(d, -tw));
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (64 * k < len) v[k] = tile[off[k]];
  }
  int out[3] = {v[0], v[1], v[2]};
  int dcv = 0;
  if (w3 & IP_FILTER) {
    // [1 2 1] smoothing, or the bilinear one of flat 32x32 blocks; the end samples stay (TComPattern.cpp:263-356)
    int r0 = 0, r2n = 0, r4n = 0;
    bool strong = false;
    if (log2n == 5 && strong_intra) {
      r0 = hr_readlane(v[0], 0);
      r2n = hr_readlane(v[1], 0);
      r4n = hr_readlane(v[2], 0);
      const int rn = hr_readlane(v[0], 32), r3n = hr_readlane(v[1], 32), thr = 1 << (bd - 5);
      strong = iabs(r0 + r2n - 2 * rn) < thr && iabs(r2n + r4n - 2 * r3n) < thr;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (64 * k >= len) continue;
      const int i = lane + 64 * k;
      // neighbours across the 64-sample slots come from the other slot's end lane
      int p = hr_lane_prev(v[k]), nx = hr_lane_next(v[k]);
      if (k > 0) {
        const int edge_p = hr_readlane(v[k - 1], 63);
        p = lane == 0 ? edge_p : p;
      }
      if (k < 2 && 64 * (k + 1) < len) {
        const int edge_n = hr_readlane(v[k + 1], 0);
        nx = lane == 63 ? edge_n : nx;
      }
      int f = (p + 2 * v[k] + nx + 2) >> 2;
      if (strong) f = k == 0 ? ((64 - lane) * r0 + lane * r2n + 32) >> 6 : ((64 - lane) * r2n + lane * r4n + 32) >> 6;
      out[k] = (i == 0 || i >= len - 1) ? v[k] : f;
    }
  } else if (mode == 1) {
    // DC value: mean of the n samples above and the n samples to the left (TLibCommon/TComPrediction.cpp:122-160)
    int part = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = lane + 64 * k;
      if (i >= n && i <= 3 * n && i != 2 * n) part += v[k];
    }
    dcv = (hr_wave_sum(part) + n) >> (log2n + 1);
  }
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (lane + 64 * k < len) ref[lane + 64 * k] = (uint16_t)out[k];
  return dcv;
}

// A block as the prediction sees it; the same in every lane.
struct IntraBlk {
  int log2n, mode, angle, inv, tw, rw, maxv, dcv;
  bool edge, has_resi;
  uint16_t *tile0;        // the block's first sample in the CTU tile
  const int16_t *resi0;   // its residual
  const uint16_t *cn;     // its reference samples: cn[0] corner, cn[-1-j] left[j], cn[1+i] above[i]
};
__device__ __forceinline__ IntraBlk intra_blk(const SeqInfo &s, uint16_t *tile_y, uint16_t *tile_cb, uint16_t *tile_cr, const int16_t *resi_y,
                                              const int16_t *resi_cb, const int16_t *resi_cr, int ctux, int ctuy, unsigned w0, unsigned w1,
                                              unsigned w3, int dcv, const uint16_t *refs, int inv_tab) {
  IntraBlk B;
  B.log2n = w1 & 255;
  const int n = 1 << B.log2n;
  const int x0 = w0 & 0xffff, y0 = w0 >> 16;
  const int comp = (w1 >> 8) & 255, flags = w1 >> 24;
  B.mode = (w1 >> 16) & 255;
  const int sh = comp ? 1 : 0;
  B.maxv = (1 << (comp ? s.bd_c : s.bd_y)) - 1;
  B.tw = comp ? IT_CW : IT_YW;
  B.rw = comp ? 32 : 64;
  const int tx0 = x0 - (ctux >> sh) + IT_X0, ty0 = y0 - (ctuy >> sh) + 1;
  B.tile0 = (comp == 0 ? tile_y : comp == 1 ? tile_cb : tile_cr) + ty0 * B.tw + tx0;
  B.resi0 = (comp == 0 ? resi_y : comp == 1 ? resi_cb : resi_cr) + (ty0 - 1) * B.rw + tx0 - IT_X0;
  B.cn = refs + 2 * n;
  B.edge = comp == 0 && n <= 16;
  B.has_resi = (flags & IN_HAS_RESI) != 0;
  B.angle = (int)((w3 >> IP_ANGLE_SHIFT) & 127) - 32;
  B.inv = hr_readlane(inv_tab, (int)((w3 >> IP_ABS_SHIFT) & 15));
  B.dcv = dcv;
  return B;
}

// Prediction + residual + clip of one intra block, written into the CTU tile (TLibCommon/TComPrediction.cpp:183-390,
// 691-763).  The block is cut into groups of G samples that lie side by side ACROSS the prediction direction (in
// a row for planar, DC and the vertical modes; in a column for the horizontal modes, which are the vertical ones on
// the transposed block: sgn flips the index direction of the reference array).  The samples of a group share
// their position along the direction -- so the angular modes' reference offset and fraction -- and read G + 1
// neighbouring reference samples; row groups of four load their residual and store their result as 8 bytes.
// G = 4 for blocks of 16x16 and more (a 16x16 block is one pass of a wave), G = 1 for the small ones, where one
// sample per lane is the shorter chain.  This call handles groups g_first, g_first + g_step, ... below ngr.
template <int G>
__device__ __forceinline__ void intra_predict(const IntraBlk &B, int g_first, int g_step, int ngr) {
  constexpr int LG = G == 4 ? 2 : 0;
  static_assert(G == 4 || G == 1, "groups of four samples or single samples");
  const int log2n = B.log2n, n = 1 << log2n;
  const int mode = B.mode, maxv = B.maxv, tw = B.tw, rw = B.rw, dcv = B.dcv;
  const uint16_t *cn = B.cn;
  const bool hor = mode >= 2 && mode < 18;
  const int sgn = hor ? -1 : 1;  // MAIN(i) = cn[sgn * i], SIDE(i) = cn[-sgn * i]
  const int angle = B.angle, inv = B.inv;
  for (int g = g_first; g < ngr; g += g_step) {
    // kk: position along the prediction direction, ee: first of the G positions across it.  Neighbouring lanes take
    // neighbouring columns either way (kk for the horizontal modes, ee for the others), so that a wave's tile and
    // residual accesses run along rows.
    const int kk = hor ? g & (n - 1) : g >> (log2n - LG);
    const int ee = hor ? (g >> log2n) << LG : (g & ((n >> LG) - 1)) << LG;
    int v[G];
    if (mode == 0) {  // planar: kk = row, ee = column
      const int left = cn[-1 - kk], tr = cn[1 + n], bl = cn[-1 - n];
#pragma unroll
      for (int e = 0; e < G; e++) {
        const int top = cn[1 + ee + e];
        v[e] = ((left << log2n) + n + __mul24(ee + e + 1, tr - left) + (top << log2n) + __mul24(kk + 1, bl - top)) >> (log2n + 1);
      }
    } else if (mode == 1) {  // DC, with its edge smoothing for luma blocks up to 16x16
#pragma unroll
      for (int e = 0; e < G; e++) v[e] = dcv;
      if (B.edge) {
        if (kk == 0) {
#pragma unroll
          for (int e = 0; e < G; e++) v[e] = (cn[1 + ee + e] + 3 * dcv + 2) >> 2;
          if (ee == 0) v[0] = (cn[1] + cn[-1] + 2 * dcv + 2) >> 2;
        } else if (ee == 0) {
          v[0] = (cn[-1 - kk] + 3 * dcv + 2) >> 2;
        }
      }
    } else if (angle == 0) {  // pure vertical / horizontal
#pragma unroll
      for (int e = 0; e < G; e++) v[e] = cn[sgn * (ee + e + 1)];
      if (B.edge && ee == 0) v[0] = clip3i(0, maxv, cn[sgn] + ((cn[-sgn * (kk + 1)] - cn[0]) >> 1));
    } else {
      const int t = __mul24(kk + 1, angle), i0 = ee + 1 + (t >> 5), f = t & 31;
      int m[G + 1];
      if (angle > 0) {
#pragma unroll
        for (int j = 0; j < G + 1; j++) m[j] = cn[sgn * (i0 + j)];
      } else {
        // negative indices address the side reference projected onto the main one
#pragma unroll
        for (int j = 0; j < G + 1; j++) {
          const int idx = i0 + j;
          m[j] = cn[idx >= 0 ? sgn * idx : -sgn * ((128 - __mul24(idx, inv)) >> 8)];
        }
      }
      // (f = 0 gives m[e] exactly, so the interpolation needs no special case)
#pragma unroll
      for (int e = 0; e < G; e++) v[e] = (__mul24(32 - f, m[e]) + __mul24(f, m[e + 1]) + 16) >> 5;
    }
    if constexpr (G == 4) {
      if (!hor) {
        const int at = __mul24(kk, tw) + ee;
        if (B.has_resi) {
          const uint2 r = *(const uint2 *)(B.resi0 + __mul24(kk, rw) + ee);
          v[0] += (int)(int16_t)(r.x & 0xffffu), v[1] += (int)r.x >> 16, v[2] += (int)(int16_t)(r.y & 0xffffu), v[3] += (int)r.y >> 16;
        }
#pragma unroll
        for (int e = 0; e < G; e++) v[e] = clip3i(0, maxv, v[e]);
        *(uint2 *)(B.tile0 + at) = make_uint2((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16));
        continue;
      }
    }
    if (!hor) {
      // (single samples) kk = row, ee = column
      int x = v[0];
      if (B.has_resi) x += B.resi0[__mul24(kk, rw) + ee];
      B.tile0[__mul24(kk, tw) + ee] = (uint16_t)clip3i(0, maxv, x);
    } else {
#pragma unroll
      for (int e = 0; e < G; e++) {
        int x = v[e];
        if (B.has_resi) x += B.resi0[__mul24(ee + e, rw) + kk];
        B.tile0[__mul24(ee + e, tw) + kk] = (uint16_t)clip3i(0, maxv, x);
      }
    }
  }
}

// The 32x32 blocks have a launch of their own (33 KB of LDS and 103 registers per lane would cost the small blocks
// their occupancy); the 16x16, 8x8 and 4x4 blocks of a picture share one: its workgroups take the 16x16 groups first.
template <typename pel_t, int LOG2N>
__global__ __launch_bounds__(256) void k_transform(const PicJob *jobs, SeqInfo s) {
  constexpr int N = 1 << LOG2N;
  __shared__ int lds[(256 / N) * N * (N + 1)];
  transform_blocks<pel_t, LOG2N>(jobs[blockIdx.y], s, blockIdx.x, lds);
}

template <typename pel_t>
__global__ __launch_bounds__(256) void k_transform_small(const PicJob *jobs, SeqInfo s) {
  __shared__ int lds[16 * 16 * 17];  // the largest of 16 x (16 x 17), 32 x (8 x 9), 64 x (4 x 5)
  const PicJob &J = jobs[blockIdx.y];
  const int g16 = (J.n_tu[2] + 15) >> 4, g8 = (J.n_tu[1] + 31) >> 5, g4 = (J.n_tu[0] + 63) >> 6;
  const int b = blockIdx.x;
  if (b < g16)
    transform_blocks<pel_t, 4>(J, s, b, lds);
  else if (b < g16 + g8)
    transform_blocks<pel_t, 3>(J, s, b - g16, lds);
  else if (b < g16 + g8 + g4)
    transform_blocks<pel_t, 2>(J, s, b - g16 - g8, lds);
}

// ------------------------------------------------------------------------------------------------
// k_intra: intra prediction + residual add.  One launch per batch; a workgroup works on one CTU at a time.
// ------------------------------------------------------------------------------------------------
// The CTU's samples live in LDS together with the row above it (which reaches 32 samples into the
// above-right CTU, as far as any block's above-right neighbours go) and the column left of it.  What a CTU reads
// from outside was finished by earlier launches (inter prediction) or by neighbouring CTUs of this launch, whose
// "done" words the workgroup waits for.  Inside the CTU the blocks are processed level by level (blocks of a level
// do not depend on each other), IT_WAVES blocks per round: one wave per block gathers, substitutes and smooths the
// block's reference samples and predicts the block; the passes of a 32x32 block are shared out among the waves.
//
// Reference-sample array r[0..4n]: r[2n] is the corner, r[2n-1-j] the left column going down,
// r[2n+1+i] the row above going right (TComPattern::initAdiPattern / fillReferenceSamples,
// TLibCommon/TComPattern.cpp:213-602).  Availability comes per 4-luma-sample unit from the packer.
#define IT_MAX_ITEMS HR_INTRA_MAX_ITEMS
#define IT_MAX_ROUNDS HR_INTRA_MAX_ROUNDS

// Diagnostic build only (-DHEVCRECON_INTRA_STAMPS): cycles that lane 0 of every workgroup spends in each phase of a
// CTU, summed over the launch; read back and printed by hr_intra_stamps_report().
#ifdef HEVCRECON_INTRA_STAMPS
__device__ unsigned long long hr_intra_stamps[16];
#define HR_STAMP(k)                                                                   \
  if (tid == 0) {                                                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                     \
    acc_[k] += now_ - stamp_;                                                         \
    stamp_ = now_;                                                                    \
  }
void hr_intra_stamps_report() {
  unsigned long long v[16];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(hr_intra_stamps), sizeof v) != hipSuccess) return;
  const char *names[16] = {"ticket", "wait for neighbours", "staging", "rounds (rest)", "write-back issue", "drain + barrier", "CTUs", "rounds",
                           "round: set-up", "round: references + prediction", "", "", "round: barrier", "", "", ""};
  for (int k = 0; k < 14; k++) fprintf(stderr, "k_intra %-20s %llu\n", names[k], v[k]);
}
#else
#define HR_STAMP(k)
#endif

// INSIDE: the batch holds pictures whose CTUs carry samples before this kernel runs (inter prediction in P and B
// pictures, IPCM units): those CTUs are staged whole -- their samples are requested before the wait for the
// neighbours, like the residual -- where a batch of plain intra pictures needs the borders only.  Two
// instantiations, so that the second keeps the registers free for three workgroups per compute unit.
template <typename pel_t, bool INSIDE>
__global__ __launch_bounds__(IT_NTHR, INSIDE ? 4 : 6) void k_intra(const PicJob *jobs, SeqInfo s, const unsigned *list, int n_items, unsigned *ticket) {
  __shared__ __attribute__((aligned(16))) uint16_t tile_y[65 * IT_YW];
  __shared__ __attribute__((aligned(16))) uint16_t tile_c[2][33 * IT_CW];
  __shared__ __attribute__((aligned(16))) int16_t resi_y[64 * 64];
  __shared__ __attribute__((aligned(16))) int16_t resi_c[2][32 * 32];
  __shared__ __attribute__((aligned(16))) IntraItem items_s[IT_MAX_ITEMS];
  __shared__ uint16_t rounds_s[IT_MAX_ROUNDS];
  __shared__ uint16_t ref_s[IT_WAVES][IT_REFW];
  __shared__ unsigned ticket_s;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int inv_tab = kIntraInvAngle[lane & 15];
  // One launch per batch.  The batch's CTUs (those holding intra blocks) stand in a list ordered so that every CTU
  // comes after the CTUs it reads from (all pictures' first waves, then their second waves, ...; for a single
  // picture the order of its own table, list = NULL).  Workgroups draw the next entry until the list is exhausted
  // and wait, per CTU, for the neighbouring CTUs it depends on.  Whatever the number of workgroups resident at a
  // time, the earliest unfinished entry never waits for an entry that has not been drawn, so the launch drains.
#ifdef HEVCRECON_INTRA_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
  unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // summed per workgroup, added to the totals once at the end
#endif
  for (;;) {
  __syncthreads();  // the previous CTU's tile has been written back by every wave
  if (tid == 0) ticket_s = hr_ticket_take(ticket);
  __syncthreads();
  const unsigned entry_idx = hr_uniform(ticket_s);  // the same in every lane: what follows from it stays in scalar registers
  if (entry_idx >= (unsigned)n_items) break;
  HR_STAMP(0)
  const unsigned entry = list ? list[entry_idx] : entry_idx;
  const PicJob &J = jobs[entry >> 16];
  const IntraCtu ci = J.ictu[entry & 0xffffu];
  const int ctux = ci.cx * 64, ctuy = ci.cy * 64;
  // What the CTU needs that no neighbouring CTU produces -- its residual, its block list -- is requested now and
  // lands in registers while lane 0 waits for the neighbours: one item of each kind per lane.
  const int nit = ci.n_items < IT_MAX_ITEMS ? ci.n_items : IT_MAX_ITEMS;
  uint4 pre_ry = make_uint4(0, 0, 0, 0), pre_item = make_uint4(0, 0, 0, 0);
  uint2 pre_rc = make_uint2(0, 0);
  // (IPCM units are why an intra picture may need its CTUs staged whole: the transform kernel has already stored them)
  const bool need_inside = INSIDE && (J.slice_type != 2 || J.has_pcm);
  constexpr int SPV = 16 / sizeof(pel_t);                // samples per 16-byte vector
  constexpr int VPR_Y = 64 / SPV, VPR_C = 32 / SPV;      // vectors per luma / chroma row of the CTU
  constexpr int NV_Y = (64 * VPR_Y + IT_NTHR - 1) / IT_NTHR, NV_C = (2 * 32 * VPR_C + IT_NTHR - 1) / IT_NTHR;
  uint4 pre_in_y[INSIDE ? NV_Y : 1], pre_in_c[INSIDE ? NV_C : 1];
  if constexpr (INSIDE) {
    if (need_inside) {
      const pel_t *py = (const pel_t *)J.rec[0];
#pragma unroll
      for (int k = 0; k < NV_Y; k++) {
        const int v = tid + k * IT_NTHR, ty = v / VPR_Y, tx = (v % VPR_Y) * SPV;
        pre_in_y[k] = make_uint4(0, 0, 0, 0);
        if (v < 64 * VPR_Y && ctux + tx < s.w && ctuy + ty < s.h) pre_in_y[k] = *(const uint4 *)(py + (size_t)(ctuy + ty) * s.stride_y + ctux + tx);
      }
#pragma unroll
      for (int k = 0; k < NV_C; k++) {
        const int v = tid + k * IT_NTHR;
        const int c = v / (32 * VPR_C), r = v % (32 * VPR_C), ty = r / VPR_C, tx = (r % VPR_C) * SPV;
        pre_in_c[k] = make_uint4(0, 0, 0, 0);
        if (v < 2 * 32 * VPR_C && (ctux >> 1) + tx < s.cw && (ctuy >> 1) + ty < s.ch)
          pre_in_c[k] = *(const uint4 *)((const pel_t *)J.rec[1 + c] + (size_t)((ctuy >> 1) + ty) * s.stride_c + (ctux >> 1) + tx);
      }
    }
  }
  unsigned pre_lvl = 0;
  static_assert(IT_NTHR == 64 * 64 / 8 && IT_NTHR == 2 * 32 * 32 / 4 && IT_MAX_ITEMS <= IT_NTHR && IT_MAX_ROUNDS <= IT_NTHR,
                "one residual vector, one block and one round per lane");
  {
    const int ty = tid >> 3, tx = (tid & 7) * 8;
    if (ctux + tx < s.w && ctuy + ty < s.h) pre_ry = *(const uint4 *)&J.resi[0][(size_t)(ctuy + ty) * s.stride_y + ctux + tx];
    const int c = tid >> 8, i = tid & 255, cty = i >> 3, ctx = (i & 7) * 4;
    if ((ctux >> 1) + ctx < s.cw && (ctuy >> 1) + cty < s.ch)
      pre_rc = *(const uint2 *)&J.resi[1 + c][(size_t)((ctuy >> 1) + cty) * s.stride_c + (ctux >> 1) + ctx];
    if (tid < nit) pre_item = *(const uint4 *)&J.intra[ci.item_start + tid];
    if (tid < ci.n_rounds && tid < IT_MAX_ROUNDS) pre_lvl = J.rounds[ci.round_start + tid];
  }
  if (tid == 0 && (ci.n_items > IT_MAX_ITEMS || ci.n_rounds > IT_MAX_ROUNDS)) atomicOr(J.error_flag, HR_ERR_INTRA_CAPACITY);
  if (wv == 0 && ci.deps) {
    // the neighbouring CTUs whose intra blocks this one reads: left, above-left, above, above-right; lane d of the
    // first wave watches neighbour d, so that the four flags are read side by side
    const int d = lane & 3;
    const int at = d == 0 ? -1 : d == 1 ? -s.ctus_w - 1 : d == 2 ? -s.ctus_w : -s.ctus_w + 1;
    const unsigned *flag = J.ctu_done + (size_t)ci.cy * s.ctus_w + ci.cx + at;
    bool waiting = lane < 4 && ((ci.deps >> d) & 1) != 0;
    bool timed_out = false;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
      if (waiting && hr_flag_read(flag) != 0) waiting = false;
      if (!__any(waiting)) break;
      // Bounded: after 20 s without progress (or, where there is no clock, 2^26 polls) the neighbour is taken
      // never to finish (a corrupt list) and the launch gives up instead of hanging.
      if ((++spins & 0xfff) == 0) {
        const unsigned long long now = hr_clock100();
        if (t0 == 0) t0 = now;
        if (now - t0 > 2000000000ull || spins >= (1u << 26)) {
          timed_out = true;
          if (waiting && atomicOr(J.error_flag + 1, 1) == 0) {  // first report: which entry waited for which neighbour
            J.error_flag[2] = (int)entry_idx;
            J.error_flag[3] = (int)entry;
            J.error_flag[4] = d;
            J.error_flag[5] = (int)hr_flag_read(ticket);
            J.error_flag[6] = n_items;
            J.error_flag[7] = ci.cx | (ci.cy << 8) | (ci.deps << 16);
          }
          break;
        }
      }
      hr_nap();
    }
    if (lane == 0) {
      if (spins > 4096) atomicMax(J.error_flag + 8, (int)spins);  // longest wait of the launch, for tuning the bound
      if (timed_out) atomicOr(J.error_flag, HR_ERR_INTRA_TIMEOUT);
    }
    hr_acquire();
  }
  __syncthreads();
  HR_STAMP(1)

  // ---- stage the CTU and its top / left borders in LDS ----
  // A picture of intra blocks only writes every sample of the CTU before reading it, so only the borders are needed.
  {
    // The border loads are independent of each other (one round trip to memory in total); the CTU's own samples,
    // where needed, are in registers already.
    if constexpr (INSIDE) {
      if (need_inside) {
        // 16 bytes of the picture -> SPV tile samples, two per dword, stored as 8-byte pieces
        auto put = [](uint16_t *t, const uint4 &qv) {
          const unsigned w[4] = {qv.x, qv.y, qv.z, qv.w};
          if constexpr (sizeof(pel_t) == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++)
              *(uint2 *)(t + 4 * j) = make_uint2((w[j] & 0xffu) | ((w[j] & 0xff00u) << 8), ((w[j] >> 16) & 0xffu) | ((w[j] >> 24) << 16));
          } else {
            *(uint2 *)t = make_uint2(w[0], w[1]);
            *(uint2 *)(t + 4) = make_uint2(w[2], w[3]);
          }
        };
#pragma unroll
        for (int k = 0; k < NV_Y; k++) {
          const int v = tid + k * IT_NTHR, ty = v / VPR_Y, tx = (v % VPR_Y) * SPV;
          if (v < 64 * VPR_Y && ctux + tx < s.w && ctuy + ty < s.h) put(&tile_y[(ty + 1) * IT_YW + tx + IT_X0], pre_in_y[k]);
        }
#pragma unroll
        for (int k = 0; k < NV_C; k++) {
          const int v = tid + k * IT_NTHR;
          const int c = v / (32 * VPR_C), r = v % (32 * VPR_C), ty = r / VPR_C, tx = (r % VPR_C) * SPV;
          if (v < 2 * 32 * VPR_C && (ctux >> 1) + tx < s.cw && (ctuy >> 1) + ty < s.ch) put(&tile_c[c][(ty + 1) * IT_CW + tx + IT_X0], pre_in_c[k]);
        }
      }
    }
    {
      // border: luma row above (x = -1 .. 95) by threads 0..96, luma column left (y = 0 .. 63) by threads 128..191;
      // chroma rows above (x = -1 .. 47) by threads 0..48 / 64..112, chroma columns left by threads 192..223 / 224..255
      const pel_t *py = (const pel_t *)J.rec[0];
      if (tid < 97) {
        const int gx = ctux + tid - 1, gy = ctuy - 1;
        if (gx >= 0 && gy >= 0 && gx < s.w) tile_y[tid + IT_X0 - 1] = py[(size_t)gy * s.stride_y + gx];
      }
      if (tid >= 128 && tid < 192) {
        const int gx = ctux - 1, gy = ctuy + tid - 128;
        if (gx >= 0 && gy < s.h) tile_y[(tid - 128 + 1) * IT_YW + IT_X0 - 1] = py[(size_t)gy * s.stride_y + gx];
      }
      {
        const int c = tid >> 6, i = tid & 63;
        if (c < 2 && i < 49) {
          const int gx = (ctux >> 1) + i - 1, gy = (ctuy >> 1) - 1;
          if (gx >= 0 && gy >= 0 && gx < s.cw) tile_c[c][i + IT_X0 - 1] = ((const pel_t *)J.rec[1 + c])[(size_t)gy * s.stride_c + gx];
        }
      }
      if (tid >= 192 && tid < 256) {
        const int c = (tid - 192) >> 5, j = (tid - 192) & 31;
        const int gx = (ctux >> 1) - 1, gy = (ctuy >> 1) + j;
        if (gx >= 0 && gy < s.ch) tile_c[c][(j + 1) * IT_CW + IT_X0 - 1] = ((const pel_t *)J.rec[1 + c])[(size_t)gy * s.stride_c + gx];
      }
    }
    // residual of the CTU's coded intra blocks (positions without one hold garbage and are never read), block list
    // and rounds: fetched above
    *(uint4 *)&resi_y[(tid >> 3) * 64 + (tid & 7) * 8] = pre_ry;
    *(uint2 *)&resi_c[tid >> 8][((tid & 255) >> 3) * 32 + (tid & 7) * 4] = pre_rc;
    if (tid < nit) *(uint4 *)&items_s[tid] = pre_item;
    if (tid < ci.n_rounds && tid < IT_MAX_ROUNDS) rounds_s[tid] = (uint16_t)pre_lvl;
  }
  __syncthreads();
  HR_STAMP(2)

  const int n_rounds = ci.n_rounds < IT_MAX_ROUNDS ? ci.n_rounds : IT_MAX_ROUNDS;
  // A round's description and the block each wave takes in it are fetched one round ahead (the reads are issued
  // before the barrier that ends the round before, and used after it).
  unsigned rword = n_rounds > 0 ? hr_uniform((unsigned)rounds_s[0]) : 0u;
  uint4 q = make_uint4(0, 0, 0, 0);
  {
    const int b0 = (int)(rword & IR_BASE_MASK) + wv;
    if (wv < (int)((rword >> IR_COUNT_SHIFT) & IR_COUNT_MASK)) q = *(const uint4 *)&items_s[b0 < nit ? b0 : 0];
  }
  for (int r = 0; r < n_rounds; r++) {
      // The round: its first block and how many (the packer's table).  One wave per block builds the block's
      // reference samples and predicts it: one pass of the wave for blocks up to 16x16, four for a 32x32 block, and
      // the round costs one barrier.  (Handing the passes of a 32x32 block to the other waves through LDS, at the
      // price of a second barrier, measured the same.)
      const int rbase = rword & IR_BASE_MASK, nb = min((int)((rword >> IR_COUNT_SHIFT) & IR_COUNT_MASK), IT_WAVES);
      const unsigned rword_next_v = rounds_s[r + 1 < n_rounds ? r + 1 : r];
      HR_STAMP(8)
      // (which wave takes which block rotates from round to round: workgroups that share a compute unit then spread
      // their busy waves over its four SIMDs instead of all loading the first ones)
      const int slot = (wv + 3 * r) & (IT_WAVES - 1);
      if (slot < nb && rbase + slot < nit) {
        // The block's description is the same in every lane: keep it in scalar registers.
        const unsigned w0 = hr_uniform(q.x), w1 = hr_uniform(q.y), w2 = hr_uniform(q.z), w3 = hr_uniform(q.w);
        const int x0 = w0 & 0xffff, y0 = w0 >> 16;
        const int log2n = w1 & 255, comp = (w1 >> 8) & 255, mode = (w1 >> 16) & 255, flags = w1 >> 24;
        const int sh = comp ? 1 : 0;
        const int tw = comp ? IT_CW : IT_YW;
        // position of the block inside the tile (tile row 0 = the row above the CTU, tile column IT_X0 = its first column)
        const int tx0 = x0 - (ctux >> sh) + IT_X0, ty0 = y0 - (ctuy >> sh) + 1;
        const int dcv = intra_references(comp ? tile_c[comp - 1] : tile_y, tw, (ty0 - 1) * tw + tx0 - 1, log2n, comp, mode, flags, w2, w3,
                                         comp ? s.bd_c : s.bd_y, s.strong_intra, ref_s[slot], lane);
        hr_wave_sync();  // the reference samples were written by other lanes of this wave
        const IntraBlk B = intra_blk(s, tile_y, tile_c[0], tile_c[1], resi_y, resi_c[0], resi_c[1], ctux, ctuy, w0, w1, w3, dcv, ref_s[slot], inv_tab);
        if (log2n == 5) {  // this entry's share of the block's 256 groups
          const int part = (int)(w3 >> IP_PART_SHIFT), share = 256 / HR_INTRA_SPLIT32;
          intra_predict<4>(B, part * share + lane, 64, (part + 1) * share);
        } else if (log2n == 4)
          intra_predict<4>(B, lane, 64, 64);
        else
          intra_predict<1>(B, lane, 64, 1 << (2 * log2n));
      }
      HR_STAMP(9)
      {
        // the next round's description and this wave's block of it: read now, used after the barrier
        rword = hr_uniform(rword_next_v);
        const int slot_n = (wv + 3 * (r + 1)) & (IT_WAVES - 1), b_n = (int)(rword & IR_BASE_MASK) + slot_n;
        if (r + 1 < n_rounds && slot_n < (int)((rword >> IR_COUNT_SHIFT) & IR_COUNT_MASK)) q = *(const uint4 *)&items_s[b_n < nit ? b_n : 0];
      }
      __syncthreads();  // the next round reads what this one wrote into the tile and reuses the reference arrays
      HR_STAMP(12)
#ifdef HEVCRECON_INTRA_STAMPS
      if (tid == 0) acc_[7] += 1;
#endif
  }
  HR_STAMP(3)
  // ---- write the CTU back (in P/B pictures the inter samples staged above go back unchanged) ----
  // Sixteen bytes per store, written through to memory: a neighbouring CTU may be waiting for these samples in another
  // workgroup of this launch.  (Rows are padded to a multiple of 128 samples, so a store that begins inside the
  // picture may end in the row's padding; it never reaches into another CTU.)
  {
    constexpr int SPS = 16 / sizeof(pel_t);  // samples per store
    // 16 bytes of a tile row (SPS samples, two per dword) -> SPS samples of the picture
    auto pack16b = [](const uint16_t *t) -> uint4 {
      const uint2 *q = (const uint2 *)t;
      if constexpr (sizeof(pel_t) == 1) {
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint2 d = q[j];  // samples 4j .. 4j+3
          o[j] = (d.x & 0xffu) | ((d.x >> 8) & 0xff00u) | ((d.y & 0xffu) << 16) | ((d.y & 0xff0000u) << 8);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
      } else {
        const uint2 a = q[0], b = q[1];
        return make_uint4(a.x, a.y, b.x, b.y);
      }
    };
    pel_t *py = (pel_t *)J.rec[0];
    for (int i = tid; i < 64 * 64 / SPS; i += IT_NTHR) {
      const int ty = i / (64 / SPS), tx = (i % (64 / SPS)) * SPS;
      const int gx = ctux + tx, gy = ctuy + ty;
      if (gx < s.w && gy < s.h) hr_store_through16(py + (size_t)gy * s.stride_y + gx, pack16b(&tile_y[(ty + 1) * IT_YW + tx + IT_X0]));
    }
    for (int i = tid; i < 2 * 32 * 32 / SPS; i += IT_NTHR) {
      const int c = i / (32 * 32 / SPS), r = i % (32 * 32 / SPS);
      const int ty = r / (32 / SPS), tx = (r % (32 / SPS)) * SPS;
      const int gx = (ctux >> 1) + tx, gy = (ctuy >> 1) + ty;
      if (gx < s.cw && gy < s.ch)
        hr_store_through16((pel_t *)J.rec[1 + c] + (size_t)gy * s.stride_c + gx, pack16b(&tile_c[c][(ty + 1) * IT_CW + tx + IT_X0]));
    }
  }
  HR_STAMP(4)
  hr_drain_stores();
  __syncthreads();
  if (tid == 0) hr_flag_raise(J.ctu_done + (size_t)ci.cy * s.ctus_w + ci.cx);
  HR_STAMP(5)
#ifdef HEVCRECON_INTRA_STAMPS
  if (tid == 0) acc_[6] += 1;
#endif
  }  // next CTU
#ifdef HEVCRECON_INTRA_STAMPS
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) atomicAdd(&hr_intra_stamps[k], acc_[k]);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_deblock: one lane per 4-sample edge segment
// ------------------------------------------------------------------------------------------------
// a 16-byte motion record through a global pointer
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ Part4 hr_load_part4(const Part4 *p) { return *p; }
#else
__device__ __forceinline__ Part4 hr_load_part4(const HR_GLOBAL Part4 *p) {
  const hr_u32x4 q = *(const HR_GLOBAL hr_u32x4 *)p;
  Part4 r;
  __builtin_memcpy(&r, &q, sizeof(r));
  return r;
}
#endif
__device__ __forceinline__ int tu_size_of(const Part4 &p) { return (64 >> (p.flags & P4_DEPTH_MASK)) >> (p.tu & P4_TRIDX_MASK); }

// TComLoopFilter::xGetBoundaryStrengthSingle (TLibCommon/TComLoopFilter.cpp:421-543)
__device__ __forceinline__ int boundary_strength(const PicJob &J, const Part4 &P, const Part4 &Q, bool tu_edge) {
  if ((P.flags | Q.flags) & P4_INTRA) return 2;
  if (tu_edge && ((P.tu | Q.tu) & P4_CBF_Y)) return 1;
  const int none = -0x7fffffff;
  const int rp0 = P.ref_idx[0] < 0 ? none : J.ref_poc[0][P.ref_idx[0]];
  const int rq0 = Q.ref_idx[0] < 0 ? none : J.ref_poc[0][Q.ref_idx[0]];
#define FAR(a, b) (iabs((a)[0] - (b)[0]) >= 4 || iabs((a)[1] - (b)[1]) >= 4)
  if (J.slice_type == 0) {
    const int rp1 = P.ref_idx[1] < 0 ? none : J.ref_poc[1][P.ref_idx[1]];
    const int rq1 = Q.ref_idx[1] < 0 ? none : J.ref_poc[1][Q.ref_idx[1]];
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
      if (rp0 != rp1) {
        if (rp0 == rq0) return FAR(P.mv[0], Q.mv[0]) || FAR(P.mv[1], Q.mv[1]);
        return FAR(P.mv[0], Q.mv[1]) || FAR(P.mv[1], Q.mv[0]);
      }
      return (FAR(P.mv[0], Q.mv[1]) || FAR(P.mv[1], Q.mv[0])) && (FAR(P.mv[0], Q.mv[0]) || FAR(P.mv[1], Q.mv[1]));
    }
    return 1;
  }
  return rp0 != rq0 || FAR(P.mv[0], Q.mv[0]);
#undef FAR
}

// DIR 0: vertical edges (filtering across x), DIR 1: horizontal edges.  Edges lie on the 8-sample
// grid and change at most 3 samples on either side while reading 4, so all segments of one direction
// are independent and the pass runs in place (TComLoopFilter::loopFilterPic, TLibCommon/TComLoopFilter.cpp:142-174,
// does all vertical edges of the picture before any horizontal one; so do the two launches).
// Edge selection: xSetEdgefilterTU / xSetEdgefilterPU / xSetLoopfilterParam (:276-419) for one slice, one tile.
// Luma filter: xEdgeFilterLuma / xPelFilterLuma (:546-660,809-868); chroma: xEdgeFilterChroma / xPelFilterChroma (:663-792,879-900).
template <typename pel_t, int DIR>
__global__ __launch_bounds__(256) void k_deblock(const PicJob *jobs, SeqInfo s) {
  const PicJob &J = jobs[blockIdx.y];
  if (J.deblock_disable) return;
  // segment grid: along the edge in 4-sample units, across in 8-sample steps
  const int n_edge = DIR == 0 ? s.w / 8 : s.h / 8;  // edge positions (position 0 = picture border, skipped)
  const int n_unit = DIR == 0 ? s.uh : s.uw;
  const int fast_n = DIR == 0 ? n_edge : n_unit;     // fastest-varying index runs along x in both cases
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int fi = gid % fast_n, si = gid / fast_n;
  const int e = DIR == 0 ? fi : si, u = DIR == 0 ? si : fi;
  if (e == 0 || e >= n_edge || u >= n_unit) return;
  const int x = DIR == 0 ? e * 8 : u * 4, y = DIR == 0 ? u * 4 : e * 8;
  const int ux = x >> 2, uy = y >> 2;
  const HR_GLOBAL Part4 *r4 = (const HR_GLOBAL Part4 *)J.rec4;
  const Part4 Q = hr_load_part4(r4 + (size_t)uy * s.uw + ux);
  const Part4 P = hr_load_part4(DIR == 0 ? r4 + (size_t)uy * s.uw + ux - 1 : r4 + (size_t)(uy - 1) * s.uw + ux);
  const int pos = DIR == 0 ? x : y;
  const bool tu_edge = (pos & (tu_size_of(Q) - 1)) == 0;
  if (!tu_edge) {
    // prediction-block edge inside the coding unit?
    const int cus = 64 >> (Q.flags & P4_DEPTH_MASK);
    const int rel = pos & (cus - 1);
    const int ps = (Q.flags >> P4_PARTSIZE_SHIFT) & 7;
    int pe = -1;
    if (DIR == 0) {
      if (ps == 2 || ps == 3) pe = cus >> 1;
      if (ps == 6) pe = cus >> 2;
      if (ps == 7) pe = cus - (cus >> 2);
    } else {
      if (ps == 1 || ps == 3) pe = cus >> 1;
      if (ps == 4) pe = cus >> 2;
      if (ps == 5) pe = cus - (cus >> 2);
    }
    if (rel != pe) return;
  }
  // Slices with parameters of their own, slices or tiles that are not filtered across: the CTU holding the edge's Q
  // side decides (pcCU->getSlice() and m_stLFCUParam.bLeftEdge / bTopEdge of TComLoopFilter::xSetLoopfilterParam,
  // TLibCommon/TComLoopFilter.cpp:363-419; slices end at CTU boundaries)
  int tc_off_div2 = J.tc_off_div2, beta_off_div2 = J.beta_off_div2;
  if (J.ctu_info) {
    const CtuInfo ci = J.ctu_info[(y >> 6) * s.ctus_w + (x >> 6)];
    if (ci.flags & CI_DEBLOCK_OFF) return;
    if ((pos & 63) == 0 && !(ci.flags & (DIR == 0 ? CI_EDGE_LEFT : CI_EDGE_TOP))) return;
    tc_off_div2 = ci.tc_off_div2;
    beta_off_div2 = ci.beta_off_div2;
  }
  const int bs = boundary_strength(J, P, Q, tu_edge);
  if (!bs) return;
  // a side that belongs to a lossless coding unit is left as it is (bPartPNoFilter / bPartQNoFilter,
  // TLibCommon/TComLoopFilter.cpp:633-642,856-867; TComDataCU::isLosslessCoded, TLibCommon/TComDataCU.cpp:1703-1706)
  const bool keep_p = (P.flags & P4_BYPASS) != 0, keep_q = (Q.flags & P4_BYPASS) != 0;
  const int qp = (P.qp + Q.qp + 1) >> 1;
  {
    const int scale = 1 << (s.bd_y - 8), maxv = (1 << s.bd_y) - 1;
    const int tc = kTc[clip3i(0, 53, qp + 2 * (bs - 1) + 2 * tc_off_div2)] * scale;
    const int beta = kBeta[clip3i(0, 51, qp + 2 * beta_off_div2)] * scale;
    const int step = DIR == 0 ? 1 : s.stride_y, line = DIR == 0 ? s.stride_y : 1;
    HR_GLOBAL pel_t *sp = (HR_GLOBAL pel_t *)J.rec[0] + (size_t)y * s.stride_y + x;
    int m[4][8];  // m[k][0..3] = p3..p0, m[k][4..7] = q0..q3 of line k
    if constexpr (sizeof(pel_t) == 1) {
      // byte planes: the 4x8 (8x4) samples around the segment as aligned dwords (x is a multiple of 4)
      if (DIR == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned lo = *(const HR_GLOBAL unsigned *)(sp + k * line - 4), hi = *(const HR_GLOBAL unsigned *)(sp + k * line);
#pragma unroll
          for (int i = 0; i < 4; i++) m[k][i] = (lo >> (8 * i)) & 255, m[k][4 + i] = (hi >> (8 * i)) & 255;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const unsigned w = *(const HR_GLOBAL unsigned *)(sp + (i - 4) * step);
#pragma unroll
          for (int k = 0; k < 4; k++) m[k][i] = (w >> (8 * k)) & 255;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) m[k][i] = sp[k * line + (i - 4) * step];
    }
    const int dp0 = iabs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = iabs(m[0][4] - 2 * m[0][5] + m[0][6]);
    const int dp3 = iabs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = iabs(m[3][4] - 2 * m[3][5] + m[3][6]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
    if (d < beta) {
      const int side = (beta + (beta >> 1)) >> 3;
      const bool fp = dp0 + dp3 < side, fq = dq0 + dq3 < side;
      const bool sw0 = iabs(m[0][0] - m[0][3]) + iabs(m[0][7] - m[0][4]) < (beta >> 3) && 2 * d0 < (beta >> 2) &&
                       iabs(m[0][3] - m[0][4]) < ((tc * 5 + 1) >> 1);
      const bool sw3 = iabs(m[3][0] - m[3][3]) + iabs(m[3][7] - m[3][4]) < (beta >> 3) && 2 * d3 < (beta >> 2) &&
                       iabs(m[3][3] - m[3][4]) < ((tc * 5 + 1) >> 1);
      const bool sw = sw0 && sw3;
      int r[4][8];  // results; r[k][0] and r[k][7] (p3, q3) never change
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int m0 = m[k][0], m1 = m[k][1], m2 = m[k][2], m3 = m[k][3], m4 = m[k][4], m5 = m[k][5], m6 = m[k][6], m7 = m[k][7];
#pragma unroll
        for (int i = 0; i < 8; i++) r[k][i] = m[k][i];
        if (sw) {
          if (!keep_p) {
            r[k][3] = clip3i(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
            r[k][2] = clip3i(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
            r[k][1] = clip3i(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
          }
          if (!keep_q) {
            r[k][4] = clip3i(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
            r[k][5] = clip3i(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
            r[k][6] = clip3i(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
          }
        } else {
          int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
          if (iabs(delta) < tc * 10) {
            delta = clip3i(-tc, tc, delta);
            const int tc2 = tc >> 1;
            if (!keep_p) {
              r[k][3] = clip3i(0, maxv, m3 + delta);
              if (fp) r[k][2] = clip3i(0, maxv, m2 + clip3i(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
            }
            if (!keep_q) {
              r[k][4] = clip3i(0, maxv, m4 - delta);
              if (fq) r[k][5] = clip3i(0, maxv, m5 + clip3i(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
            }
          }
        }
      }
      // write back.  Samples p3 / q3 and, for byte planes, whole dwords are rewritten with their old values where
      // nothing changed: no other segment of this pass touches them (edges are 8 samples apart and change at most 3
      // samples on either side), so this is not a race.
      if constexpr (sizeof(pel_t) == 1) {
        if (DIR == 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            *(HR_GLOBAL unsigned *)(sp + k * line - 4) = pack4(r[k][0], r[k][1], r[k][2], r[k][3]);
            *(HR_GLOBAL unsigned *)(sp + k * line) = pack4(r[k][4], r[k][5], r[k][6], r[k][7]);
          }
        } else {
#pragma unroll
          for (int i = 1; i < 7; i++) *(HR_GLOBAL unsigned *)(sp + (i - 4) * step) = pack4(r[0][i], r[1][i], r[2][i], r[3][i]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int i = 1; i < 7; i++)
            if (r[k][i] != m[k][i]) sp[k * line + (i - 4) * step] = (pel_t)r[k][i];
      }
    }
  }
  // chroma: only strength 2, only edges on the 8-chroma-sample grid; this unit owns 2 chroma lines
  if (bs == 2 && (pos & 15) == 0) {
    const int scale = 1 << (s.bd_c - 8), maxv = (1 << s.bd_c) - 1;
    const int step = DIR == 0 ? 1 : s.stride_c, line = DIR == 0 ? s.stride_c : 1;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int qc = qp + (c == 0 ? J.cb_off_pps : J.cr_off_pps);
      qc = qc < 0 ? qc : (qc > 57 ? qc - 6 : kChromaScale[qc]);
      const int tc = kTc[clip3i(0, 53, qc + 2 + 2 * tc_off_div2)] * scale;
      HR_GLOBAL pel_t *sp = (HR_GLOBAL pel_t *)J.rec[1 + c] + (size_t)(y >> 1) * s.stride_c + (x >> 1);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        HR_GLOBAL pel_t *o = sp + k * line;
        const int m2 = o[-2 * step], m3 = o[-step], m4 = o[0], m5 = o[step];
        const int delta = clip3i(-tc, tc, ((((m4 - m3) * 4) + m2 - m5 + 4) >> 3));
        if (!keep_p) o[-step] = (pel_t)clip3i(0, maxv, m3 + delta);
        if (!keep_q) o[0] = (pel_t)clip3i(0, maxv, m4 - delta);
      }
    }
  }
}

// Does any lane of the wave that is still in the kernel hold `p`?  (The CPU stand-in's wave vote waits for every lane
// of the wave, also those that have left the kernel, so there the answer is the safe one.)
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ bool hr_any_lane(bool) { return true; }
#else
__device__ __forceinline__ bool hr_any_lane(bool p) { return __any(p) != 0; }
#endif

// Two 16-bit values per dword (v_pk_* instructions), and byte selection across two dwords (v_perm_b32: selector byte
// 0-3 picks a byte of lo, 4-7 a byte of hi, 0x0c gives zero).
#ifdef HEVCRECON_EMULATE
#define HR_PK2(name, T, expr)                                                          \
  __device__ __forceinline__ unsigned name(unsigned a, unsigned b) {                   \
    const T x0 = (T)(a & 0xffffu), x1 = (T)(a >> 16), y0 = (T)(b & 0xffffu), y1 = (T)(b >> 16); \
    unsigned r0, r1;                                                                   \
    { const T x = x0, y = y0; r0 = (unsigned)(uint16_t)(expr); }                       \
    { const T x = x1, y = y1; r1 = (unsigned)(uint16_t)(expr); }                       \
    return r0 | (r1 << 16);                                                            \
  }
HR_PK2(hr_pk_add, uint16_t, x + y)
HR_PK2(hr_pk_sub, uint16_t, x - y)
HR_PK2(hr_pk_max_i16, int16_t, x > y ? x : y)
HR_PK2(hr_pk_min_i16, int16_t, x < y ? x : y)
HR_PK2(hr_pk_lshr, uint16_t, x >> (y & 15))
#undef HR_PK2
__device__ __forceinline__ unsigned hr_perm(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned c = (sel >> (8 * i)) & 255u;
    r |= (c < 8 ? (unsigned)((v >> (8 * c)) & 255u) : 0u) << (8 * i);
  }
  return r;
}
#else
typedef short hr_i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short hr_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hr_pk_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (hr_u16x2)(__builtin_bit_cast(hr_u16x2, a) + __builtin_bit_cast(hr_u16x2, b)));
}
__device__ __forceinline__ unsigned hr_pk_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (hr_u16x2)(__builtin_bit_cast(hr_u16x2, a) - __builtin_bit_cast(hr_u16x2, b)));
}
__device__ __forceinline__ unsigned hr_pk_max_i16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(hr_i16x2, a), __builtin_bit_cast(hr_i16x2, b)));
}
__device__ __forceinline__ unsigned hr_pk_min_i16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(hr_i16x2, a), __builtin_bit_cast(hr_i16x2, b)));
}
__device__ __forceinline__ unsigned hr_pk_lshr(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (hr_u16x2)(__builtin_bit_cast(hr_u16x2, a) >> __builtin_bit_cast(hr_u16x2, b)));
}
__device__ __forceinline__ unsigned hr_perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#endif

// ------------------------------------------------------------------------------------------------
// k_sao: sample adaptive offset, deblocked picture -> final picture
// ------------------------------------------------------------------------------------------------
// TComSampleAdaptiveOffset::processSaoUnitAll / processSaoCuOrg (TLibCommon/TComSampleAdaptiveOffset.cpp:814-1039,
// 1097-1261).  The reference filters in place and saves the row above / column left of each CTU so that
// classification always sees deblocked samples; reading a separate source picture is the same thing.
// blockIdx.z = component.  One lane handles 4 consecutive samples of a row.
template <typename pel_t>
__global__ __launch_bounds__(256) void k_sao(const PicJob *jobs, SeqInfo s) {
  // grid: x = 64-lane chunks of a row (16 samples per lane), y = groups of 4 rows, z = picture * 3 + component
  const PicJob &J = jobs[blockIdx.z / 3];
  const int comp = blockIdx.z % 3;
  if (J.rec[comp] == J.out[comp]) return;  // picture without SAO: reconstructed in place, nothing to do
  const int pw = comp ? s.cw : s.w, ph = comp ? s.ch : s.h, stride = comp ? s.stride_c : s.stride_y;
  const int bd = comp ? s.bd_c : s.bd_y, maxv = (1 << bd) - 1;
  const int x = (blockIdx.x * 64 + threadIdx.x) * 16, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= pw || y >= ph) return;
  const HR_GLOBAL pel_t *src = (const HR_GLOBAL pel_t *)J.rec[comp];
  HR_GLOBAL pel_t *dst = (HR_GLOBAL pel_t *)J.out[comp];
  const int csh = comp ? 5 : 6;
  const SaoCtu sp = J.sao[((size_t)(y >> csh) * s.ctus_w + (x >> csh)) * 3 + comp];
  const bool on = (comp ? J.sao_chroma : J.sao_luma) && sp.type >= 0;
  const int inc = bd > 10 ? bd - 10 : 0;
  // Everything that depends on the CTU's parameters is data, not control flow (the 64 lanes of a wave span 16 CTUs):
  //   edge offset: neighbour a lies towards (-dx,-dy), b towards (+dx,+dy); class = sign(c-a) + sign(c-b) + 2 picks a byte
  //                of the table (offsets 0 and 1, zero, offsets 2 and 3); rows at the top / bottom picture border and lanes
  //                without SAO get a table of zero offsets, the two samples at the left / right border are put back afterwards
  //   band offset: band index relative to the first band picks a byte of the table (the four offsets, then zero)
  // Table bytes hold (offset << inc) + 128 (offsets are at most 31 << 2 in magnitude), so that they are unsigned.
  const bool eo = on && sp.type < 4, bo = on && sp.type == 4;
  const int dx = !eo || sp.type == 1 ? 0 : (sp.type == 3 ? -1 : 1), dy = eo && sp.type != 0 ? 1 : 0;
  const bool rows_ok = dy == 0 || (y > 0 && y + 1 < ph);
  auto biased = [&](int k) -> unsigned { return (unsigned)((int)sp.offset[k] * (1 << inc) + 128) & 255u; };
  const unsigned kNeutral = 0x80808080u;
  unsigned tlo = kNeutral, thi = kNeutral;
  if (eo && rows_ok) {
    tlo = biased(0) | (biased(1) << 8) | (128u << 16) | (biased(2) << 24);
    thi = biased(3) | 0x80808000u;
  }
  if (bo) tlo = biased(0) | (biased(1) << 8) | (biased(2) << 16) | (biased(3) << 24);
  const int dyr = dy && rows_ok ? stride : 0;
  const HR_GLOBAL pel_t *row = src + (size_t)y * stride;
  // A row's samples x-4/B .. x+16+4/B-1 as packed dwords (B = bytes per sample): one dword left of the lane's 16
  // samples, the samples, one dword right.  Reads past the picture width stay inside the padded row (or the slack
  // after the last plane) and are never used: such neighbours make the sample a border sample.
  constexpr int SPD = 4 / sizeof(pel_t);   // samples per dword
  constexpr int ND = 16 / SPD;             // dwords holding the lane's 16 samples
  auto fetch = [&](const HR_GLOBAL pel_t *rp, unsigned w[ND + 2]) {
    w[0] = x > 0 ? *(const HR_GLOBAL unsigned *)(rp + x - SPD) : 0u;
#pragma unroll
    for (int k = 0; k < ND / 4; k++) hr_load16(rp + x + 4 * SPD * k, w + 1 + 4 * k);
    w[ND + 1] = *(const HR_GLOBAL unsigned *)(rp + x + 16);
  };
  // the 16 samples starting `shift` samples (-1, 0, 1) from the lane's first one, still packed
  auto shifted = [&](const unsigned w[ND + 2], int shift, unsigned o[ND]) {
    const bool back = shift < 0;
    const int bytes = shift == 0 ? 0 : (back ? 4 - (int)sizeof(pel_t) : (int)sizeof(pel_t));
    unsigned lo[ND + 1];
#pragma unroll
    for (int k = 0; k <= ND; k++) lo[k] = back ? w[k] : w[k + 1];
#pragma unroll
    for (int k = 0; k < ND; k++) o[k] = hr_alignbyte(lo[k + 1], lo[k], bytes);
  };
  auto sample = [&](const unsigned w[ND], int i) -> int {
    return sizeof(pel_t) == 1 ? (int)((w[i >> 2] >> (8 * (i & 3))) & 255u) : (int)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
  };
  unsigned wc[ND + 2], wa[ND + 2], wb[ND + 2], pc[ND], pa[ND], pb[ND];
  fetch(row, wc);
  fetch(row - dyr, wa);
  fetch(row + dyr, wb);
  shifted(wc, 0, pc);
  shifted(wa, -dx, pa);
  shifted(wb, dx, pb);

  // Slices or tiles that are not filtered across (TComSampleAdaptiveOffset::processSaoBlock,
  // TLibCommon/TComSampleAdaptiveOffset.cpp:592-809): a sample is classified only when both neighbours along the
  // class direction lie in its own CTU or in a neighbouring CTU that may be used; the others stay as they are.
  // Picture borders are unusable neighbours in the same table.  Only the first and the last sample of a CTU's row
  // can have a neighbour to the side; the others share one answer.
  bool ok_mid = true, ok_first = true, ok_last = true, at_left = false;
  int i_last = -1;
  if (J.ctu_info && eo) {
    const int cs = comp ? 32 : 64;
    const unsigned am = J.ctu_info[(y >> csh) * s.ctus_w + (x >> csh)].sao_avail;
    auto usable = [&](int ox, int oy) -> bool {
      if (ox == 0 && oy == 0) return true;
      const int bit = oy == 0 ? (ox < 0 ? 0 : 1) : ox == 0 ? (oy < 0 ? 2 : 3) : (oy < 0 ? (ox < 0 ? 4 : 5) : (ox < 0 ? 6 : 7));
      return ((am >> bit) & 1u) != 0;
    };
    const bool top = (y & (cs - 1)) == 0, bottom = (y & (cs - 1)) == cs - 1 || y == ph - 1;
    const int oy_a = dy && top ? -1 : 0, oy_b = dy && bottom ? 1 : 0;  // neighbour a lies towards (-dx,-dy), b towards (+dx,+dy)
    ok_mid = usable(0, oy_a) && usable(0, oy_b);
    at_left = (x & (cs - 1)) == 0;                 // the lane's first sample is the first of its CTU's row
    i_last = ((x + 16) & (cs - 1)) == 0 ? 15 : -1;  // index of the last sample of the CTU's row, when the lane holds it
    if (x + 16 >= pw) i_last = pw - 1 - x;
    ok_first = usable(dx == 1 ? -1 : 0, oy_a) && usable(dx == -1 ? -1 : 0, oy_b);
    ok_last = usable(dx == -1 ? 1 : 0, oy_a) && usable(dx == 1 ? 1 : 0, oy_b);
  }
  const unsigned rlo = tlo, rhi = thi;  // the CTU's own table, for the rare sample that is classified although its row is not
  if (!ok_mid) tlo = thi = kNeutral;

  // Classification and offset on two samples per dword (16 bits each): 8-bit planes are split into even and odd
  // samples first.  sign(c - a) + 1 = min(max(c + 1 - a, 0), 2).
  constexpr int NP = 8;
  unsigned cp[NP], ap[NP], bp[NP], r[NP];
  if constexpr (sizeof(pel_t) == 1) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cp[2 * k] = hr_perm(0u, pc[k], 0x0c020c00u), cp[2 * k + 1] = hr_perm(0u, pc[k], 0x0c030c01u);
      ap[2 * k] = hr_perm(0u, pa[k], 0x0c020c00u), ap[2 * k + 1] = hr_perm(0u, pa[k], 0x0c030c01u);
      bp[2 * k] = hr_perm(0u, pb[k], 0x0c020c00u), bp[2 * k + 1] = hr_perm(0u, pb[k], 0x0c030c01u);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NP; k++) cp[k] = pc[k], ap[k] = pa[k], bp[k] = pb[k];
  }
  const unsigned maxvp = (unsigned)maxv * 0x00010001u;
  const unsigned bshp = (unsigned)(bd - 5) * 0x00010001u, band0p = ((unsigned)sp.band & 0xffffu) * 0x00010001u;
  const bool any_bo = hr_any_lane(bo);
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const unsigned c1 = hr_pk_add(cp[k], 0x00010001u);
    const unsigned sa = hr_pk_min_i16(hr_pk_max_i16(hr_pk_sub(c1, ap[k]), 0u), 0x00020002u);
    const unsigned sb = hr_pk_min_i16(hr_pk_max_i16(hr_pk_sub(c1, bp[k]), 0u), 0x00020002u);
    unsigned cls = hr_pk_add(sa, sb);
    if (any_bo) {
      const unsigned band = hr_pk_min_i16(hr_pk_sub(hr_pk_lshr(cp[k], bshp), band0p) & 0x001f001fu, 0x00040004u);
      cls = bo ? band : cls;
    }
    const unsigned e = hr_perm(thi, tlo, cls | 0x0c000c00u);
    r[k] = hr_pk_min_i16(hr_pk_max_i16(hr_pk_add(hr_pk_sub(cp[k], 0x00800080u), e), 0u), maxvp);
  }
  unsigned out[ND];
  if constexpr (sizeof(pel_t) == 1) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = hr_perm(r[2 * k + 1], r[2 * k], 0x06020400u);
  } else {
#pragma unroll
    for (int k = 0; k < NP; k++) out[k] = r[k];
  }
  // sample i of the result replaced by `val`
  auto put = [&](int i, int val) {
    constexpr int BITS = 8 * (int)sizeof(pel_t);
    const int k = i / SPD, sh = BITS * (i % SPD);
    out[k] = (out[k] & ~((BITS == 8 ? 0xffu : 0xffffu) << sh)) | ((unsigned)val << sh);
  };
  // one sample the plain way, with the CTU's own table
  auto one = [&](int i) -> int {
    const int c = sample(pc, i);
    const int cls = clip3i(-1, 1, c - sample(pa, i)) + clip3i(-1, 1, c - sample(pb, i)) + 2;
    const int e = (int)(((((unsigned long long)rhi << 32) | rlo) >> (8 * cls)) & 255u);
    return clip3i(0, maxv, c - 128 + e);
  };
  const int i_pic_last = pw - 1 - x;  // index of the picture's last sample of this row, when the lane holds it (< 16)
  if (dx != 0) {  // picture-border samples of the horizontal and diagonal classes stay as they are
    if (x == 0) put(0, sample(pc, 0));
    if (i_pic_last == 15) {
      put(15, sample(pc, 15));
    } else if (i_pic_last < 15) {
#pragma unroll
      for (int i = 0; i < 15; i++)
        if (i == i_pic_last) put(i, sample(pc, i));
    }
  }
  if (J.ctu_info && eo && rows_ok) {
    if (at_left && ok_first != ok_mid) put(0, ok_first ? one(0) : sample(pc, 0));
    if (i_last >= 0 && ok_last != ok_mid) {
      if (i_last == 15) {
        put(15, ok_last ? one(15) : sample(pc, 15));
      } else {
#pragma unroll
        for (int i = 0; i < 15; i++)
          if (i == i_last) put(i, ok_last ? one(i) : sample(pc, i));
      }
    }
  }
  if (J.has_bypass) {
    // TComSampleAdaptiveOffset::xPCMRestoration (TLibCommon/TComSampleAdaptiveOffset.cpp:1394-1515): after SAO the
    // samples of lossless coding units are put back to their reconstruction, which deblocking has not touched either
    const int ush = comp ? 1 : 2;  // log2 of the plane's samples per 4x4 luma unit
    const int uy = y >> ush;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int ux = (x + i) >> ush;
      if (ux < s.uw && (J.rec4[(size_t)uy * s.uw + ux].flags & P4_BYPASS)) put(i, sample(pc, i));
    }
  }
  HR_GLOBAL pel_t *orow = dst + (size_t)y * stride + x;
  if constexpr (sizeof(pel_t) == 1) {
    if (x + 16 <= pw) {
      hr_store16(orow, out[0], out[1], out[2], out[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (x + 4 * q < pw) *(HR_GLOBAL unsigned *)(orow + 4 * q) = out[q];
    }
  } else {
    if (x + 16 <= pw) {
      hr_store16(orow, out[0], out[1], out[2], out[3]);
      hr_store16(orow + 8, out[4], out[5], out[6], out[7]);
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (x + i < pw) orow[i] = (pel_t)sample(out, i);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_expand: run-length coded motion records -> rec4.  One workgroup per CTU, one lane per 4x4 unit.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand(const PicJob *jobs, SeqInfo s) {
  __shared__ __attribute__((aligned(16))) Part4 runs_s[256];  // the CTU's runs: searched and read here, one trip to memory for all of it
  const PicJob &J = jobs[blockIdx.y];
  if (!J.need_expand) return;
  const Part4 *runs = J.runs;
  const uint32_t *ctu_run_off = J.ctu_run_off;
  Part4 *rec4 = const_cast<Part4 *>(J.rec4);
  const int ctu = blockIdx.x, z = threadIdx.x;
  const unsigned first = ctu_run_off[ctu], n = min(ctu_run_off[ctu + 1] - first, 256u);
  if ((unsigned)z < n) runs_s[z] = runs[first + z];
  __syncthreads();
  int x = 0, y = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    x |= ((z >> (2 * b)) & 1) << b;
    y |= ((z >> (2 * b + 1)) & 1) << b;
  }
  const int ux = (ctu % s.ctus_w) * 16 + x, uy = (ctu / s.ctus_w) * 16 + y;
  if (ux >= s.uw || uy >= s.uh || n == 0) return;
  // last run that starts at or before z (runs of a CTU are sorted by their first unit; at most 256 of them)
  int lo = 0, hi = (int)n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (runs_s[mid].pad <= z) lo = mid; else hi = mid - 1;
  }
  Part4 r = runs_s[lo];
  if (z < r.pad || z > r.pad + r.pad2[0]) {  // not covered: cannot happen with a well-formed list
    atomicOr(J.error_flag, HR_ERR_EXPAND);
    return;
  }
  r.pad = 0;
  r.pad2[0] = 0;
  rec4[(size_t)uy * s.uw + ux] = r;
}

// ------------------------------------------------------------------------------------------------
// k_output: a picture as the reference's file writer stores it (TVideoIOYuv::write / writePlane / scalePlane /
// invScalePlane, TLibVideoIO/TVideoIOYuv.cpp:62-126,305-340,422-489): planes back to back, rows without padding,
// samples scaled from the internal to the file bit depth (down: rounded and clipped to the file range; up: shifted),
// one byte per sample, or two (low byte first) when either file depth exceeds 8.  One lane per sample.
// ------------------------------------------------------------------------------------------------
template <typename pel_t, bool WIDE>
__global__ __launch_bounds__(256) void k_output(const pel_t *src, int stride, uint8_t *dst, int width, int height, int shift, int maxv) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= width || y >= height) return;
  int v = src[(size_t)y * stride + x];
  if (shift > 0)
    v = clip3i(0, maxv, (v + (1 << (shift - 1))) >> shift);
  else
    v = v << (-shift);
  if (WIDE)
    ((uint16_t *)dst)[(size_t)y * width + x] = (uint16_t)v;
  else
    dst[(size_t)y * width + x] = (uint8_t)v;
}

// ------------------------------------------------------------------------------------------------
// k_md5: the picture hash of the reference (calcMD5, TLibCommon/TComPicYuvMD5.cpp:188-213; MD5 itself: RFC 1321,
// the reference's copy is libmd5/libmd5.c): every plane's samples in raster order, one byte per sample up to 8 bits,
// else two bytes, low byte first -- which is how the planes lie in memory, row padding aside.
// MD5 is one serial chain per plane (64 dependent steps per 64-byte block), so a plane is one lane's work: lane 0 of
// the plane's wave.  The other 63 lanes fetch for it: the wave reads 64 blocks at a time (lane = block, all 16 words
// of a block in flight at once), parks them in LDS while lane 0 works through the previous 64, and lane 0 takes its
// message words from LDS a block ahead -- the chain never waits for memory.
// ------------------------------------------------------------------------------------------------
#ifdef HEVCRECON_EMULATE
__device__ __forceinline__ unsigned hr_rotl(unsigned v, int sh) { return (v << sh) | (v >> (32 - sh)); }
#else
__device__ __forceinline__ unsigned hr_rotl(unsigned v, int sh) { return __builtin_amdgcn_alignbit(v, v, 32 - sh); }  // one v_alignbit_b32
#endif
struct Md5Tables {
  unsigned k[64];
  int s[64], g[64];
};
constexpr Md5Tables make_md5_tables() {
  const unsigned k[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
      0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
      0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
      0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
      0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
      0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
  const int sh[16] = {7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21};
  Md5Tables t = {};
  for (int i = 0; i < 64; i++) {
    t.k[i] = k[i];
    t.s[i] = sh[(i >> 4) * 4 + (i & 3)];
    t.g[i] = i < 16 ? i : i < 32 ? (5 * i + 1) & 15 : i < 48 ? (3 * i + 5) & 15 : (7 * i) & 15;
  }
  return t;
}

// Word j of message block blk of a plane: its bytes, then the padding of RFC 1321 (0x80, zeros, the length in bits).
__device__ __forceinline__ void md5_block_words(const Md5Plane &pl, unsigned len, unsigned nblocks, unsigned blk, unsigned w[16]) {
  unsigned pos = blk * 64u;
  unsigned row = pos / pl.row_bytes, col = pos - row * pl.row_bytes;  // (row_bytes is a multiple of 4: a word never straddles rows)
#pragma unroll
  for (int j = 0; j < 16; j++, pos += 4) {
    unsigned v = 0;
    if (pos < len) {
      v = *(const HR_GLOBAL unsigned *)((const HR_GLOBAL char *)pl.base + (size_t)row * pl.stride_bytes + col);
      col += 4;
      if (col == pl.row_bytes) col = 0, row++;
    } else if (pos == len) {
      v = 0x80u;
    } else if (blk == nblocks - 1 && j == 14) {
      v = len << 3;
    } else if (blk == nblocks - 1 && j == 15) {
      v = len >> 29;
    }
    w[j] = v;
  }
}

__global__ __launch_bounds__(64) void k_md5(const Md5Plane *planes, int n, uint8_t *digests) {
  constexpr Md5Tables T = make_md5_tables();
  __shared__ __attribute__((aligned(16))) unsigned buf[2][64][16];
  const int lane = threadIdx.x;
  const int idx = blockIdx.x;
  if (idx >= n) return;
  const Md5Plane pl = planes[idx];
  const unsigned len = pl.row_bytes * pl.rows;  // at most 8192 x 8192 samples of two bytes
  const unsigned nblocks = (len + 8) / 64 + 1;
  const unsigned nchunks = (nblocks + 63) / 64;
  unsigned a0 = 0x67452301u, b0 = 0xefcdab89u, c0 = 0x98badcfeu, d0 = 0x10325476u;
  unsigned w[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = 0;
  if ((unsigned)lane < nblocks) md5_block_words(pl, len, nblocks, lane, w);
  for (unsigned chunk = 0; chunk < nchunks; chunk++) {
    unsigned(*cur)[16] = buf[chunk & 1];
#pragma unroll
    for (int j = 0; j < 16; j += 4) *(uint4 *)&cur[lane][j] = make_uint4(w[j], w[j + 1], w[j + 2], w[j + 3]);
    hr_wave_sync();
    // the next 64 blocks are requested now and arrive while lane 0 works
    const unsigned nb = (chunk + 1) * 64 + lane;
    if (chunk + 1 < nchunks && nb < nblocks) md5_block_words(pl, len, nblocks, nb, w);
    if (lane == 0) {
      const unsigned left = nblocks - chunk * 64, here = left < 64u ? left : 64u;
      unsigned m[16], mn[16];
#pragma unroll
      for (int j = 0; j < 16; j++) m[j] = cur[0][j];
      for (unsigned b = 0; b < here; b++) {
        const unsigned bn = b + 1 < here ? b + 1 : b;
#pragma unroll
        for (int j = 0; j < 16; j++) mn[j] = cur[bn][j];
        unsigned a = a0, bb = b0, c = c0, d = d0;
#pragma unroll
        for (int i = 0; i < 64; i++) {
          unsigned f;
          if (i < 16) f = d ^ (bb & (c ^ d));
          else if (i < 32) f = c ^ (d & (bb ^ c));
          else if (i < 48) f = bb ^ c ^ d;
          else f = c ^ (bb | ~d);
          f = f + (a + T.k[i] + m[T.g[i]]);
          a = d, d = c, c = bb;
          bb = bb + hr_rotl(f, T.s[i]);
        }
        a0 += a, b0 += bb, c0 += c, d0 += d;
#pragma unroll
        for (int j = 0; j < 16; j++) m[j] = mn[j];
      }
    }
    hr_wave_sync();
  }
  if (lane == 0) {
    unsigned *out = (unsigned *)(digests + (size_t)idx * 16);
    out[0] = a0, out[1] = b0, out[2] = c0, out[3] = d0;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)

template <typename pel_t>
static void launch_all_t(int kind, const PicJob *jobs, int n, const SeqInfo &s, hipStream_t st, int arg, int arg2) {
  switch (kind) {
    case K_INTER:
      // arg = 0..2: uniform regions of 64 / 32 / 16 samples, 3: 8x8 blocks unit by unit, +4: some picture of the batch uses
      // explicit weighted prediction; arg2 = the largest count of such regions over the batch.  Tile grid.x is a
      // multiple of 8, so that blockIdx.x & 7 is the XCD for every picture of the batch.
      {
        const dim3 g((arg2 + 7) / 8 * 8, n);
        if (arg == 0) LAUNCH((k_inter_tile<pel_t, 64, 512, false>), g, dim3(512), st, jobs, s);
        if (arg == 1) LAUNCH((k_inter_tile<pel_t, 32, 128, false>), g, dim3(128), st, jobs, s);
        if (arg == 2) LAUNCH((k_inter_tile<pel_t, 16, 64, false>), g, dim3(64), st, jobs, s);
        if (arg == 4) LAUNCH((k_inter_tile<pel_t, 64, 512, true>), g, dim3(512), st, jobs, s);
        if (arg == 5) LAUNCH((k_inter_tile<pel_t, 32, 128, true>), g, dim3(128), st, jobs, s);
        if (arg == 6) LAUNCH((k_inter_tile<pel_t, 16, 64, true>), g, dim3(64), st, jobs, s);
      }
      if (arg == 3) LAUNCH((k_inter_blk<pel_t, false>), dim3((arg2 + 63) / 64, n), dim3(256), st, jobs, s);
      if (arg == 7) LAUNCH((k_inter_blk<pel_t, true>), dim3((arg2 + 63) / 64, n), dim3(256), st, jobs, s);
      // arg = 8, 9 (with explicit weights): 8x8 blocks of one motion, 8-bit pictures only (the packer makes none otherwise)
      if constexpr (sizeof(pel_t) == 1) {
        if (arg == 8) LAUNCH((k_inter_blk8<false>), dim3((arg2 + 15) / 16, n), dim3(256), st, jobs, s);
        if (arg == 9) LAUNCH((k_inter_blk8<true>), dim3((arg2 + 15) / 16, n), dim3(256), st, jobs, s);
      }
      break;
    case K_TRANSFORM: {
      // arg = 5: the 32x32 blocks, arg2 = their largest count over the batch; arg = 0: the smaller blocks, arg2 = the
      // largest number of workgroups a picture of the batch needs for them
      if (arg == 5) LAUNCH((k_transform<pel_t, 5>), dim3((arg2 + 7) / 8, n), dim3(256), st, jobs, s);
      if (arg == 0) LAUNCH((k_transform_small<pel_t>), dim3(arg2, n), dim3(256), st, jobs, s);
      break;
    }
    case K_OTHER:
      LAUNCH(k_expand, dim3(s.ctus_w * s.ctus_h, n), dim3(256), st, jobs, s);
      break;
    case K_DEBLOCK_V:
      LAUNCH((k_deblock<pel_t, 0>), dim3(((s.w / 8) * s.uh + 255) / 256, n), dim3(256), st, jobs, s);
      break;
    case K_DEBLOCK_H:
      LAUNCH((k_deblock<pel_t, 1>), dim3(((s.h / 8) * s.uw + 255) / 256, n), dim3(256), st, jobs, s);
      break;
    case K_SAO:
      LAUNCH(k_sao<pel_t>, dim3((s.w / 16 + 64) / 64, (s.h + 3) / 4, n * 3), dim3(64, 4), st, jobs, s);
      break;
    default:
      break;
  }
}

void hr_launch_output(const void *const planes[3], void *dst, const SeqInfo &s, int width, int height, int file_bd_y, int file_bd_c,
                      bool wide, hipStream_t st) {
  uint8_t *out = (uint8_t *)dst;
  const int ws = wide ? 2 : 1;
  for (int k = 0; k < 3; k++) {
    const int w = k ? width / 2 : width, h = k ? height / 2 : height, stride = k ? s.stride_c : s.stride_y;
    const int bd = k ? s.bd_c : s.bd_y, fbd = k ? file_bd_c : file_bd_y;
    const int shift = bd - fbd, maxv = (1 << fbd) - 1;
    const dim3 g((w + 255) / 256, h);
    if (s.bd_y <= 8 && s.bd_c <= 8) {
      if (wide) LAUNCH((k_output<uint8_t, true>), g, dim3(256), st, (const uint8_t *)planes[k], stride, out, w, h, shift, maxv);
      else LAUNCH((k_output<uint8_t, false>), g, dim3(256), st, (const uint8_t *)planes[k], stride, out, w, h, shift, maxv);
    } else {
      if (wide) LAUNCH((k_output<uint16_t, true>), g, dim3(256), st, (const uint16_t *)planes[k], stride, out, w, h, shift, maxv);
      else LAUNCH((k_output<uint16_t, false>), g, dim3(256), st, (const uint16_t *)planes[k], stride, out, w, h, shift, maxv);
    }
    out += (size_t)w * h * ws;
  }
}

void hr_launch_intra(const PicJob *jobs, const SeqInfo &s, const unsigned *list, int n_items, int max_parallel, bool dense, bool inside,
                     unsigned *ticket, hipStream_t st) {
  // No more workgroups than CTUs that can ever be worked on at a time (the largest wave of the batch), and no more
  // than fill the device: a workgroup that has drawn a CTU holds its place on the device while it waits for that
  // CTU's neighbours.  One workgroup per compute unit as a rule: more of them crowd out the kernels of other engine
  // contexts on the device (random-access streams measured 4 % slower with three).  A batch of intra pictures, where
  // this kernel is most of the work, gets three for every two units: a chain of dependent LDS round trips per
  // block leaves a unit's SIMDs idle most of the time, and a second context does the same (1080p all-intra, two
  // contexts: 37.7 k -> 41.3 k frames/s).  HEVCRECON_INTRA_GRID overrides both.
  int grid = n_items < max_parallel ? n_items : max_parallel;
  static const int forced = std::getenv("HEVCRECON_INTRA_GRID") ? std::atoi(std::getenv("HEVCRECON_INTRA_GRID")) : 0;
  const int cap = forced > 0 ? forced : dense ? 384 : 256;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  const bool wide = s.bd_y > 8 || s.bd_c > 8;
  if (inside) {
    if (!wide)
      LAUNCH((k_intra<uint8_t, true>), dim3(grid), dim3(IT_NTHR), st, jobs, s, list, n_items, ticket);
    else
      LAUNCH((k_intra<uint16_t, true>), dim3(grid), dim3(IT_NTHR), st, jobs, s, list, n_items, ticket);
  } else {
    if (!wide)
      LAUNCH((k_intra<uint8_t, false>), dim3(grid), dim3(IT_NTHR), st, jobs, s, list, n_items, ticket);
    else
      LAUNCH((k_intra<uint16_t, false>), dim3(grid), dim3(IT_NTHR), st, jobs, s, list, n_items, ticket);
  }
}

void hr_launch_md5(const Md5Plane *planes, int n, uint8_t *digests, hipStream_t st) {
  LAUNCH(k_md5, dim3(n), dim3(64), st, planes, n, digests);  // one wave per plane
}

void hr_launch(int kind, const PicJob *jobs, int n, const SeqInfo &s, hipStream_t st, int arg, int arg2) {
  if (s.bd_y <= 8 && s.bd_c <= 8)
    launch_all_t<uint8_t>(kind, jobs, n, s, st, arg, arg2);
  else
    launch_all_t<uint16_t>(kind, jobs, n, s, st, arg, arg2);
}
