// Micro-benchmark: bi-predicted 8-bit luma interpolation (HM's 8-tap separable filter, TComInterpolationFilter.cpp)
// formulated on the matrix cores, to be compared with the dot-product kernels of the library.
//
// One wave produces a 25x25 block of output samples from a 32x32 window of each of two reference pictures:
//   X  = P . Th        v_mfma_i32_32x32x32_i8: P = 32 rows x 32 bytes straight from memory (A operand, one
//                      unaligned 16-byte load per lane), Th = 32x32 band matrix of the horizontal taps;
//                      columns 0..24 of X are the horizontally filtered rows (exact in 32 bits)
//   Y  = Tv . X        X stays in its accumulator registers: low and high bytes are packed into two B
//                      operands (two MFMAs), Tv = band matrix of the vertical taps with its k index in the
//                      accumulator's row order; rows 0..24 of Y are valid
//   out = clip((Y0 >> 6) + (Y1 >> 6) + 16448 >> 7), one byte store per register.
// The program checks every output of the first batch entry against scalar code and prints the rate.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/bin/mc_mfma tools/ubench/mc_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

#define PITCH 2176
#define ROWS 1216
#define MARGIN 64
#define PIC_W 1920
#define PIC_H 1080
#define BLK 25
#define BLKS_X ((PIC_W + BLK - 1) / BLK)
#define BLKS_Y ((PIC_H + BLK - 1) / BLK)

static const signed char kTaps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ unsigned long long kTapWords[4];

__host__ __device__ inline unsigned hash3(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}
// motion vector of a block in quarter samples: integer part in [-8, 7], any fraction
__host__ __device__ inline void block_mv(unsigned blk, unsigned pic, int list, int *mx, int *my) {
  const unsigned h = hash3(blk, pic, (unsigned)list);
  *mx = (int)(h & 63) - 32;
  *my = (int)((h >> 6) & 63) - 32;
}

// bytes q .. q+3 of the eight tap bytes extended with zeros on both sides
__device__ inline unsigned band_word(unsigned long long taps, int q) {
  if (q <= -4 || q >= 8) return 0;
  return q >= 0 ? (unsigned)(taps >> (8 * q)) : (unsigned)(taps << (-8 * q));
}

__device__ inline v16i filter_list(const unsigned char *ref, int x0, int y0, int mx, int my, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned char *src = ref + (size_t)(MARGIN + y0 + (my >> 2) - 3 + r) * PITCH + MARGIN + x0 + (mx >> 2) - 3 + 16 * h;
  u32x4_unaligned raw = *(const u32x4_unaligned *)src;
  v4i a, th, tv;
  const unsigned long long th_taps = kTapWords[mx & 3], tv_taps = kTapWords[my & 3];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    a[d] = (int)(raw[d] ^ 0x80808080u);
    th[d] = (int)band_word(th_taps, 4 * d + 16 * h - r);      // B[k = 16h + j][n = r] = tap[k - n]
    tv[d] = (int)band_word(tv_taps, 8 * d + 4 * h - r);       // A[m = r][element j of half h] = tap[row(j, h) - m]
  }
  v16i zero = {};
  v16i x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, th, zero, 0, 0, 0);
  v4i lo, hi;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const unsigned p01 = __builtin_amdgcn_perm((unsigned)x[4 * g + 1], (unsigned)x[4 * g], 0x05040100u);
    const unsigned p23 = __builtin_amdgcn_perm((unsigned)x[4 * g + 3], (unsigned)x[4 * g + 2], 0x05040100u);
    lo[g] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
    hi[g] = (int)__builtin_amdgcn_perm(p23, p01, 0x07050301u);
  }
  v16i bias;
#pragma unroll
  for (int j = 0; j < 16; j++) bias[j] = 8192;                 // 128 * (sum of the taps) for the xor on the low bytes
  v16i ylo = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, lo, bias, 0, 0, 0);
  v16i yhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, hi, zero, 0, 0, 0);
  v16i v;
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = (ylo[j] + (yhi[j] << 8)) >> 6;
  return v;
}

__global__ __launch_bounds__(256) void k_mc_mfma(const unsigned char *ref0, const unsigned char *ref1, unsigned char *out, int n_blocks) {
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= n_blocks) return;
  const int pic = blockIdx.y;
  const int x0 = (blk % BLKS_X) * BLK, y0 = (blk / BLKS_X) * BLK;
  int mx0, my0, mx1, my1;
  block_mv(blk, pic, 0, &mx0, &my0);
  block_mv(blk, pic, 1, &mx1, &my1);
  const size_t plane = (size_t)PITCH * ROWS;
  v16i v0 = filter_list(ref0 + (size_t)pic * plane, x0, y0, mx0, my0, lane);
  v16i v1 = filter_list(ref1 + (size_t)pic * plane, x0, y0, mx1, my1, lane);
  const int n = lane & 31, h = lane >> 5;
  unsigned char *dst = out + (size_t)pic * plane + (size_t)(MARGIN + y0) * PITCH + MARGIN + x0 + n;
  const bool col_ok = n < BLK && x0 + n < PIC_W;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int m = (j & 3) + 8 * (j >> 2) + 4 * h;
    int s = (v0[j] + v1[j] + 16448) >> 7;
    s = s < 0 ? 0 : (s > 255 ? 255 : s);
    if (col_ok && m < BLK && y0 + m < PIC_H) dst[(size_t)m * PITCH] = (unsigned char)s;
  }
}

static int filt14(const unsigned char *ref, int x, int y, int mx, int my) {
  // 14-bit intermediate of one list at output sample (x, y)
  int col[8];
  for (int k = 0; k < 8; k++) {
    const unsigned char *row = ref + (size_t)(MARGIN + y + (my >> 2) - 3 + k) * PITCH + MARGIN + x + (mx >> 2) - 3;
    int s = 0;
    for (int t = 0; t < 8; t++) s += kTaps[mx & 3][t] * row[t];
    col[k] = s - 8192;
  }
  int s = 0;
  for (int k = 0; k < 8; k++) s += kTaps[my & 3][k] * col[k];
  return s >> 6;
}

int main(int argc, char **argv) {
  const int pics = argc > 1 ? atoi(argv[1]) : 32;
  const size_t plane = (size_t)PITCH * ROWS;
  std::vector<unsigned char> h0(plane * pics), h1(plane * pics), hout(plane);
  unsigned s = 12345;
  for (size_t i = 0; i < h0.size(); i++) { s = s * 1664525u + 1013904223u; h0[i] = s >> 24; h1[i] = s >> 16; }
  unsigned long long words[4];
  for (int f = 0; f < 4; f++) memcpy(&words[f], kTaps[f], 8);
  hipMemcpyToSymbol(HIP_SYMBOL(kTapWords), words, sizeof words);
  unsigned char *d0, *d1, *dout;
  if (hipMalloc(&d0, h0.size()) || hipMalloc(&d1, h1.size()) || hipMalloc(&dout, h0.size())) { printf("allocation failed\n"); return 1; }
  hipMemcpy(d0, h0.data(), h0.size(), hipMemcpyHostToDevice);
  hipMemcpy(d1, h1.data(), h1.size(), hipMemcpyHostToDevice);
  hipMemset(dout, 0, h0.size());
  const int n_blocks = BLKS_X * BLKS_Y;
  dim3 grid((n_blocks + 3) / 4, pics);
  k_mc_mfma<<<grid, 256>>>(d0, d1, dout, n_blocks);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(hout.data(), dout, plane, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int y = 0; y < PIC_H; y++)
    for (int x = 0; x < PIC_W; x++) {
      const unsigned blk = (y / BLK) * BLKS_X + x / BLK;
      int mx0, my0, mx1, my1;
      block_mv(blk, 0, 0, &mx0, &my0);
      block_mv(blk, 0, 1, &mx1, &my1);
      int v = (filt14(h0.data(), x, y, mx0, my0) + filt14(h1.data(), x, y, mx1, my1) + 16448) >> 7;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      if (hout[(size_t)(MARGIN + y) * PITCH + MARGIN + x] != v) {
        if (bad < 5) printf("mismatch at (%d, %d): %d, expected %d\n", x, y, hout[(size_t)(MARGIN + y) * PITCH + MARGIN + x], v);
        bad++;
      }
    }
  printf("check against scalar code: %ld of %d samples differ\n", bad, PIC_W * PIC_H);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) k_mc_mfma<<<grid, 256>>>(d0, d1, dout, n_blocks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_launch = ms / reps;
  const double samples = (double)PIC_W * PIC_H * pics;
  printf("bi-predicted luma on the matrix cores: %.3f ms per %d pictures of 1920x1080, %.1f G output samples/s, %.1f G list samples/s\n",
         per_launch, pics, samples / per_launch * 1e-6, 2 * samples / per_launch * 1e-6);
  return bad != 0;
}
