// Micro-benchmark: the 32x32 inverse transform (both passes of TComTrQuant.cpp's partialButterflyInverse32 in
// matrix form: 16-bit clip after ">> 7" and after ">> 12") on the matrix cores, for dense de-quantised blocks.
//
// One wave per block.  The block is stored column-major, so that a lane's A operand is 16 consecutive 16-bit
// coefficients; 16-bit values are split into a low byte (biased by 128 to make it signed) and a high byte, one
// v_mfma_i32_32x32x32_i8 each, and the bias is undone through the accumulator's start value.
//   pass 1:  D1[j][i] = sum_k X[k][j] * C[k][i]      (A = X transposed, B = C)      T = clip16((D1 + 64) >> 7)
//   pass 2:  Z[m][i]  = sum_j C[j][m] * T[j][i]      (A = C transposed with k in the accumulator's row order,
//                                                      B = T straight from pass 1's registers)
//   residual[i][m] = clip16((Z + 2048) >> 12): lane i holds 4 x 4 consecutive residuals of row i.
// The program checks every block against scalar code and prints the rate.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/bin/itx32_mfma tools/ubench/itx32_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Tables {
  int8_t ct[32][32];   // ct[i][k] = C[k][i]
  int colsum[32];      // 128 * sum_k C[k][i]
};
static Tables make_tables() {
  const int mag[33] = {0, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
  Tables t;
  for (int i = 0; i < 32; i++) {
    int sum = 0;
    for (int k = 0; k < 32; k++) {
      int val = 64;
      if (k > 0) {
        int m = (k * (2 * i + 1)) & 127;
        if (m > 64) m = 128 - m;
        val = m > 32 ? -mag[64 - m] : mag[m];
      }
      t.ct[i][k] = (int8_t)val;
      sum += val;
    }
    t.colsum[i] = 128 * sum;
  }
  return t;
}
__constant__ Tables kT;

__device__ inline int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__global__ __launch_bounds__(256) void k_itx32(const int16_t *coef_cm, int16_t *resi, int n_blocks) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= n_blocks) return;
  const uint4 *src = (const uint4 *)(coef_cm + (size_t)blk * 1024 + r * 32 + 16 * h);
  const uint4 w0 = src[0], w1 = src[1];
  const unsigned w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  v4i alo, ahi, c1, c2;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    alo[d] = (int)(__builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x06040200u) ^ 0x80808080u);
    ahi[d] = (int)__builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x07050301u);
    c1[d] = *(const int *)&kT.ct[r][16 * h + 4 * d];           // B[k = 16h + e][i = r] = C[k][i]
    c2[d] = *(const int *)&kT.ct[r][8 * d + 4 * h];            // A[m = r][e] = C[j(e, h)][m], j = (e & 3) + 8 (e >> 2) + 4h
  }
  v16i zero = {}, init1;
#pragma unroll
  for (int e = 0; e < 16; e++) init1[e] = kT.colsum[r] + 64;
  v16i dlo = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo, c1, init1, 0, 0, 0);
  v16i dhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi, c1, zero, 0, 0, 0);
  int t[16];
#pragma unroll
  for (int e = 0; e < 16; e++) t[e] = clip16((dlo[e] + (dhi[e] << 8)) >> 7);
  v4i blo, bhi;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const unsigned p01 = __builtin_amdgcn_perm((unsigned)t[4 * g + 1], (unsigned)t[4 * g], 0x05040100u);
    const unsigned p23 = __builtin_amdgcn_perm((unsigned)t[4 * g + 3], (unsigned)t[4 * g + 2], 0x05040100u);
    blo[g] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080u);
    bhi[g] = (int)__builtin_amdgcn_perm(p23, p01, 0x07050301u);
  }
  v16i init2;
#pragma unroll
  for (int e = 0; e < 16; e++) init2[e] = kT.colsum[(e & 3) + 8 * (e >> 2) + 4 * h] + 2048;
  v16i zlo = __builtin_amdgcn_mfma_i32_32x32x32_i8(c2, blo, init2, 0, 0, 0);
  v16i zhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(c2, bhi, zero, 0, 0, 0);
  int16_t *dst = resi + (size_t)blk * 1024 + r * 32;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    int v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = clip16((zlo[4 * g + e] + (zhi[4 * g + e] << 8)) >> 12);
    uint2 o;
    o.x = ((unsigned)v[0] & 0xffffu) | ((unsigned)v[1] << 16);
    o.y = ((unsigned)v[2] & 0xffffu) | ((unsigned)v[3] << 16);
    *(uint2 *)(dst + 8 * g + 4 * h) = o;
  }
}

int main(int argc, char **argv) {
  const int n_blocks = argc > 1 ? atoi(argv[1]) : 65536;
  const Tables tab = make_tables();
  if (hipMemcpyToSymbol(HIP_SYMBOL(kT), &tab, sizeof tab) != hipSuccess) return 1;
  std::vector<int16_t> x((size_t)n_blocks * 1024), out(x.size());
  unsigned s = 777;
  for (size_t i = 0; i < x.size(); i++) {
    s = s * 1664525u + 1013904223u;
    // every tenth block uses the whole 16-bit range, the others a range that is typical of de-quantised blocks
    x[i] = (i / 1024) % 10 == 0 ? (int16_t)(s >> 16) : (int16_t)((int)((s >> 16) & 2047) - 1024);
  }
  int16_t *dx, *dr;
  if (hipMalloc(&dx, x.size() * 2) || hipMalloc(&dr, x.size() * 2)) return 1;
  hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
  dim3 grid((n_blocks + 3) / 4);
  k_itx32<<<grid, 256>>>(dx, dr, n_blocks);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(out.data(), dr, x.size() * 2, hipMemcpyDeviceToHost);
  long bad = 0;
  const int check = n_blocks < 512 ? n_blocks : 512;
  for (int b = 0; b < check; b++) {
    const int16_t *X = &x[(size_t)b * 1024];  // X[k][j] at [j * 32 + k]
    static int T[32][32];
    for (int j = 0; j < 32; j++)
      for (int i = 0; i < 32; i++) {
        long sum = 0;
        for (int k = 0; k < 32; k++) sum += (long)X[j * 32 + k] * tab.ct[i][k];
        long v = (sum + 64) >> 7;
        T[j][i] = (int)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
      }
    for (int i = 0; i < 32; i++)
      for (int m = 0; m < 32; m++) {
        long sum = 0;
        for (int j = 0; j < 32; j++) sum += (long)T[j][i] * tab.ct[m][j];
        long v = (sum + 2048) >> 12;
        v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
        if (out[(size_t)b * 1024 + i * 32 + m] != (int16_t)v) {
          if (bad < 5) printf("block %d (%d, %d): %d, expected %ld\n", b, i, m, out[(size_t)b * 1024 + i * 32 + m], v);
          bad++;
        }
      }
  }
  printf("check against scalar code: %ld of %d residuals differ\n", bad, check * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 20;
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; i++) k_itx32<<<grid, 256>>>(dx, dr, n_blocks);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_launch = ms / reps;
  printf("32x32 inverse transform on the matrix cores: %.3f ms per %d blocks, %.2f M blocks/s, %.1f G residuals/s, %.0f GB/s read+written\n",
         per_launch, n_blocks, n_blocks / per_launch * 1e-3, n_blocks * 1024.0 / per_launch * 1e-6, n_blocks * 4096.0 / per_launch * 1e-6);
  return bad != 0;
}
