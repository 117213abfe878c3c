// Micro-benchmark: issue rate of the integer instructions the interpolation kernels could be built from.
// Every kernel runs N_IT iterations of 32 independent operations per lane; the table printed is
// cycles per wave-instruction per SIMD at full occupancy (8 waves per SIMD) and with one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 2048
typedef short s2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
  unsigned a[8], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x01020304u;
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 17 + i;
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) a[i] = __builtin_amdgcn_sdot4((int)b, (int)c, (int)a[i], false);
        if (OP == 1) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, b), __builtin_bit_cast(s2, c), (int)a[i], false);
        if (OP == 2) a[i] = (unsigned)(__mul24((int)(b & 0xffff), (int)(c & 0xffff)) + (int)a[i]);  // v_mad_i32_i24
        if (OP == 3) a[i] = a[i] * b + c;                                       // 32-bit multiply-add
        if (OP == 4) a[i] = __builtin_amdgcn_alignbyte(a[i], b, 1);
        if (OP == 5) a[i] = (a[i] ^ b) + c;                                     // two simple ALU ops
        if (OP == 6) {                                                          // v_pk_mad_i16
          s2 x = __builtin_bit_cast(s2, a[i]), y = __builtin_bit_cast(s2, b), z = __builtin_bit_cast(s2, c);
          x = x * y + z;
          a[i] = __builtin_bit_cast(unsigned, x);
        }
        if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], b, c);
        if (OP == 8) a[i] = __builtin_amdgcn_udot4(b, c, a[i], false);
      }
    b += a[0];
  }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(int waves_per_simd, unsigned *d) {
  // 256 CUs x 4 SIMDs; a 256-thread block is 4 waves = one per SIMD of a CU
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, 2);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)N_IT * 32 * waves_per_simd;  // wave-instructions issued on one SIMD
  return ms * 1e-3 * 2.4e9 / insts_per_simd;                           // cycles at a nominal 2.4 GHz
}

int main() {
  unsigned *d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  const char *names[] = {"v_dot4_i32_i8", "v_dot2_i32_i16", "v_mad_i32_i24", "mul_lo+add u32", "v_alignbyte", "xor+add (2 ops)", "v_pk_mad_i16", "v_perm_b32", "v_dot4_u32_u8"};
  double r8[9], r1[9];
  r8[0] = run<0>(8, d); r1[0] = run<0>(1, d);
  r8[1] = run<1>(8, d); r1[1] = run<1>(1, d);
  r8[2] = run<2>(8, d); r1[2] = run<2>(1, d);
  r8[3] = run<3>(8, d); r1[3] = run<3>(1, d);
  r8[4] = run<4>(8, d); r1[4] = run<4>(1, d);
  r8[5] = run<5>(8, d); r1[5] = run<5>(1, d);
  r8[6] = run<6>(8, d); r1[6] = run<6>(1, d);
  r8[7] = run<7>(8, d); r1[7] = run<7>(1, d);
  r8[8] = run<8>(8, d); r1[8] = run<8>(1, d);
  printf("%-18s %14s %14s\n", "operation", "cyc/op 8w/SIMD", "cyc/op 1w/SIMD");
  for (int i = 0; i < 9; i++) printf("%-18s %14.2f %14.2f\n", names[i], r8[i], r1[i]);
  return 0;
}
