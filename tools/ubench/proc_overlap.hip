// proc_overlap.hip — do kernels of different PROCESSES overlap on this GPU?  Each process launches `reps` kernels of one
// wave that spin for `ms` milliseconds (on the 100 MHz clock) and prints the wall time of its loop; run N copies at
// once and compare with one alone.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/proc_overlap.hip -o tools/ubench/bin/proc_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin(unsigned long long ticks, unsigned *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) n++;
  if (threadIdx.x == 0) *out = n;
}

int main(int argc, char **argv) {
  const int ms = argc > 1 ? atoi(argv[1]) : 50, reps = argc > 2 ? atoi(argv[2]) : 10, streams = argc > 3 ? atoi(argv[3]) : 1;
  unsigned *out;
  if (hipMalloc((void **)&out, 64) != hipSuccess) return 1;
  hipStream_t st[8];
  for (int i = 0; i < streams && i < 8; i++) hipStreamCreate(&st[i]);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[0], 1000ull, out);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[r % streams], (unsigned long long)ms * 100000ull, out);
  hipDeviceSynchronize();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%d kernels of %d ms on %d stream(s): %.3f s\n", reps, ms, streams, dt);
  return 0;
}
