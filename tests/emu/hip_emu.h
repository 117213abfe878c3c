// hip_emu.h — a tiny HIP look-alike that runs kernels on CPU threads.
//
// TEST/DEBUG HARNESS ONLY.  It exists because kernels are developed on a machine without a GPU:
// compiling hevc_amd/csrc/*.hip with -DHEVCRECON_EMULATE against this header yields
// tests/emu/libhevcrecon_emu.so, which lets the kernel LOGIC (indexing, barriers, arithmetic) be
// checked against the oracle and stepped through in gdb.  The product library never includes this
// file and hevc_amd never loads the emulated library; GPU tests and the benchmark run the real
// gfx950 build only.
//
// Model: blocks run one after another; the threads of a block are real OS threads taken from a
// pool, so __syncthreads() and the 64-lane wave primitives behave as on hardware (wave = 64
// consecutive threads).  __shared__ variables are function-local statics, i.e. one instance
// shared by the threads of the running block.
#ifndef HIP_EMU_H
#define HIP_EMU_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct EmuEvent { double t; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int count = 0, waiting = 0, gen = 0;
  void reset(int n) { count = n; waiting = 0; }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    int g = gen;
    if (++waiting == count) {
      waiting = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return g != gen; });
    }
  }
};

struct Block {
  Barrier block_barrier;
  Barrier wave_barrier[16];
  uint64_t xchg[16][64];
  int nthreads;
};
Block &current_block();
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);

inline int lane() { return threadIdx.x & 63; }
inline int wave() { return threadIdx.x >> 6; }
inline int wave_size() {
  int n = current_block().nthreads - wave() * 64;
  return n > 64 ? 64 : n;
}
inline uint64_t exchange(uint64_t v, int src_lane) {
  Block &b = current_block();
  int w = wave();
  b.xchg[w][lane()] = v;
  b.wave_barrier[w].wait();
  uint64_t r = (src_lane >= 0 && src_lane < wave_size()) ? b.xchg[w][src_lane] : v;
  b.wave_barrier[w].wait();
  return r;
}

template <typename... Params, typename... Args>
void launch(void (*kernel)(Params...), dim3 grid, dim3 block, Args... args) {
  run_grid(grid, block, [=]() { kernel(args...); });
}

}  // namespace emu

inline void __syncthreads() { emu::current_block().block_barrier.wait(); }
inline int __shfl_xor(int v, int mask) { return (int)emu::exchange((uint64_t)(uint32_t)v, emu::lane() ^ mask); }
inline int __shfl(int v, int src) { return (int)emu::exchange((uint64_t)(uint32_t)v, src); }
inline unsigned long long __ballot(int pred) {
  emu::Block &b = emu::current_block();
  int w = emu::wave();
  b.xchg[w][emu::lane()] = pred ? 1 : 0;
  b.wave_barrier[w].wait();
  unsigned long long m = 0;
  for (int i = 0; i < emu::wave_size(); i++) m |= (unsigned long long)(b.xchg[w][i] & 1) << i;
  b.wave_barrier[w].wait();
  return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __mul24(int a, int b) { return a * b; }
inline int atomicOr(int *p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int *p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch(kernel, grid, block, __VA_ARGS__)

// ---- host runtime stand-ins (single "device" = host memory) ----
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
extern long g_emu_allocations;  // device / pinned allocations and event / stream creations so far (tests assert on it)
inline hipError_t hipMalloc(void **p, size_t n) { g_emu_allocations++; *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, int, hipStream_t) {
  for (size_t y = 0; y < h; y++) std::memcpy((char *)d + y * dp, (const char *)s + y * sp, w);
  return hipSuccess;
}
template <typename T> inline hipError_t hipGetSymbolAddress(void **p, T &symbol) { *p = (void *)&symbol; return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { g_emu_allocations++; *s = nullptr; return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { g_emu_allocations++; *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { g_emu_allocations++; *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
double emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t *e) { g_emu_allocations++; *e = new EmuEvent{0}; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { g_emu_allocations++; *e = new EmuEvent{0}; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }  // everything runs in order
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

#endif
