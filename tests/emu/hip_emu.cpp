// hip_emu.cpp — thread pool behind hip_emu.h (test/debug harness only, see that header).
#include "hip_emu.h"
#include <chrono>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

namespace {
const int kMaxThreads = 1024;

struct Pool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  int generation = 0, remaining = 0, active = 0;
  bool stop = false;
  const std::function<void()> *body = nullptr;
  dim3 grid, block, bidx;
  Block blk;

  Pool() {
    for (int t = 0; t < kMaxThreads; t++) workers.emplace_back([this, t] { loop(t); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      generation++;
    }
    cv_start.notify_all();
    for (auto &w : workers) w.join();
  }
  void loop(int tid) {
    int seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_start.wait(lk, [&] { return generation != seen; });
        seen = generation;
        if (stop) return;
        if (tid >= active) continue;
      }
      threadIdx = dim3(tid % block.x, (tid / block.x) % block.y, tid / (block.x * block.y));
      blockIdx = bidx;
      blockDim = block;
      gridDim = grid;
      (*body)();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--remaining == 0) cv_done.notify_one();
      }
    }
  }
  void run_block(dim3 g, dim3 b, dim3 bi, const std::function<void()> &f) {
    int n = (int)(b.x * b.y * b.z);
    blk.nthreads = n;
    blk.block_barrier.reset(n);
    for (int w = 0; w * 64 < n; w++) blk.wave_barrier[w].reset(std::min(64, n - w * 64));
    {
      std::lock_guard<std::mutex> lk(m);
      grid = g;
      block = b;
      bidx = bi;
      body = &f;
      active = n;
      remaining = n;
      generation++;
    }
    cv_start.notify_all();
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return remaining == 0; });
  }
};

Pool &pool() {
  static Pool p;
  return p;
}
}  // namespace

Block &current_block() { return pool().blk; }

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
  if (block.x * block.y * block.z > (unsigned)kMaxThreads) std::abort();
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) pool().run_block(grid, block, dim3(x, y, z), body);
}

}  // namespace emu

double emu_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

long g_emu_allocations = 0;
// For tests of the entry point that must not allocate: how often the engine has asked the (emulated) runtime for
// device or pinned memory, an event or a stream.
extern "C" long hevcrecon_emu_allocations() { return g_emu_allocations; }
