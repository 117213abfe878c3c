// h2d_ceiling.hip — what the host link of this machine carries from pinned memory: the ceiling the syntax hand-over is
// measured against.  One 256 MB hipMemcpyAsync, the same bytes as copies of a packed picture's size (180 KB) and of a
// batch's (8 MB) on one stream and spread over three, and device-to-host for comparison.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/h2d_ceiling.hip -o tools/ubench/bin/h2d_ceiling
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t total = (size_t)256 << 20;
  char *host = nullptr, *dev = nullptr;
  CHECK(hipHostMalloc((void **)&host, total));
  CHECK(hipMalloc((void **)&dev, total));
  memset(host, 1, total);
  hipStream_t st[3];
  for (int i = 0; i < 3; i++) CHECK(hipStreamCreate(&st[i]));
  CHECK(hipMemcpy(dev, host, total, hipMemcpyHostToDevice));  // warm-up
  printf("{");
  const size_t pieces[4] = {total, (size_t)8 << 20, (size_t)1 << 20, (size_t)180 << 10};
  const char *names[4] = {"256MB", "8MB", "1MB", "180KB"};
  bool first = true;
  for (int dir = 0; dir < 2; dir++)
    for (int ns = 1; ns <= 3; ns += 2)
      for (int k = 0; k < 4; k++) {
        const size_t piece = pieces[k];
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          CHECK(hipDeviceSynchronize());
          const double t0 = now();
          int n = 0;
          for (size_t off = 0; off + piece <= total; off += piece, n++) {
            if (dir == 0) CHECK(hipMemcpyAsync(dev + off, host + off, piece, hipMemcpyHostToDevice, st[n % ns]));
            else CHECK(hipMemcpyAsync(host + off, dev + off, piece, hipMemcpyDeviceToHost, st[n % ns]));
          }
          for (int i = 0; i < ns; i++) CHECK(hipStreamSynchronize(st[i]));
          const double dt = now() - t0;
          if (dt < best) best = dt;
        }
        const size_t moved = total / piece * piece;
        printf("%s\"%s_%s_%dstream_gbps\": %.2f", first ? "" : ", ", dir ? "d2h" : "h2d", names[k], ns, moved / best / 1e9);
        first = false;
      }
  printf("}\n");
  return 0;
}
