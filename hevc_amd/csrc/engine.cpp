// engine.cpp — the C ABI of include/hevcrecon.h: device pictures (DPB), per-picture syntax objects
// (host packer + HBM copy), and the launch sequence that reconstructs a batch of pictures.
//
// Launch sequence per batch (one HIP stream, in order):
//   k_inter  ->  k_transform x4 sizes  ->  k_intra x levels  ->  k_deblock V  ->  k_deblock H  ->  k_sao
// Inter prediction goes first because intra blocks of P/B pictures read their inter neighbours'
// reconstruction; availability for intra is decided by the packer from coding order, not by what
// happens to be in memory, so reconstructing inter blocks early cannot change results.
#ifdef HEVCRECON_EMULATE
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <mutex>
#include <string>
#include <deque>
#include <vector>
#include "../../include/hevcrecon.h"
#include "common.h"
#include "kernels.h"
#include "packer.h"

// Fixed work area of hevcrecon_reconstruct_noalloc: job descriptor, working picture and residual planes of ONE
// picture, for callers that must not allocate.  Every GPU the library runs on holds its own copy of the array, at
// its own address, until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress: the address kernels and copies use is the one that call returns (g_scratch_ptr), looked up
// once per device in hevcrecon_create.
#define HR_SCRATCH_BYTES ((size_t)16 << 20)
static __device__ __align__(256) unsigned char scratch[HR_SCRATCH_BYTES];

namespace {
unsigned char *g_scratch_ptr[64];  // per device; guarded by g_copy_mutex
// Who used the device's fixed work area last, and the event recorded behind that call's last kernel: a call from
// another context makes its stream wait for it before anything is written to the array.  Guarded by g_copy_mutex.
struct ScratchUser {
  hevcrecon_ctx *ctx;
  hipEvent_t done;
};
ScratchUser g_scratch_user[64];
const size_t kScratchJobBytes = 4096;    // one PicJob
const size_t kScratchSyncBytes = 61440;  // the intra kernel's scheduling words (ticket counter + one word per CTU)
const int kMaxPics = 1 << 16;
const int kRing = 16;
// Copy streams.  A copy command costs the copy engine ~10 us whatever its size; pictures are ~0.2 MB, so one stream
// would spend most of its time on that.  Copies of a batch go round-robin over several streams: three, which with
// the kernels' stream makes the four hardware queues a process gets (measured with-upload rate of the benchmark:
// 1 stream 53 k, 3 streams 64 k, 4 streams 65 k, 8 streams 44 k frames/s; a kernel reading the pinned buffers
// itself reached only 10 GB/s over the link).
const int kCopyStreams = 3;
// How many of them take copies: one.  A batch's kernels wait for an event on every copy stream that holds copies of
// its pictures, and each such cross-stream wait costs the kernels' stream more than the copy engine gains from a
// second queue (1080p hand-over of 128 streams: 74.3 k frames/s with three, 82.0 k with two, 82.3 k with one).
// HEVCRECON_COPY_STREAMS=1..3 for experiments.
static const int g_copy_used = [] {
  const char *e = std::getenv("HEVCRECON_COPY_STREAMS");
  const int n = e ? std::atoi(e) : 1;
  return n < 1 ? 1 : n > kCopyStreams ? kCopyStreams : n;
}();

// The copy streams belong to the device, not to a context: several contexts on one GPU (each with a kernel stream of
// its own) share them, so that the process stays near its four hardware queues.
struct SharedCopyStreams {
  hipStream_t s[kCopyStreams];
  int refs;
};
SharedCopyStreams g_copy[64];
std::mutex g_copy_mutex;

bool acquire_copy_streams(int device, hipStream_t out[kCopyStreams]) {
  if (device < 0 || device >= 64) return false;
  std::lock_guard<std::mutex> lock(g_copy_mutex);
  SharedCopyStreams &g = g_copy[device];
  if (g.refs == 0)
    for (int i = 0; i < kCopyStreams; i++)
      if (hipStreamCreate(&g.s[i]) != hipSuccess) {
        for (int k = 0; k < i; k++) hipStreamDestroy(g.s[k]);
        return false;
      }
  g.refs++;
  for (int i = 0; i < kCopyStreams; i++) out[i] = g.s[i];
  return true;
}

// Device address of this device's copy of the fixed work area (the device must be current).
bool lookup_scratch(int device) {
  if (device < 0 || device >= 64) return false;
  std::lock_guard<std::mutex> lock(g_copy_mutex);
  if (g_scratch_ptr[device]) return true;
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) {
    (void)hipGetLastError();
    return false;
  }
  g_scratch_ptr[device] = (unsigned char *)p;
  return true;
}

void release_copy_streams(int device) {
  std::lock_guard<std::mutex> lock(g_copy_mutex);
  SharedCopyStreams &g = g_copy[device];
  if (g.refs > 0 && --g.refs == 0)
    for (int i = 0; i < kCopyStreams; i++) hipStreamDestroy(g.s[i]);
}

struct DevPicture {
  void *base = nullptr;
  void *p[3] = {nullptr, nullptr, nullptr};
  int poc = 0;
  bool valid = false;
};

// One hevcrecon_picture_md5_submit in flight: copies of the pictures to hash (the originals may be overwritten as soon
// as the call returns: a decoder recycles its picture buffers faster than a hash kernel runs), descriptors and
// digests on the device, digests and the kernels' error words in pinned memory, and the event behind the copy that
// brings them back.
struct Md5Slot {
  int64_t ticket = -1;  // -1: free
  int n = 0;
  void *dev = nullptr;
  uint8_t *host = nullptr;
  void *snap = nullptr;  // n pictures, pic_bytes apart
  size_t cap = 0;        // pictures the descriptor and digest buffers hold
  size_t snap_cap = 0;   // pictures the copies hold
  size_t snap_pic_bytes = 0;
  hipEvent_t done = nullptr;
};
const int kMd5Slots = 64;

struct Interval {
  int kind;
  hipEvent_t a, b;
};
}  // namespace

struct hevcrecon_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream[kCopyStreams] = {};  // syntax uploads run beside the reconstruction kernels
  bool have_copy_streams = false;
  int copy_rr = 0;
  bool have_seq = false;
  hevcrecon_seq_params sp;
  SeqInfo si;
  int pel_bytes = 1;
  size_t plane_off[3] = {0, 0, 0}, pic_bytes = 0;
  size_t resi_off[3] = {0, 0, 0}, resi_bytes = 0;
  std::vector<DevPicture> pics;
  std::vector<DevPicture> work;
  std::vector<void *> work_resi;  // per batch slot: int16 residual planes, same geometry as a picture
  DevPicture stage[2];
  bool keep_stages = false;
  // Scheduling words of the intra kernel (a ticket counter, then one "done" word per CTU of every picture of the
  // batch), zeroed before each launch, and the pinned ring + device copy of the batch's CTU list.  Both grow on demand.
  uint32_t *sync_dev = nullptr;
  size_t sync_words = 0;
  uint32_t *list_host = nullptr, *list_dev = nullptr;  // kRing slots of list_cap entries
  size_t list_cap = 0;
  PicJob *jobs_host = nullptr;  // pinned ring: kRing slots of HR_MAX_BATCH jobs
  PicJob *jobs_dev = nullptr;
  hipEvent_t ring_ev[kRing] = {};
  bool ring_used[kRing];
  // "batch done" events: slot b % kRing is recorded behind batch b's last kernel.  A later batch's event in the
  // same slot also implies batch b is done (one stream), so waiting on the slot is always sufficient.
  hipEvent_t batch_ev[kRing] = {};
  int64_t batch_no = 0;
  // Cross-stream waits are expensive, and both streams run in order: the main stream waits for an upload only
  // when it is newer than the newest one it has already waited for, and the copy streams likewise for batches.
  int64_t upload_no = 0, upload_waited = -1, upload_synced = -1, batch_waited = -1;
  hipEvent_t copy_ev[kRing * kCopyStreams] = {};  // recorded on the copy streams when the main stream has to wait for uploads
  int copy_ev_pos = 0;
  int ring_pos = 0;
  // Upload ring: packed pictures are staged in one pinned buffer in the order they are finished, mirrored by one
  // device buffer at the same offsets, so that the pictures of a batch, finished one after the other, cross the host
  // link in a few large copies instead of one copy command each.  Space is handed out in ring order and taken back
  // in the same order; a picture that does not fit (the ring is full of pictures still in use) gets buffers of its own.
  struct RingEntry {
    size_t off, bytes;
    bool live;
  };
  char *up_host = nullptr, *up_dev = nullptr;
  size_t up_bytes = (size_t)64 << 20, up_head = 0;
  bool up_failed = false;
  std::deque<RingEntry> up_entries;  // in allocation order (references stay valid across push_back / pop_front)
  int32_t *error_flag_dev = nullptr;  // 16 words the kernels raise when they meet something the packer should have refused
  int32_t *error_host = nullptr;      // pinned: where hevcrecon_synchronize reads them
  int longest_intra_wait = 0;         // polls of the longest wait of an intra CTU for a neighbour seen so far (word 8)
  std::vector<void *> retired;  // device buffers that were outgrown: freed when the context is idle
  size_t largest_syntax = 0;    // device bytes of the largest packed picture so far
  void *dl_host = nullptr;  // pinned staging for downloads / uploads
  void *out_dev = nullptr;  // a picture in file format (hevcrecon_picture_download_file)
  // hevcrecon_picture_md5_submit: a stream of its own beside the reconstruction kernels, tickets in flight
  // (a hash kernel runs for milliseconds whatever it hashes: requests go round a few streams so that several overlap)
  hipStream_t md5_streams[4] = {};
  bool md5_stream = false;  // do they exist yet?
  hipEvent_t md5_fork = nullptr;  // recorded on the kernels' stream, waited for by the hash streams
  Md5Slot md5_slots[kMd5Slots];
  int64_t md5_next = 0;
  size_t out_cap = 0;
  size_t dl_cap = 0;
  bool profiling = false;
  std::vector<Interval> intervals;
  std::vector<hipEvent_t> event_pool;
  double ms[K_COUNT];
  int64_t launches[K_COUNT];
  std::string err;
};

struct hevcrecon_syntax {
  hevcrecon_ctx *ctx = nullptr;  // NULL for a detached (host-only) object
  PackedPicture pk;
  SeqInfo si;
  char *dev = nullptr;
  size_t dev_cap = 0;
  char *host = nullptr;  // pinned staging of its own (when the context's upload ring had no room)
  char *up_host = nullptr, *up_dev = nullptr;  // where the packed picture is staged and where its device copy lies
  hevcrecon_ctx::RingEntry *up_entry = nullptr;  // its space in the upload ring, if it got any
  bool dirty = false;    // the staging buffer holds a picture the device copy does not have yet
  size_t host_cap = 0;
  size_t upload_bytes = 0, off_runs = 0, off_runoff = 0;
  bool expanded = false;       // rec4 has been expanded from the uploaded runs
  int64_t last_batch = -1;     // last batch whose kernels read the device copy
  int64_t upload_seq = -1;     // number of the batch of copies that carried the last upload
  size_t off_rec4 = 0, off_tu[4] = {0, 0, 0, 0}, off_coeff = 0, off_intra = 0, off_ictu = 0, off_lvlc = 0, off_wave = 0, off_sao = 0, off_ctui = 0, off_mct[3] = {0, 0, 0}, off_mcb = 0, off_mcb8 = 0, off_scal = 0, off_wp = 0;
  bool uploaded = false;
  std::string err;
};

namespace {

int fail(hevcrecon_ctx *c, const std::string &m) {
  if (c) c->err = m;
  return 1;
}

// Space in the upload ring (bytes is a multiple of 256), or NULL when there is none.
hevcrecon_ctx::RingEntry *ring_alloc(hevcrecon_ctx *c, size_t bytes) {
  if (c->up_failed || c->up_bytes == 0 || bytes == 0 || bytes > c->up_bytes) return nullptr;
  if (!c->up_host) {
    if (hipHostMalloc((void **)&c->up_host, c->up_bytes) != hipSuccess) c->up_host = nullptr;
    if (c->up_host && hipMalloc((void **)&c->up_dev, c->up_bytes) != hipSuccess) {
      hipHostFree(c->up_host);
      c->up_host = nullptr;
    }
    if (!c->up_host) {
      (void)hipGetLastError();
      c->up_failed = true;
      return nullptr;
    }
  }
  while (!c->up_entries.empty() && !c->up_entries.front().live) c->up_entries.pop_front();
  size_t off;
  if (c->up_entries.empty()) {
    off = 0;
  } else {
    const size_t tail = c->up_entries.front().off, head = c->up_head;
    if (head > tail) {  // free: [head, end) and [0, tail)
      if (bytes <= c->up_bytes - head) off = head;
      else if (bytes <= tail) off = 0;
      else return nullptr;
    } else {  // wrapped: free is [head, tail), nothing when they meet
      if (bytes <= tail - head) off = head;
      else return nullptr;
    }
  }
  c->up_entries.push_back(hevcrecon_ctx::RingEntry{off, bytes, true});
  c->up_head = off + bytes;
  return &c->up_entries.back();
}

void ring_release(hevcrecon_syntax *s) {
  if (s->up_entry) s->up_entry->live = false;
  s->up_entry = nullptr;
}
#define HIPCHK(c, call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(c, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

SeqInfo make_seq_info(const hevcrecon_seq_params &sp) {
  SeqInfo s;
  s.w = sp.width;
  s.h = sp.height;
  s.cw = sp.width / 2;
  s.ch = sp.height / 2;
  s.uw = sp.width / 4;
  s.uh = sp.height / 4;
  s.ctus_w = (sp.width + 63) / 64;
  s.ctus_h = (sp.height + 63) / 64;
  s.bd_y = sp.bit_depth_luma;
  s.bd_c = sp.bit_depth_chroma;
  s.strong_intra = sp.strong_intra_smoothing;
  s.stride_y = (sp.width + 127) & ~127;
  s.stride_c = (sp.width / 2 + 127) & ~127;
  s.pcm_bd_y = sp.pcm_bit_depth_luma;
  s.pcm_bd_c = sp.pcm_bit_depth_chroma;
  s.pcm_lf_disable = sp.pcm_loop_filter_disable;
  return s;
}

bool seq_params_ok(const hevcrecon_seq_params &sp, std::string &why) {
  if (sp.ctu_size != 64 || (sp.ctu_size >> sp.max_cu_depth) != 4) {
    why = "only 64x64 CTUs with 4x4 minimum partitions are supported";
    return false;
  }
  if (sp.width <= 0 || sp.height <= 0 || (sp.width & 7) || (sp.height & 7) || sp.width > 8192 || sp.height > 8192) {
    why = "picture size must be a multiple of 8 and at most 8192";
    return false;
  }
  if (sp.bit_depth_luma < 8 || sp.bit_depth_luma > 12 || sp.bit_depth_chroma < 8 || sp.bit_depth_chroma > 12 ||
      (sp.bit_depth_luma > 8) != (sp.bit_depth_chroma > 8)) {
    why = "bit depths must be 8..12 and on the same side of 8 for luma and chroma";
    return false;
  }
  if (sp.pcm_bit_depth_luma < 0 || sp.pcm_bit_depth_luma > sp.bit_depth_luma || sp.pcm_bit_depth_chroma < 0 ||
      sp.pcm_bit_depth_chroma > sp.bit_depth_chroma) {
    why = "PCM bit depths must not exceed the sample bit depths";
    return false;
  }
  return true;
}

int alloc_picture(hevcrecon_ctx *c, DevPicture &p) {
  if (p.base) return 0;
  HIPCHK(c, hipMalloc(&p.base, c->pic_bytes));
  for (int i = 0; i < 3; i++) p.p[i] = (char *)p.base + c->plane_off[i];
  return 0;
}
void free_picture(DevPicture &p) {
  if (p.base) hipFree(p.base);
  p = DevPicture();
}

hipEvent_t get_event(hevcrecon_ctx *c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;  // the interval is dropped (Timed checks), the launches go ahead untimed
  }
  return e;
}

struct Timed {
  hevcrecon_ctx *c;
  int kind;
  hipEvent_t a = nullptr;
  Timed(hevcrecon_ctx *c_, int kind_, int nlaunch, bool enabled = true) : c(c_), kind(kind_) {
    c->launches[kind] += nlaunch;
    if (c->profiling && enabled) {
      a = get_event(c);
      if (a) hipEventRecord(a, c->stream);
    }
  }
  ~Timed() {
    if (a) {
      hipEvent_t b = get_event(c);
      if (b) {
        hipEventRecord(b, c->stream);
        c->intervals.push_back(Interval{kind, a, b});
      } else {
        c->event_pool.push_back(a);
      }
    }
  }
};

void drain_intervals(hevcrecon_ctx *c) {
  for (size_t i = 0; i < c->intervals.size(); i++) {
    Interval &iv = c->intervals[i];
    float ms = 0;
    hipEventSynchronize(iv.b);
    hipEventElapsedTime(&ms, iv.a, iv.b);
    c->ms[iv.kind] += ms;
    c->event_pool.push_back(iv.a);
    c->event_pool.push_back(iv.b);
  }
  c->intervals.clear();
}

int ensure_staging(hevcrecon_ctx *c, size_t n) {
  if (c->dl_cap >= n) return 0;
  if (c->dl_host) hipHostFree(c->dl_host);
  c->dl_host = nullptr;
  c->dl_cap = 0;
  HIPCHK(c, hipHostMalloc(&c->dl_host, n));
  c->dl_cap = n;
  return 0;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The kernels raise bits in the context's error words instead of indexing out of range when they meet work lists
// the packer should never produce (and when a dependency wait runs out).  Read after the stream has drained.
int check_device_errors(hevcrecon_ctx *c) {
  HIPCHK(c, hipMemcpyAsync(c->error_host, c->error_flag_dev, 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int bits = c->error_host[0];
  c->longest_intra_wait = std::max(c->longest_intra_wait, c->error_host[8]);
  if (!bits) return 0;
  HIPCHK(c, hipMemsetAsync(c->error_flag_dev, 0, 64, c->stream));
  std::string m = "the device rejected a work list:";
  if (bits & HR_ERR_EXPAND) m += " motion record runs do not cover a CTU;";
  if (bits & HR_ERR_INTRA_CAPACITY) m += " a CTU holds more intra blocks or levels than the kernel's tables;";
  if (bits & HR_ERR_INTRA_TIMEOUT) {
    m += " an intra CTU waited for a neighbour that never finished";
    char buf[160];
    std::snprintf(buf, sizeof(buf), " (entry %d = picture %d CTU table index %d at (%d,%d), dependencies %d, neighbour %d; %d of %d entries drawn);",
                  c->error_host[2], c->error_host[3] >> 16, c->error_host[3] & 0xffff, c->error_host[7] & 255, (c->error_host[7] >> 8) & 255,
                  c->error_host[7] >> 16, c->error_host[4], c->error_host[5], c->error_host[6]);
    m += buf;
  }
  return fail(c, m);
}

// Copy device planes of `p` into int16 host planes.
int download_planes(hevcrecon_ctx *c, const DevPicture &p, int16_t *y, int16_t *cb, int16_t *cr, int stride_y, int stride_c) {
  const SeqInfo &s = c->si;
  const size_t pb = c->pel_bytes;
  const size_t ny = (size_t)s.w * s.h * pb, nc = (size_t)s.cw * s.ch * pb;
  if (ensure_staging(c, ny + 2 * nc)) return 1;
  char *h = (char *)c->dl_host;
  HIPCHK(c, hipMemcpy2DAsync(h, s.w * pb, p.p[0], s.stride_y * pb, s.w * pb, s.h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(h + ny, s.cw * pb, p.p[1], s.stride_c * pb, s.cw * pb, s.ch, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(h + ny + nc, s.cw * pb, p.p[2], s.stride_c * pb, s.cw * pb, s.ch, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (check_device_errors(c)) return 1;
  int16_t *dst[3] = {y, cb, cr};
  const char *src[3] = {h, h + ny, h + ny + nc};
  for (int k = 0; k < 3; k++) {
    const int w = k ? s.cw : s.w, hh = k ? s.ch : s.h, st = k ? stride_c : stride_y;
    for (int r = 0; r < hh; r++) {
      int16_t *d = dst[k] + (size_t)r * st;
      if (pb == 1) {
        const uint8_t *q = (const uint8_t *)src[k] + (size_t)r * w;
        for (int i = 0; i < w; i++) d[i] = q[i];
      } else {
        const uint16_t *q = (const uint16_t *)src[k] + (size_t)r * w;
        for (int i = 0; i < w; i++) d[i] = (int16_t)q[i];
      }
    }
  }
  return 0;
}

// Frees whatever a context holds; also the way out of a hevcrecon_create that failed half-way (every member is
// either null or valid).
void destroy_context(hevcrecon_ctx *c) {
  hipSetDevice(c->device);
  if (std::getenv("HEVCRECON_DEBUG")) std::fprintf(stderr, "hevcrecon: longest wait of an intra CTU for a neighbour: %d polls\n", c->longest_intra_wait);
#ifdef HEVCRECON_INTRA_STAMPS
  hr_intra_stamps_report();
#endif
  // nothing may still be using the buffers freed below: the kernels' stream first, then the copy streams the
  // contexts of this device share (they read the pinned upload ring and write its device mirror)
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->have_copy_streams)
    for (int i = 0; i < kCopyStreams; i++) hipStreamSynchronize(c->copy_stream[i]);
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) hipStreamSynchronize(c->md5_streams[i]);
  {  // the fixed work area has no user any more once this context's stream has drained
    std::lock_guard<std::mutex> lock(g_copy_mutex);
    if (c->device >= 0 && c->device < 64 && g_scratch_user[c->device].ctx == c) g_scratch_user[c->device] = ScratchUser{nullptr, nullptr};
  }
  for (size_t i = 0; i < c->retired.size(); i++) hipFree(c->retired[i]);
  c->retired.clear();
  for (int i = 0; i < kMd5Slots; i++) {
    Md5Slot &m = c->md5_slots[i];
    if (m.dev) hipFree(m.dev);
    if (m.snap) hipFree(m.snap);
    if (m.host) hipHostFree(m.host);
    if (m.done) hipEventDestroy(m.done);
  }
  if (c->md5_fork) hipEventDestroy(c->md5_fork);
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) hipStreamDestroy(c->md5_streams[i]);
  drain_intervals(c);
  for (size_t i = 0; i < c->pics.size(); i++) free_picture(c->pics[i]);
  for (size_t i = 0; i < c->work.size(); i++) free_picture(c->work[i]);
  for (size_t i = 0; i < c->work_resi.size(); i++) {
    if (c->work_resi[i]) hipFree(c->work_resi[i]);
    c->work_resi[i] = nullptr;
  }
  free_picture(c->stage[0]);
  free_picture(c->stage[1]);
  for (size_t i = 0; i < c->event_pool.size(); i++) hipEventDestroy(c->event_pool[i]);
  for (int i = 0; i < kRing; i++)
    if (c->ring_ev[i]) hipEventDestroy(c->ring_ev[i]);
  for (int i = 0; i < kRing; i++)
    if (c->batch_ev[i]) hipEventDestroy(c->batch_ev[i]);
  for (int i = 0; i < kRing * kCopyStreams; i++)
    if (c->copy_ev[i]) hipEventDestroy(c->copy_ev[i]);
  if (c->jobs_host) hipHostFree(c->jobs_host);
  if (c->jobs_dev) hipFree(c->jobs_dev);
  if (c->sync_dev) hipFree(c->sync_dev);
  if (c->list_host) hipHostFree(c->list_host);
  if (c->list_dev) hipFree(c->list_dev);
  if (c->error_flag_dev) hipFree(c->error_flag_dev);
  if (c->error_host) hipHostFree(c->error_host);
  if (c->dl_host) hipHostFree(c->dl_host);
  if (c->out_dev) hipFree(c->out_dev);
  if (c->up_host) hipHostFree(c->up_host);
  if (c->up_dev) hipFree(c->up_dev);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->have_copy_streams) release_copy_streams(c->device);
  (void)hipGetLastError();
  delete c;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int hevcrecon_create(hevcrecon_ctx **out, int device) {
  if (!out) return 1;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return 2;  // no GPU: no engine
  if (hipSetDevice(device) != hipSuccess) return 2;
  hevcrecon_ctx *c = new hevcrecon_ctx();
  c->device = device;
  std::memset(&c->sp, 0, sizeof(c->sp));
  std::memset(c->ms, 0, sizeof(c->ms));
  std::memset(c->launches, 0, sizeof(c->launches));
  std::memset(c->ring_used, 0, sizeof(c->ring_used));
  bool ok = hipStreamCreate(&c->stream) == hipSuccess;
  ok = ok && lookup_scratch(device);
  ok = ok && acquire_copy_streams(device, c->copy_stream);
  c->have_copy_streams = ok;
  ok = ok && hipHostMalloc((void **)&c->jobs_host, sizeof(PicJob) * HR_MAX_BATCH * kRing) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->jobs_dev, sizeof(PicJob) * HR_MAX_BATCH * kRing) == hipSuccess;

  ok = ok && hipMalloc((void **)&c->error_flag_dev, 64) == hipSuccess;
  ok = ok && hipHostMalloc((void **)&c->error_host, 64) == hipSuccess;
  for (int i = 0; ok && i < kRing; i++) ok = hipEventCreate(&c->ring_ev[i]) == hipSuccess;
  for (int i = 0; ok && i < kRing; i++) ok = hipEventCreateWithFlags(&c->batch_ev[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < kRing * kCopyStreams; i++) ok = hipEventCreateWithFlags(&c->copy_ev[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMemsetAsync(c->error_flag_dev, 0, 64, c->stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    destroy_context(c);
    return 3;
  }
  *out = c;
  return 0;
}

extern "C" void hevcrecon_destroy(hevcrecon_ctx *c) {
  if (!c) return;
  destroy_context(c);
}

extern "C" int hevcrecon_set_upload_ring(hevcrecon_ctx *c, int64_t bytes) {
  if (!c) return 1;
  if (c->up_host) return fail(c, "the upload ring is already in use");
  if (bytes < 0 || bytes > ((int64_t)64 << 30)) return fail(c, "upload ring size out of range");
  c->up_bytes = align_up((size_t)bytes, 256);
  c->up_failed = false;
  return 0;
}

extern "C" const char *hevcrecon_last_error(const hevcrecon_ctx *c) { return c ? c->err.c_str() : "no context (is a GPU present?)"; }

extern "C" int hevcrecon_sequence_init(hevcrecon_ctx *c, const hevcrecon_seq_params *sp) {
  if (!c || !sp) return 1;
  if (c->have_seq && std::memcmp(&c->sp, sp, sizeof(*sp)) == 0) return 0;  // called once per slice by the decoder
  std::string why;
  if (!seq_params_ok(*sp, why)) return fail(c, why);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamSynchronize(c->md5_streams[i]));
  for (size_t i = 0; i < c->pics.size(); i++) free_picture(c->pics[i]);
  for (size_t i = 0; i < c->work.size(); i++) free_picture(c->work[i]);
  for (size_t i = 0; i < c->work_resi.size(); i++) {
    if (c->work_resi[i]) hipFree(c->work_resi[i]);
    c->work_resi[i] = nullptr;
  }
  free_picture(c->stage[0]);
  free_picture(c->stage[1]);
  c->sp = *sp;
  c->si = make_seq_info(*sp);
  c->pel_bytes = sp->bit_depth_luma > 8 ? 2 : 1;
  const size_t pb = c->pel_bytes;
  c->plane_off[0] = 0;
  c->plane_off[1] = align_up((size_t)c->si.stride_y * c->si.h * pb, 256);
  c->plane_off[2] = c->plane_off[1] + align_up((size_t)c->si.stride_c * c->si.ch * pb, 256);
  // + 256: the packed reference fetch of k_inter reads whole dwords and may run a few bytes past the last row
  c->pic_bytes = c->plane_off[2] + align_up((size_t)c->si.stride_c * c->si.ch * pb, 256) + 256;
  c->resi_off[0] = 0;
  c->resi_off[1] = align_up((size_t)c->si.stride_y * c->si.h * 2, 256);
  c->resi_off[2] = c->resi_off[1] + align_up((size_t)c->si.stride_c * c->si.ch * 2, 256);
  c->resi_bytes = c->resi_off[2] + align_up((size_t)c->si.stride_c * c->si.ch * 2, 256) + 256;
  c->have_seq = true;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// syntax objects
// ---------------------------------------------------------------------------------------------
extern "C" int hevcrecon_syntax_create(hevcrecon_ctx *c, hevcrecon_syntax **out) {
  if (!out) return 1;
  *out = nullptr;
  if (!c) return 1;
  if (!c->have_seq) return fail(c, "hevcrecon_sequence_init must come first");
  hevcrecon_syntax *s = new hevcrecon_syntax();
  s->ctx = c;
  s->si = c->si;
  *out = s;
  return 0;
}

extern "C" int hevcrecon_syntax_create_detached(const hevcrecon_seq_params *sp, hevcrecon_syntax **out) {
  if (!out || !sp) return 1;
  *out = nullptr;
  std::string why;
  if (!seq_params_ok(*sp, why)) return 1;
  hevcrecon_syntax *s = new hevcrecon_syntax();
  s->si = make_seq_info(*sp);
  *out = s;
  return 0;
}

extern "C" void hevcrecon_syntax_destroy(hevcrecon_syntax *s) {
  if (!s) return;
  if (s->ctx) {
    hipSetDevice(s->ctx->device);
    for (int k = 0; k < kCopyStreams; k++) hipStreamSynchronize(s->ctx->copy_stream[k]);
    hipStreamSynchronize(s->ctx->stream);
  }
  ring_release(s);
  if (s->dev) hipFree(s->dev);
  if (s->host) hipHostFree(s->host);
  delete s;
}

extern "C" int hevcrecon_syntax_begin(hevcrecon_syntax *s) {
  if (!s) return 1;
  if (s->ctx) {
    if (!s->ctx->have_seq) return fail(s->ctx, "no sequence parameters");
    s->si = s->ctx->si;
    // the previous upload must have left the staging buffer, and the kernels that read the previous picture's
    // device copy must be done, before either is rewritten
    // (copies are queued with the first batch that uses the picture and that batch's kernels wait for them: once
    // the last batch that read the picture is done, so is every copy out of its staging buffer)
    if (s->uploaded && s->last_batch >= 0) hipEventSynchronize(s->ctx->batch_ev[s->last_batch % kRing]);
  }
  ring_release(s);  // nothing reads the previous picture any more (waited for above)
  s->uploaded = false;
  s->dirty = false;
  s->last_batch = -1;
  s->pk.begin(s->si);
  return 0;
}

extern "C" int hevcrecon_syntax_set_scaling_lists(hevcrecon_syntax *s, const hevcrecon_scaling_lists *l) {
  if (!s) return 1;
  static_assert(sizeof(ScalingTables) == sizeof(hevcrecon_scaling_lists), "scaling list layouts differ");
  if (s->pk.finished) return fail(s->ctx, "scaling lists must be set between hevcrecon_syntax_begin and hevcrecon_syntax_finish");
  s->pk.use_scaling = l != nullptr;
  if (l) std::memcpy(&s->pk.scaling, l, sizeof(ScalingTables));
  return 0;
}

extern "C" int hevcrecon_syntax_set_pred_weights(hevcrecon_syntax *s, const hevcrecon_pred_weights *w) {
  if (!s) return 1;
  static_assert(sizeof(WpTables) == sizeof(hevcrecon_pred_weights), "weight table layouts differ");
  if (s->pk.finished) return fail(s->ctx, "prediction weights must be set between hevcrecon_syntax_begin and hevcrecon_syntax_finish");
  if (w) {
    // what the kernels' 24-bit multiplies and 32-bit sums assume (the syntax allows far less: denominators 0..7,
    // weights -128..255, offsets -128..127)
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < HR_MAX_REFS; r++)
        for (int c = 0; c < 3; c++)
          if (w->log2_denom[l][r][c] < 0 || w->log2_denom[l][r][c] > 7 || w->weight[l][r][c] < -256 || w->weight[l][r][c] > 255 ||
              w->offset[l][r][c] < -128 || w->offset[l][r][c] > 127)
            return fail(s->ctx, "prediction weight, offset or denominator out of range");
    std::memcpy(&s->pk.wp, w, sizeof(WpTables));
  }
  s->pk.use_wp = w != nullptr;
  return 0;
}

extern "C" int hevcrecon_syntax_set_slices(hevcrecon_syntax *s, int n, const hevcrecon_slice_params *slices) {
  if (!s) return 1;
  if (!s->pk.set_slices(n, slices)) {
    s->err = s->pk.error;
    s->pk.error.clear();  // a refused table leaves the picture as it was
    return fail(s->ctx, s->err);
  }
  return 0;
}

extern "C" int hevcrecon_syntax_add_ctu(hevcrecon_syntax *s, int ctu_addr, const hevcrecon_ctu_syntax *ctu) {
  if (!s || !ctu) return 1;
  if (!s->pk.add_ctu(ctu_addr, ctu)) {
    s->err = s->pk.error;
    return fail(s->ctx, s->pk.error);
  }
  return 0;
}

// The packed picture is ready in its pinned staging buffer.  It crosses the host link when the first batch that
// uses it is submitted: the copies of all pictures of the batch that need one go out together on the copy stream(s)
// (beside the previous batch's kernels, which the host has queued ahead), then k_expand expands the
// run-length coded records.
static int queue_upload(hevcrecon_syntax *s) {
  s->dirty = true;
  s->uploaded = true;
  s->expanded = false;
  return 0;
}

extern "C" const char *hevcrecon_syntax_last_error(const hevcrecon_syntax *s) { return s ? s->err.c_str() : "no syntax object"; }

// The packed picture in s->pk goes into pinned staging (the context's upload ring, or buffers of the object's own) and
// is marked for its trip to the device.
static int stage_packed(hevcrecon_syntax *s) {
  hevcrecon_ctx *c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  PackedPicture &k = s->pk;
  size_t off = 0;
  auto place = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  s->off_runs = place(k.runs.size() * sizeof(Part4));
  s->off_runoff = place(k.ctu_run_off.size() * sizeof(uint32_t));
  for (int i = 0; i < 4; i++) s->off_tu[i] = place(k.tu[i].size() * sizeof(TuItem));
  s->off_coeff = place(k.coeff.size() * sizeof(uint32_t));
  s->off_intra = place(k.intra.size() * sizeof(IntraItem));
  s->off_ictu = place(k.ictu.size() * sizeof(IntraCtu));
  s->off_lvlc = place(k.rounds.size() * sizeof(uint16_t));
  s->off_wave = place(k.wave_off.size() * sizeof(int32_t));
  s->off_sao = place(k.sao.size() * sizeof(SaoCtu));
  s->off_ctui = place(k.ctu_info.size() * sizeof(CtuInfo));
  for (int i = 0; i < 3; i++) s->off_mct[i] = place(k.mc_tile[i].size() * sizeof(McTile));
  s->off_mcb = place(k.mc_blk.size() * sizeof(McBlk));
  s->off_mcb8 = place(k.mc_blk8.size() * sizeof(McBlk));
  s->off_scal = place(sizeof(ScalingTables));
  s->off_wp = place(sizeof(WpTables));
  const size_t upload_bytes = off;
  s->upload_bytes = upload_bytes;
  s->off_rec4 = place((size_t)k.n_rec4 * sizeof(Part4));  // device only: expanded from the runs after the copy
  const size_t total = off;
  if (s->dev_cap < total) {
    // A larger buffer, sized by the largest picture the context has seen (intra pictures carry several times the lists
    // of the others, and every object gets one sooner or later).  The old one is not freed here: kernels in flight may
    // still read it, and freeing waits for the whole device, hash kernels included; it is freed when the context is
    // idle anyway (hevcrecon_synchronize, hevcrecon_destroy).
    if (s->dev) c->retired.push_back(s->dev);
    s->dev = nullptr;
    s->dev_cap = 0;
    c->largest_syntax = std::max(c->largest_syntax, total);
    const size_t cap = c->largest_syntax + c->largest_syntax / 4;
    HIPCHK(c, hipMalloc((void **)&s->dev, cap));
    s->dev_cap = cap;
  }
  ring_release(s);
  s->up_entry = ring_alloc(c, upload_bytes);
  if (s->up_entry) {
    s->up_host = c->up_host + s->up_entry->off;
    s->up_dev = c->up_dev + s->up_entry->off;
  } else {
    if (s->host_cap < upload_bytes) {
      if (s->host) hipHostFree(s->host);
      s->host = nullptr;
      s->host_cap = 0;
      const size_t cap = upload_bytes + upload_bytes / 4;
      HIPCHK(c, hipHostMalloc((void **)&s->host, cap));
      s->host_cap = cap;
    }
    s->up_host = s->host;
    s->up_dev = s->dev;
  }
  // (an empty vector's data() may be null, which memcpy must not be given even for zero bytes)
  auto put = [&](size_t off, const auto &v) {
    if (!v.empty()) std::memcpy(s->up_host + off, v.data(), v.size() * sizeof(v[0]));
  };
  put(s->off_runs, k.runs);
  put(s->off_runoff, k.ctu_run_off);
  for (int i = 0; i < 4; i++) put(s->off_tu[i], k.tu[i]);
  put(s->off_coeff, k.coeff);
  put(s->off_intra, k.intra);
  put(s->off_ictu, k.ictu);
  put(s->off_lvlc, k.rounds);
  put(s->off_wave, k.wave_off);
  put(s->off_sao, k.sao);
  put(s->off_ctui, k.ctu_info);
  for (int i = 0; i < 3; i++) put(s->off_mct[i], k.mc_tile[i]);
  put(s->off_mcb, k.mc_blk);
  put(s->off_mcb8, k.mc_blk8);
  std::memcpy(s->up_host + s->off_scal, &k.scaling, sizeof(ScalingTables));
  std::memcpy(s->up_host + s->off_wp, &k.wp, sizeof(WpTables));
  return queue_upload(s);
}

extern "C" int hevcrecon_syntax_finish(hevcrecon_syntax *s, const hevcrecon_pic_params *pp) {
  if (!s || !pp) return 1;
  if (!s->pk.finish(pp)) {
    s->err = s->pk.error;
    return fail(s->ctx, s->pk.error);
  }
  if (!s->ctx) return 0;  // detached: host-side packing only
  return stage_packed(s);
}

// ---- a packed picture as one block of bytes -----------------------------------------------------
// What hevcrecon_syntax_finish leaves in a syntax object, flattened so that it can cross a process boundary: a
// decoder process that never opens the GPU packs into a host-only object and exports; the process that owns the GPU
// imports into an object of its context and reconstructs.  Position independent, checked on import.
namespace {
const uint32_t kBlobMagic = 0x42525648;  // "HVRB"
const uint32_t kBlobVersion = 2;
enum { BL_RUNS, BL_RUNOFF, BL_TU0, BL_TU1, BL_TU2, BL_TU3, BL_COEFF, BL_INTRA, BL_ICTU, BL_ROUNDS, BL_WAVE, BL_SAO, BL_CTUI, BL_MCT0, BL_MCT1, BL_MCT2, BL_MCB, BL_MCB8, BL_SLICES, BL_COUNT };
struct BlobHeader {
  uint32_t magic, version;
  uint64_t total_bytes;
  SeqInfo seq;
  hevcrecon_pic_params pp;
  int64_t coded_coeffs, mc_ref_samples, n_rec4, n_rec4_intra;
  int32_t max_ref_idx[2];
  int32_t inter_without_ref, has_bypass, has_pcm, use_scaling, use_wp, pad;
  uint64_t count[BL_COUNT];  // elements of every array, in the order of the BL_ indices; the arrays follow the tables, each padded to 16 bytes
  ScalingTables scaling;
  WpTables wp;
};
const size_t kBlobElem[BL_COUNT] = {sizeof(Part4), 4, sizeof(TuItem), sizeof(TuItem), sizeof(TuItem), sizeof(TuItem), 4, sizeof(IntraItem), sizeof(IntraCtu), 2, 4,
                                    sizeof(SaoCtu), sizeof(CtuInfo), sizeof(McTile), sizeof(McTile), sizeof(McTile), sizeof(McBlk), sizeof(McBlk), sizeof(hevcrecon_slice_params)};
size_t blob_header_bytes() { return align_up(sizeof(BlobHeader), 16); }
template <class T> const void *vec_data(const std::vector<T> &v) { return v.data(); }
}  // namespace

extern "C" int hevcrecon_syntax_export(const hevcrecon_syntax *s, void *dst, int64_t cap, int64_t *bytes) {
  if (!s || !bytes) return 1;
  const PackedPicture &k = s->pk;
  if (!k.finished) return 1;
  const void *src[BL_COUNT] = {vec_data(k.runs), vec_data(k.ctu_run_off), vec_data(k.tu[0]), vec_data(k.tu[1]), vec_data(k.tu[2]), vec_data(k.tu[3]),
                               vec_data(k.coeff), vec_data(k.intra), vec_data(k.ictu), vec_data(k.rounds), vec_data(k.wave_off), vec_data(k.sao),
                               vec_data(k.ctu_info), vec_data(k.mc_tile[0]), vec_data(k.mc_tile[1]), vec_data(k.mc_tile[2]), vec_data(k.mc_blk), vec_data(k.mc_blk8),
                               vec_data(k.slices)};
  const size_t cnt[BL_COUNT] = {k.runs.size(), k.ctu_run_off.size(), k.tu[0].size(), k.tu[1].size(), k.tu[2].size(), k.tu[3].size(), k.coeff.size(),
                                k.intra.size(), k.ictu.size(), k.rounds.size(), k.wave_off.size(), k.sao.size(), k.ctu_info.size(), k.mc_tile[0].size(),
                                k.mc_tile[1].size(), k.mc_tile[2].size(), k.mc_blk.size(), k.mc_blk8.size(), k.slices.size()};
  size_t total = blob_header_bytes();
  for (int i = 0; i < BL_COUNT; i++) total += align_up(cnt[i] * kBlobElem[i], 16);
  *bytes = (int64_t)total;
  if (!dst) return 0;  // size query
  if (cap < (int64_t)total) return 1;
  BlobHeader h;
  std::memset(&h, 0, sizeof(h));
  h.magic = kBlobMagic;
  h.version = kBlobVersion;
  h.total_bytes = total;
  h.seq = k.seq;
  h.pp = k.pp;
  h.coded_coeffs = k.coded_coeffs;
  h.mc_ref_samples = k.mc_ref_samples;
  h.n_rec4 = k.n_rec4;
  h.n_rec4_intra = k.n_rec4_intra;
  h.max_ref_idx[0] = k.max_ref_idx[0];
  h.max_ref_idx[1] = k.max_ref_idx[1];
  h.inter_without_ref = k.inter_without_ref;
  h.has_bypass = k.has_bypass;
  h.has_pcm = k.has_pcm;
  h.use_scaling = k.use_scaling;
  h.use_wp = k.use_wp;
  for (int i = 0; i < BL_COUNT; i++) h.count[i] = cnt[i];
  if (k.use_scaling) h.scaling = k.scaling;
  if (k.use_wp) h.wp = k.wp;
  char *out = (char *)dst;
  std::memcpy(out, &h, sizeof(h));
  size_t off = blob_header_bytes();
  for (int i = 0; i < BL_COUNT; i++) {
    if (cnt[i]) std::memcpy(out + off, src[i], cnt[i] * kBlobElem[i]);
    off += align_up(cnt[i] * kBlobElem[i], 16);
  }
  return 0;
}

extern "C" int hevcrecon_syntax_import(hevcrecon_syntax *s, const void *src, int64_t bytes) {
  if (!s || !src) return 1;
  auto refuse = [&](const char *why) {
    s->err = why;
    return fail(s->ctx, s->err);
  };
  if (bytes < (int64_t)blob_header_bytes()) return refuse("packed picture: too short");
  BlobHeader h;
  std::memcpy(&h, src, sizeof(h));
  if (h.magic != kBlobMagic || h.version != kBlobVersion || h.total_bytes != (uint64_t)bytes) return refuse("packed picture: not one (magic, version or length)");
  size_t total = blob_header_bytes();
  for (int i = 0; i < BL_COUNT; i++) {
    if (h.count[i] > ((uint64_t)1 << 28)) return refuse("packed picture: array size out of range");
    total += align_up((size_t)h.count[i] * kBlobElem[i], 16);
  }
  if (total != (size_t)bytes) return refuse("packed picture: array sizes do not add up to its length");
  if (int rc = hevcrecon_syntax_begin(s)) return rc;  // waits until the GPU is done with the picture the object held
  if (std::memcmp(&h.seq, &s->si, sizeof(SeqInfo)) != 0) return refuse("packed picture: packed for different sequence parameters");
  const size_t nctu = (size_t)h.seq.ctus_w * h.seq.ctus_h;
  for (int l = 0; l < 2; l++)
    if (h.pp.num_ref[l] < 0 || h.pp.num_ref[l] > HR_MAX_REFS) return refuse("packed picture: bad reference list size");
  if (h.count[BL_RUNOFF] != nctu + 1 || h.count[BL_SAO] != nctu * 3 || (h.count[BL_CTUI] != 0 && h.count[BL_CTUI] != nctu) || h.count[BL_ICTU] > nctu)
    return refuse("packed picture: per-CTU tables of the wrong size");
  PackedPicture &k = s->pk;
  const char *in = (const char *)src + blob_header_bytes();
  auto take = [&](int i, auto &vec) {
    typedef typename std::remove_reference<decltype(vec)>::type V;
    vec.resize((size_t)h.count[i]);
    if (h.count[i]) std::memcpy((void *)vec.data(), in, (size_t)h.count[i] * sizeof(typename V::value_type));
    in += align_up((size_t)h.count[i] * kBlobElem[i], 16);
  };
  take(BL_RUNS, k.runs);
  take(BL_RUNOFF, k.ctu_run_off);
  take(BL_TU0, k.tu[0]);
  take(BL_TU1, k.tu[1]);
  take(BL_TU2, k.tu[2]);
  take(BL_TU3, k.tu[3]);
  take(BL_COEFF, k.coeff);
  take(BL_INTRA, k.intra);
  take(BL_ICTU, k.ictu);
  take(BL_ROUNDS, k.rounds);
  take(BL_WAVE, k.wave_off);
  take(BL_SAO, k.sao);
  take(BL_CTUI, k.ctu_info);
  take(BL_MCT0, k.mc_tile[0]);
  take(BL_MCT1, k.mc_tile[1]);
  take(BL_MCT2, k.mc_tile[2]);
  take(BL_MCB, k.mc_blk);
  take(BL_MCB8, k.mc_blk8);
  take(BL_SLICES, k.slices);
  // what the kernels index with has to lie inside the arrays (the exporting packer guarantees it; a blob is foreign input)
  const size_t n_runs = k.runs.size(), n_intra = k.intra.size(), n_rounds = k.rounds.size(), n_coeff = k.coeff.size();
  for (size_t a = 0; a + 1 < k.ctu_run_off.size(); a++)
    if (k.ctu_run_off[a] > k.ctu_run_off[a + 1] || k.ctu_run_off[a + 1] > n_runs) return refuse("packed picture: run table out of range");
  for (size_t i = 0; i < k.ictu.size(); i++) {
    const IntraCtu &ic = k.ictu[i];
    if ((size_t)ic.item_start + ic.n_items > n_intra || (size_t)ic.round_start + ic.n_rounds > n_rounds || ic.n_items > HR_INTRA_MAX_ITEMS ||
        ic.n_rounds > HR_INTRA_MAX_ROUNDS || ic.cx >= h.seq.ctus_w || ic.cy >= h.seq.ctus_h)
      return refuse("packed picture: intra CTU table out of range");
  }
  for (size_t w = 0; w + 1 < k.wave_off.size(); w++)
    if (k.wave_off[w] < 0 || k.wave_off[w] > k.wave_off[w + 1] || (size_t)k.wave_off[w + 1] > k.ictu.size()) return refuse("packed picture: intra wave table out of range");
  for (int t = 0; t < 4; t++)
    for (size_t i = 0; i < k.tu[t].size(); i++)
      if (k.tu[t][i].coeff_off >= n_coeff || k.tu[t][i].comp > 2) return refuse("packed picture: transform block out of range");
  for (int t = 0; t < 3; t++)
    for (size_t i = 0; i < k.mc_tile[t].size(); i++) {
      const McTile &m = k.mc_tile[t][i];
      if (m.x >= h.seq.w || m.y >= h.seq.h || m.ref[0] >= h.pp.num_ref[0] || m.ref[1] >= h.pp.num_ref[1] || (m.ref[0] < 0 && m.ref[1] < 0))
        return refuse("packed picture: motion region out of range");
    }
  for (size_t i = 0; i < k.mc_blk.size(); i++)
    if (k.mc_blk[i].x >= h.seq.w || k.mc_blk[i].y >= h.seq.h || ((k.mc_blk[i].x | k.mc_blk[i].y) & 7)) return refuse("packed picture: motion block out of range");
  for (size_t i = 0; i < k.mc_blk8.size(); i++)
    if (k.mc_blk8[i].x >= h.seq.w || k.mc_blk8[i].y >= h.seq.h || ((k.mc_blk8[i].x | k.mc_blk8[i].y) & 7) || h.seq.bd_y > 8 || h.seq.bd_c > 8)
      return refuse("packed picture: 8x8 motion block out of range");
  // reference indices as the records hold them (not as the header claims)
  int max_ref[2] = {-1, -1};
  bool no_ref = false;
  for (size_t i = 0; i < n_runs; i++) {
    const Part4 &r = k.runs[i];
    if ((size_t)r.pad + r.pad2[0] > 255) return refuse("packed picture: run table out of range");
    if (r.flags & P4_INTRA) continue;
    for (int l = 0; l < 2; l++) max_ref[l] = std::max<int>(max_ref[l], r.ref_idx[l]);
    no_ref = no_ref || (r.ref_idx[0] < 0 && r.ref_idx[1] < 0);
  }
  h.max_ref_idx[0] = max_ref[0];
  h.max_ref_idx[1] = max_ref[1];
  h.inter_without_ref = no_ref;
  k.release_work();
  k.seq = h.seq;
  k.pp = h.pp;
  k.coded_coeffs = h.coded_coeffs;
  k.mc_ref_samples = h.mc_ref_samples;
  k.n_rec4 = h.n_rec4;
  k.n_rec4_intra = h.n_rec4_intra;
  k.max_ref_idx[0] = h.max_ref_idx[0];
  k.max_ref_idx[1] = h.max_ref_idx[1];
  k.inter_without_ref = h.inter_without_ref != 0;
  k.has_bypass = h.has_bypass != 0;
  k.has_pcm = h.has_pcm != 0;
  k.use_scaling = h.use_scaling != 0;
  k.use_wp = h.use_wp != 0;
  k.scaling = h.scaling;
  k.wp = h.wp;
  k.finished = true;
  k.error.clear();
  if (!s->ctx) return 0;
  return stage_packed(s);
}

// No counterpart in the reference: marks an already packed picture for another trip to the device
// (hevcrecon_syntax_finish does it once).  For hosts that keep packed pictures and for measuring the hand-over.
extern "C" int hevcrecon_syntax_upload(hevcrecon_syntax *s) {
  if (!s) return 1;
  if (!s->ctx || !s->pk.finished || !s->dev) return fail(s->ctx, "nothing to upload: the picture is not finished on a context");
  return queue_upload(s);
}

extern "C" int hevcrecon_syntax_upload_batch(int n, hevcrecon_syntax *const *syn) {
  if (n < 0 || (n > 0 && !syn)) return 1;
  for (int i = 0; i < n; i++)
    if (int rc = hevcrecon_syntax_upload(syn[i])) return rc;
  return 0;
}

extern "C" int hevcrecon_syntax_stats(const hevcrecon_syntax *s, int64_t out[32]) {
  if (!s || !out) return 1;
  const PackedPicture &k = s->pk;
  for (int i = 0; i < 32; i++) out[i] = 0;
  for (int i = 0; i < 4; i++) out[i] = (int64_t)k.tu[i].size();
  // motion compensation lists: regions per tile size, mixed 16x16 regions, reference blocks read per tile size
  for (int t = 0; t < 3; t++) {
    out[16 + t] = (int64_t)k.mc_tile[t].size();
    const int ts = 64 >> t;
    for (size_t i = 0; i < k.mc_tile[t].size(); i++) {
      const McTile &m = k.mc_tile[t][i];
      const int nref = (m.ref[0] >= 0) + (m.ref[1] >= 0);
      const int64_t area = (int64_t)std::min<int>(ts, k.seq.w - m.x) * std::min<int>(ts, k.seq.h - m.y);
      out[20 + t] += nref;
      out[23 + t] += area;
      out[26 + t] += area * nref;
    }
  }
  out[19] = (int64_t)k.mc_blk.size();
  out[4] = k.coded_coeffs;
  out[31] = (int64_t)k.coeff.size() * 4;
  out[5] = (int64_t)k.intra.size();
  out[6] = k.wave_off.empty() ? 0 : (int64_t)k.wave_off.size() - 1;
  out[11] = (int64_t)k.ictu.size();
  out[12] = (int64_t)k.lvl_counts.size();
  out[7] = k.mc_ref_samples;
  out[8] = k.n_rec4;
  out[9] = k.n_rec4_intra;
  out[29] = (int64_t)s->upload_bytes;
  out[30] = (int64_t)k.runs.size();
  out[10] = k.finished ? 1 : 0;
  // intra schedule: rounds on the critical path (slowest CTU of every wave) and in total
  if (k.finished && !k.wave_off.empty()) {
    int64_t crit = 0, total = 0, crit_lvl = 0;
    for (size_t w = 0; w + 1 < k.wave_off.size(); w++) {
      int64_t worst = 0, worst_lvl = 0;
      for (int i = k.wave_off[w]; i < k.wave_off[w + 1]; i++) {
        const int64_t r = k.ictu[i].n_rounds;
        total += r;
        if (r > worst) worst = r;
        if (k.ictu[i].n_lvl > worst_lvl) worst_lvl = k.ictu[i].n_lvl;
      }
      crit += worst;
      crit_lvl += worst_lvl;
    }
    out[13] = crit;
    out[14] = total;
    out[15] = crit_lvl;
  }
  return 0;
}

extern "C" int64_t hevcrecon_syntax_algorithmic_bytes(const hevcrecon_syntax *s) { return s ? s->pk.algorithmic_bytes() : 0; }

// ---------------------------------------------------------------------------------------------
// reconstruction
// ---------------------------------------------------------------------------------------------
namespace {

// Device memory a batch works in besides the pictures it reads and writes: the job descriptors, per picture of the
// batch a working picture (prediction, reconstruction and deblocking happen there when SAO is on) and the residual
// planes of its intra blocks.  hevcrecon_reconstruct_batch takes them from the context's pools, which grow on
// demand; hevcrecon_reconstruct_noalloc from the library's fixed device array.
struct WorkSet {
  PicJob *jobs_host;  // pinned
  PicJob *jobs_dev;
  void *work_base[HR_MAX_BATCH];  // working pictures (planes at the context's plane_off)
  void *resi_base[HR_MAX_BATCH];  // residual planes (at the context's resi_off)
  uint32_t *sync_dev;             // intra kernel: ticket counter (word 0), "done" words of picture i from word kSyncHead + i * CTUs
  uint32_t *list_host, *list_dev; // intra kernel: the batch's CTU list (pinned / device); unused for a single picture
  int slot;                       // entry of the pinned job ring that jobs_host belongs to
  bool fixed;                     // the allocation-free entry point: no timers, no stage copies
};
const size_t kSyncHead = 64;  // words in front of the "done" words (the ticket counter has a 256-byte line of its own)

// Everything the kernels will index with is checked before anything is launched.
int validate_batch(hevcrecon_ctx *c, int n, hevcrecon_syntax *const *syn, const int *dst_ids) {
  const SeqInfo &s = c->si;
  for (int i = 0; i < n; i++) {
    hevcrecon_syntax *y = syn[i];
    if (!y || y->ctx != c || !y->uploaded || !y->pk.finished) return fail(c, "syntax object not finished on this context");
    if (std::memcmp(&y->si, &s, sizeof(s)) != 0) return fail(c, "syntax object packed for different sequence parameters");
    if (dst_ids[i] < 0 || dst_ids[i] >= kMaxPics) return fail(c, "picture id out of range");
    for (int j = 0; j < i; j++)
      if (dst_ids[j] == dst_ids[i]) return fail(c, "duplicate destination picture in batch");
    const hevcrecon_pic_params &pp = y->pk.pp;
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < pp.num_ref[l]; r++) {
        const int id = pp.ref_pic_id[l][r];
        if (id < 0 || id >= (int)c->pics.size() || !c->pics[id].valid) return fail(c, "reference picture is not resident");
        for (int j = 0; j < n; j++)
          if (dst_ids[j] == id) return fail(c, "a picture of the batch references another picture of the batch");
      }
    // every reference index stored in the motion records must exist in the lists (checked once, at pack time)
    if (y->pk.inter_without_ref || y->pk.max_ref_idx[0] >= pp.num_ref[0] || y->pk.max_ref_idx[1] >= pp.num_ref[1])
      return fail(c, "motion record refers to a reference index outside the lists");
  }
  return 0;
}

// Queues the hand-over of the stale pictures, the job descriptors and the kernels of one batch.  The caller has
// validated the batch and made sure every destination picture and everything `ws` names exists.  Book-keeping that
// later calls rely on (which pictures have been copied and expanded, the ring position) is committed only once
// everything has been queued: after a failure half-way the next call starts from the state before this one.
int reconstruct_core(hevcrecon_ctx *c, int n, hevcrecon_syntax *const *syn, const int *dst_ids, const WorkSet &ws) {
  const SeqInfo &s = c->si;
  PicJob *jh = ws.jobs_host, *jd = ws.jobs_dev;
  if (c->ring_used[ws.slot]) HIPCHK(c, hipEventSynchronize(c->ring_ev[ws.slot]));
  int max_tu[4] = {0, 0, 0, 0}, max_small_groups = 0, max_levels = 0, max_mct[3] = {0, 0, 0}, max_mcb = 0, max_mcb8 = 0;
  bool any_deblock = false, any_sao = false, any_wp = false, any_expand = false;
  for (int i = 0; i < n; i++) {
    hevcrecon_syntax *y = syn[i];
    const PackedPicture &k = y->pk;
    PicJob &j = jh[i];
    std::memset(&j, 0, sizeof(j));
    // Without SAO the deblocked picture IS the final picture: reconstruct straight into the DPB entry
    // (k_sao then has nothing to do for it).  With SAO the working picture is separate because SAO
    // classifies on deblocked samples only.
    const bool sao_on = k.pp.sao_luma || k.pp.sao_chroma;
    for (int p = 0; p < 3; p++) {
      j.out[p] = c->pics[dst_ids[i]].p[p];
      j.rec[p] = sao_on ? (void *)((char *)ws.work_base[i] + c->plane_off[p]) : j.out[p];
    }
    any_sao = any_sao || sao_on;
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < k.pp.num_ref[l]; r++) {
        const DevPicture &rp = c->pics[k.pp.ref_pic_id[l][r]];
        for (int p = 0; p < 3; p++) j.ref[l][r][p] = rp.p[p];
        j.ref_poc[l][r] = k.pp.ref_poc[l][r];
      }
    j.rec4 = (const Part4 *)(y->dev + y->off_rec4);
    j.runs = (const Part4 *)(y->up_dev + y->off_runs);
    j.ctu_run_off = (const uint32_t *)(y->up_dev + y->off_runoff);
    j.need_expand = y->expanded ? 0 : 1;
    any_expand = any_expand || !y->expanded;
    for (int t = 0; t < 3; t++) {
      j.mc_tile[t] = (const McTile *)(y->up_dev + y->off_mct[t]);
      j.n_mc_tile[t] = (int)k.mc_tile[t].size();
      max_mct[t] = std::max(max_mct[t], j.n_mc_tile[t]);
    }
    j.mc_blk = (const McBlk *)(y->up_dev + y->off_mcb);
    j.n_mc_blk = (int)k.mc_blk.size();
    max_mcb = std::max(max_mcb, j.n_mc_blk);
    j.mc_blk8 = (const McBlk *)(y->up_dev + y->off_mcb8);
    j.n_mc_blk8 = (int)k.mc_blk8.size();
    max_mcb8 = std::max(max_mcb8, j.n_mc_blk8);
    for (int t = 0; t < 4; t++) {
      j.tu[t] = (const TuItem *)(y->up_dev + y->off_tu[t]);
      j.n_tu[t] = (int)k.tu[t].size();
      if (j.n_tu[t] > max_tu[t]) max_tu[t] = j.n_tu[t];
    }
    // workgroups of the launch that the 16x16, 8x8 and 4x4 blocks share (16, 32 and 64 blocks per workgroup)
    max_small_groups = std::max(max_small_groups, (j.n_tu[2] + 15) / 16 + (j.n_tu[1] + 31) / 32 + (j.n_tu[0] + 63) / 64);
    j.coeff = (const uint32_t *)(y->up_dev + y->off_coeff);
    for (int p = 0; p < 3; p++) j.resi[p] = (int16_t *)((char *)ws.resi_base[i] + c->resi_off[p]);
    j.intra = (const IntraItem *)(y->up_dev + y->off_intra);
    j.ictu = (const IntraCtu *)(y->up_dev + y->off_ictu);
    j.rounds = (const uint16_t *)(y->up_dev + y->off_lvlc);
    j.wave_off = (const int32_t *)(y->up_dev + y->off_wave);
    j.n_waves = k.wave_off.empty() ? 0 : (int)k.wave_off.size() - 1;
    if (j.n_waves > max_levels) max_levels = j.n_waves;
    j.ctu_done = ws.sync_dev + kSyncHead + (size_t)i * s.ctus_w * s.ctus_h;
    j.sao = (const SaoCtu *)(y->up_dev + y->off_sao);
    j.ctu_info = k.ctu_info.empty() ? nullptr : (const CtuInfo *)(y->up_dev + y->off_ctui);
    j.scaling = k.use_scaling ? (const ScalingTables *)(y->up_dev + y->off_scal) : nullptr;
    j.wp = k.use_wp ? (const WpTables *)(y->up_dev + y->off_wp) : nullptr;
    any_wp = any_wp || k.use_wp;
    j.slice_type = k.pp.slice_type;
    // with a slice table the picture is deblocked unless every slice says no (the kernel then looks at each CTU's slice)
    bool deblock_off = k.pp.deblocking_disable != 0;
    if (!k.slices.empty()) {
      deblock_off = true;
      for (size_t q = 0; q < k.slices.size(); q++) deblock_off = deblock_off && k.slices[q].deblocking_disable;
    }
    j.deblock_disable = deblock_off ? 1 : 0;
    j.beta_off_div2 = k.pp.beta_offset_div2;
    j.tc_off_div2 = k.pp.tc_offset_div2;
    j.cb_off_pps = k.pp.cb_qp_offset_pps;
    j.cr_off_pps = k.pp.cr_qp_offset_pps;
    j.cb_off = k.pp.cb_qp_offset_pps + k.pp.cb_qp_delta_slice;
    j.cr_off = k.pp.cr_qp_offset_pps + k.pp.cr_qp_delta_slice;
    j.sao_luma = k.pp.sao_luma;
    j.sao_chroma = k.pp.sao_chroma;
    j.has_bypass = k.has_bypass ? 1 : 0;
    j.has_pcm = k.has_pcm ? 1 : 0;
    j.error_flag = c->error_flag_dev;
    any_deblock = any_deblock || !deblock_off;
  }
  bool uploaded_any = false;
  {  // hand-over of the pictures whose device copy is stale
    int64_t newest_reader = -1;
    bool any = false;
    for (int i = 0; i < n; i++)
      if (syn[i]->dirty) {
        any = true;
        newest_reader = std::max(newest_reader, syn[i]->last_batch);
      }
    if (any) {
      // kernels already queued may still read a device copy that is about to be rewritten
      if (newest_reader > c->batch_waited) {
        for (int k = 0; k < g_copy_used; k++) HIPCHK(c, hipStreamWaitEvent(c->copy_stream[k], c->batch_ev[newest_reader % kRing], 0));
        c->batch_waited = newest_reader;
      }
      // pictures that lie back to back in the upload ring travel in one copy (cut into pieces of a few megabytes so
      // that several copy streams, when in use, share the work)
      // (in the order of their staging addresses, whatever the order of the batch)
      const size_t kPiece = (size_t)8 << 20;
      hevcrecon_syntax *stale[HR_MAX_BATCH];  // (on the stack: this path allocates nothing)
      size_t n_stale = 0;
      for (int i = 0; i < n; i++)
        if (syn[i]->dirty) stale[n_stale++] = syn[i];
      std::sort(stale, stale + n_stale, [](const hevcrecon_syntax *a, const hevcrecon_syntax *b) { return a->up_host < b->up_host; });
      char *run_h = nullptr, *run_d = nullptr;
      size_t run_n = 0;
      for (size_t i = 0; i <= n_stale; i++) {
        hevcrecon_syntax *y = i < n_stale ? stale[i] : nullptr;
        if (y && run_n && run_n + y->upload_bytes <= kPiece && y->up_host == run_h + run_n && y->up_dev == run_d + run_n) {
          run_n += y->upload_bytes;
        } else {
          if (run_n) {
            HIPCHK(c, hipMemcpyAsync(run_d, run_h, run_n, hipMemcpyHostToDevice, c->copy_stream[c->copy_rr]));
            c->copy_rr = (c->copy_rr + 1) % g_copy_used;
          }
          run_h = y ? y->up_host : nullptr;
          run_d = y ? y->up_dev : nullptr;
          run_n = y ? y->upload_bytes : 0;
        }
      }
      uploaded_any = true;
    }
    // device copies complete: events behind everything the copy streams hold cover the whole batch
    bool wait = uploaded_any;
    for (int i = 0; i < n; i++) wait = wait || syn[i]->upload_seq > c->upload_waited;
    if (wait) {
      for (int k = 0; k < g_copy_used; k++) {
        hipEvent_t e = c->copy_ev[c->copy_ev_pos * kCopyStreams + k];
        HIPCHK(c, hipEventRecord(e, c->copy_stream[k]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
      }
      c->copy_ev_pos = (c->copy_ev_pos + 1) % kRing;
    }
  }
  // The intra kernel takes the whole batch in one launch.  Its CTUs go into a list in which every CTU follows the
  // ones it reads from: the first waves of all pictures, then their second waves, and so on (a single picture's own
  // table is in that order already and needs no list).  The list travels with the job descriptors, ahead of the
  // event that tells the host when this ring slot's pinned memory may be written again.
  size_t n_intra_ctus = 0;
  int intra_parallel = 0;  // CTUs of the batch's largest wave
  const uint32_t *intra_list = nullptr;
  if (max_levels > 0) {
    uint32_t *lh = ws.list_host;
    for (int l = 0; l < max_levels; l++) {
      int in_wave = 0;
      for (int i = 0; i < n; i++) {
        const std::vector<int32_t> &lo = syn[i]->pk.wave_off;
        if (l + 1 < (int)lo.size()) {
          in_wave += lo[l + 1] - lo[l];
          if (n > 1)
            for (int32_t k = lo[l]; k < lo[l + 1]; k++) lh[n_intra_ctus++] = ((uint32_t)i << 16) | (uint32_t)k;
        }
      }
      intra_parallel = std::max(intra_parallel, in_wave);
    }
    if (n == 1) {
      n_intra_ctus = syn[0]->pk.ictu.size();
    } else {
      HIPCHK(c, hipMemcpyAsync(ws.list_dev, lh, n_intra_ctus * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      intra_list = ws.list_dev;
    }
  }
  HIPCHK(c, hipMemcpyAsync(jd, jh, sizeof(PicJob) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ring_ev[ws.slot], c->stream));
  c->ring_used[ws.slot] = true;
  const bool stages = c->keep_stages && !ws.fixed;
  // ---- launches ----
  if (any_expand) {
    Timed tm(c, K_OTHER, 1, !ws.fixed);
    hr_launch(K_OTHER, jd, n, s, c->stream, 0, 0);
  }
  if (max_mct[0] > 0 || max_mct[1] > 0 || max_mct[2] > 0 || max_mcb > 0 || max_mcb8 > 0) {
    // arg = 0..2: uniform regions of 64, 32, 16 samples; 3: the mixed rest; +4: the variants with explicit weights
    const int cls[4] = {K_INTER, K_INTER_T32, K_INTER_T16, K_INTER_BLK};
    for (int t = 0; t < 3; t++)
      if (max_mct[t] > 0) {
        Timed tm(c, cls[t], 1, !ws.fixed);
        hr_launch(K_INTER, jd, n, s, c->stream, t + (any_wp ? 4 : 0), max_mct[t]);
      }
    if (max_mcb > 0 || max_mcb8 > 0) {  // the mixed rest: 8x8 blocks of one motion, then unit by unit (one timer for both)
      Timed tm(c, cls[3], (max_mcb > 0) + (max_mcb8 > 0), !ws.fixed);
      if (max_mcb8 > 0) hr_launch(K_INTER, jd, n, s, c->stream, 8 + (any_wp ? 1 : 0), max_mcb8);
      if (max_mcb > 0) hr_launch(K_INTER, jd, n, s, c->stream, 3 + (any_wp ? 4 : 0), max_mcb);
    }
  }
  {
    // the 32x32 blocks in one launch, all smaller blocks in another
    const int nl = (max_tu[3] > 0) + (max_small_groups > 0);
    if (nl) {
      Timed tm(c, K_TRANSFORM, nl, !ws.fixed);
      if (max_tu[3] > 0) hr_launch(K_TRANSFORM, jd, n, s, c->stream, 5, max_tu[3]);
      if (max_small_groups > 0) hr_launch(K_TRANSFORM, jd, n, s, c->stream, 0, max_small_groups);
    }
  }
  if (n_intra_ctus > 0) {
    HIPCHK(c, hipMemsetAsync(ws.sync_dev, 0, (kSyncHead + (size_t)n * s.ctus_w * s.ctus_h) * sizeof(uint32_t), c->stream));
    Timed tm(c, K_INTRA, 1, !ws.fixed);
    // "dense": pictures coded mostly in intra blocks, and enough of them to keep more workgroups busy
    const bool dense = n >= 8 && n_intra_ctus * 4 >= (size_t)n * s.ctus_w * s.ctus_h * 3;
    bool inside = false;
    for (int i = 0; i < n; i++) inside = inside || jh[i].slice_type != 2 || jh[i].has_pcm;
    hr_launch_intra(jd, s, intra_list, (int)n_intra_ctus, intra_parallel, dense, inside, ws.sync_dev, c->stream);
  }
  if (stages) {
    Timed tm(c, K_COPY, 1);
    HIPCHK(c, hipMemcpyAsync(c->stage[0].base, jh[0].rec[0], c->pic_bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  if (any_deblock) {
    {
      Timed tm(c, K_DEBLOCK_V, 1, !ws.fixed);
      hr_launch(K_DEBLOCK_V, jd, n, s, c->stream, 0, 0);
    }
    {
      Timed tm(c, K_DEBLOCK_H, 1, !ws.fixed);
      hr_launch(K_DEBLOCK_H, jd, n, s, c->stream, 0, 0);
    }
  }
  if (stages) {
    Timed tm(c, K_COPY, 1);
    HIPCHK(c, hipMemcpyAsync(c->stage[1].base, jh[0].rec[0], c->pic_bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  if (any_sao) {
    Timed tm(c, K_SAO, 1, !ws.fixed);
    hr_launch(K_SAO, jd, n, s, c->stream, 0, 0);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->batch_ev[c->batch_no % kRing], c->stream));
  // ---- everything is queued: commit ----
  for (int i = 0; i < n; i++) {
    hevcrecon_syntax *y = syn[i];
    if (y->dirty) {
      y->dirty = false;
      y->upload_seq = c->upload_no;
    }
    y->expanded = true;
    y->last_batch = c->batch_no;
  }
  if (uploaded_any) c->upload_no++;
  c->upload_waited = c->upload_no - 1;
  c->batch_no++;
  for (int i = 0; i < n; i++) {
    c->pics[dst_ids[i]].valid = true;
    c->pics[dst_ids[i]].poc = syn[i]->pk.pp.poc;
  }
  if (!ws.fixed && c->intervals.size() > 4096) drain_intervals(c);
  return 0;
}

}  // namespace

extern "C" int hevcrecon_reconstruct_batch(hevcrecon_ctx *c, int n, hevcrecon_syntax *const *syn, const int *dst_ids) {
  if (!c || !syn || !dst_ids) return 1;
  if (n <= 0 || n > HR_MAX_BATCH) return fail(c, "batch size must be 1..256");
  if (!c->have_seq) return fail(c, "no sequence parameters");
  HIPCHK(c, hipSetDevice(c->device));
  if (validate_batch(c, n, syn, dst_ids)) return 1;
  // ---- device pictures ----
  for (int i = 0; i < n; i++) {
    if ((int)c->pics.size() <= dst_ids[i]) c->pics.resize(dst_ids[i] + 1);
    if (alloc_picture(c, c->pics[dst_ids[i]])) return 1;
  }
  if ((int)c->work.size() < n) c->work.resize(n);
  if ((int)c->work_resi.size() < n) c->work_resi.resize(n, nullptr);
  for (int i = 0; i < n; i++) {
    if (alloc_picture(c, c->work[i])) return 1;
    if (!c->work_resi[i]) HIPCHK(c, hipMalloc(&c->work_resi[i], c->resi_bytes));
  }
  if (c->keep_stages)
    for (int k = 0; k < 2; k++)
      if (alloc_picture(c, c->stage[k])) return 1;
  const size_t nctu = (size_t)c->si.ctus_w * c->si.ctus_h;
  if (c->sync_words < kSyncHead + n * nctu) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->sync_dev) hipFree(c->sync_dev);
    c->sync_dev = nullptr;
    c->sync_words = 0;
    const size_t words = kSyncHead + std::max<size_t>(n, 16) * nctu;
    HIPCHK(c, hipMalloc((void **)&c->sync_dev, words * sizeof(uint32_t)));
    c->sync_words = words;
  }
  if (n > 1 && c->list_cap < n * nctu) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->list_host) hipHostFree(c->list_host);
    if (c->list_dev) hipFree(c->list_dev);
    c->list_host = c->list_dev = nullptr;
    c->list_cap = 0;
    const size_t cap = std::max<size_t>(n, 16) * nctu;
    HIPCHK(c, hipHostMalloc((void **)&c->list_host, cap * kRing * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&c->list_dev, cap * kRing * sizeof(uint32_t)));
    c->list_cap = cap;
  }
  WorkSet ws;
  ws.slot = c->ring_pos;
  ws.jobs_host = c->jobs_host + (size_t)ws.slot * HR_MAX_BATCH;
  ws.jobs_dev = c->jobs_dev + (size_t)ws.slot * HR_MAX_BATCH;
  ws.sync_dev = c->sync_dev;
  ws.list_host = c->list_host + (size_t)ws.slot * c->list_cap;
  ws.list_dev = c->list_dev + (size_t)ws.slot * c->list_cap;
  ws.fixed = false;
  for (int i = 0; i < n; i++) {
    ws.work_base[i] = c->work[i].base;
    ws.resi_base[i] = c->work_resi[i];
  }
  if (int rc = reconstruct_core(c, n, syn, dst_ids, ws)) return rc;
  c->ring_pos = (c->ring_pos + 1) % kRing;
  return 0;
}

// The allocation-free variant of hevcrecon_reconstruct: same validation, same kernels, but job descriptor, working
// picture and residual planes lie in the library's fixed device array, the destination picture must exist already,
// and nothing here (or below) calls hipMalloc, hipHostMalloc, hipFree or creates an event.
extern "C" int hevcrecon_reconstruct_noalloc(hevcrecon_ctx *c, hevcrecon_syntax *syn, int dst_pic_id) {
  if (!c || !syn) return 1;
  if (!c->have_seq) return fail(c, "no sequence parameters");
  HIPCHK(c, hipSetDevice(c->device));
  hevcrecon_syntax *one[1] = {syn};
  if (validate_batch(c, 1, one, &dst_pic_id)) return 1;
  if (dst_pic_id >= (int)c->pics.size() || !c->pics[dst_pic_id].base)
    return fail(c, "hevcrecon_reconstruct_noalloc allocates nothing: the destination picture must exist already "
                   "(created by hevcrecon_picture_upload, hevcrecon_picture_copy or an earlier hevcrecon_reconstruct)");
  const size_t off_work = kScratchJobBytes + kScratchSyncBytes, off_resi = off_work + align_up(c->pic_bytes, 256);
  if (off_resi + c->resi_bytes > HR_SCRATCH_BYTES || (kSyncHead + (size_t)c->si.ctus_w * c->si.ctus_h) * sizeof(uint32_t) > kScratchSyncBytes)
    return fail(c, "hevcrecon_reconstruct_noalloc: picture too large for the fixed 16 MiB work area "
                   "(it holds pictures up to 1920x1088 at any bit depth); use hevcrecon_reconstruct");
  unsigned char *base = g_scratch_ptr[c->device];
  if (!base) return fail(c, "the fixed work area was not found on this device");
  static_assert(sizeof(PicJob) <= kScratchJobBytes, "job descriptor outgrew its place in the fixed work area");
  WorkSet ws;
  ws.slot = c->ring_pos;
  ws.jobs_host = c->jobs_host + (size_t)ws.slot * HR_MAX_BATCH;  // pinned ring allocated by hevcrecon_create
  ws.jobs_dev = (PicJob *)base;
  ws.sync_dev = (uint32_t *)(base + kScratchJobBytes);  // zeroed by the call itself before the intra kernel starts
  ws.list_host = ws.list_dev = nullptr;                 // a single picture needs no list
  ws.work_base[0] = base + off_work;
  ws.resi_base[0] = base + off_resi;
  ws.fixed = true;
  // Contexts of one GPU share the array: a call orders itself behind the last call of another context (a wait in the
  // stream, nothing allocated), so that overlapping use from two contexts is serialised instead of corrupting both.
  std::lock_guard<std::mutex> lock(g_copy_mutex);
  ScratchUser &u = g_scratch_user[c->device];
  if (u.ctx && u.ctx != c && u.done) HIPCHK(c, hipStreamWaitEvent(c->stream, u.done, 0));
  const int64_t batch = c->batch_no;
  if (int rc = reconstruct_core(c, 1, one, &dst_pic_id, ws)) return rc;
  c->ring_pos = (c->ring_pos + 1) % kRing;
  u.ctx = c;
  u.done = c->batch_ev[batch % kRing];  // recorded behind the call's last kernel (a later record of the slot is later still)
  return 0;
}

extern "C" int hevcrecon_reconstruct(hevcrecon_ctx *c, hevcrecon_syntax *syn, int dst_pic_id) {
  hevcrecon_syntax *one[1] = {syn};
  return hevcrecon_reconstruct_batch(c, 1, one, &dst_pic_id);
}

// TDecTop::xCreateLostPicture (TLibDecoder/TDecTop.cpp:214-261) conceals a lost reference with a copy of the
// nearest picture (TComPicYuv::copyToPic, :244): here a device-to-device copy, queued behind the kernels that
// write the source.
extern "C" int hevcrecon_picture_copy(hevcrecon_ctx *c, int dst_pic_id, int src_pic_id, int dst_poc) {
  if (!c) return 1;
  if (!c->have_seq) return fail(c, "no sequence parameters");
  if (src_pic_id < 0 || src_pic_id >= (int)c->pics.size() || !c->pics[src_pic_id].valid) return fail(c, "source picture is not resident");
  if (dst_pic_id < 0 || dst_pic_id >= kMaxPics) return fail(c, "picture id out of range");
  if (dst_pic_id == src_pic_id) {
    c->pics[dst_pic_id].poc = dst_poc;
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  if ((int)c->pics.size() <= dst_pic_id) c->pics.resize(dst_pic_id + 1);
  if (alloc_picture(c, c->pics[dst_pic_id])) return 1;
  HIPCHK(c, hipMemcpyAsync(c->pics[dst_pic_id].base, c->pics[src_pic_id].base, c->pic_bytes, hipMemcpyDeviceToDevice, c->stream));
  c->pics[dst_pic_id].valid = true;
  c->pics[dst_pic_id].poc = dst_poc;
  return 0;
}

extern "C" int hevcrecon_picture_download(hevcrecon_ctx *c, int pic_id, int16_t *y, int16_t *cb, int16_t *cr, int stride_y,
                                          int stride_c) {
  if (!c || !y || !cb || !cr) return 1;
  if (pic_id < 0 || pic_id >= (int)c->pics.size() || !c->pics[pic_id].valid) return fail(c, "picture is not resident");
  HIPCHK(c, hipSetDevice(c->device));
  return download_planes(c, c->pics[pic_id], y, cb, cr, stride_y, stride_c);
}

extern "C" int hevcrecon_picture_download_file(hevcrecon_ctx *c, int pic_id, int file_bit_depth_luma, int file_bit_depth_chroma, int width,
                                               int height, void *out, int64_t out_bytes) {
  if (!c || !out) return 1;
  if (pic_id < 0 || pic_id >= (int)c->pics.size() || !c->pics[pic_id].valid) return fail(c, "picture is not resident");
  const SeqInfo &s = c->si;
  if (width <= 0 || height <= 0 || width > s.w || height > s.h || (width & 1) || (height & 1) || file_bit_depth_luma < 8 ||
      file_bit_depth_luma > 16 || file_bit_depth_chroma < 8 || file_bit_depth_chroma > 16)
    return fail(c, "bad output size or file bit depth");
  const bool wide = file_bit_depth_luma > 8 || file_bit_depth_chroma > 8;
  const size_t need = ((size_t)width * height + 2 * (size_t)(width / 2) * (height / 2)) * (wide ? 2 : 1);
  if (out_bytes < (int64_t)need) return fail(c, "output buffer too small");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->out_cap < need) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->out_dev) hipFree(c->out_dev);
    c->out_dev = nullptr;
    c->out_cap = 0;
    HIPCHK(c, hipMalloc(&c->out_dev, need));
    c->out_cap = need;
  }
  if (ensure_staging(c, need)) return 1;
  const void *planes[3] = {c->pics[pic_id].p[0], c->pics[pic_id].p[1], c->pics[pic_id].p[2]};
  hr_launch_output(planes, c->out_dev, s, width, height, file_bit_depth_luma, file_bit_depth_chroma, wide, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->dl_host, c->out_dev, need, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out, c->dl_host, need);
  return check_device_errors(c);  // a picture reconstructed from a work list a kernel refused is not a result
}

extern "C" int hevcrecon_picture_md5_batch(hevcrecon_ctx *c, int n, const int *pic_ids, uint8_t *digests) {
  if (!c || !pic_ids || !digests) return 1;
  if (n <= 0 || n > HR_MD5_MAX_PICTURES) return fail(c, "number of pictures to digest out of range");
  for (int i = 0; i < n; i++)
    if (pic_ids[i] < 0 || pic_ids[i] >= (int)c->pics.size() || !c->pics[pic_ids[i]].valid) return fail(c, "picture is not resident");
  HIPCHK(c, hipSetDevice(c->device));
  const SeqInfo &s = c->si;
  const size_t pb = c->pel_bytes;
  const size_t desc_bytes = align_up(sizeof(Md5Plane) * 3 * n, 256), dig_bytes = (size_t)48 * n;
  if (ensure_staging(c, desc_bytes + dig_bytes)) return 1;
  if (c->out_cap < desc_bytes + dig_bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->out_dev) hipFree(c->out_dev);
    c->out_dev = nullptr;
    c->out_cap = 0;
    HIPCHK(c, hipMalloc(&c->out_dev, desc_bytes + dig_bytes));
    c->out_cap = desc_bytes + dig_bytes;
  }
  // lanes of a wave should run equally long: all luma planes first, then the chroma planes
  Md5Plane *d = (Md5Plane *)c->dl_host;
  for (int i = 0; i < n; i++) {
    const DevPicture &p = c->pics[pic_ids[i]];
    d[i] = Md5Plane{p.p[0], (uint32_t)(s.w * pb), (uint32_t)s.h, (uint32_t)(s.stride_y * pb), 0};
    d[n + 2 * i] = Md5Plane{p.p[1], (uint32_t)(s.cw * pb), (uint32_t)s.ch, (uint32_t)(s.stride_c * pb), 0};
    d[n + 2 * i + 1] = Md5Plane{p.p[2], (uint32_t)(s.cw * pb), (uint32_t)s.ch, (uint32_t)(s.stride_c * pb), 0};
  }
  uint8_t *dig_dev = (uint8_t *)c->out_dev + desc_bytes;
  HIPCHK(c, hipMemcpyAsync(c->out_dev, d, sizeof(Md5Plane) * 3 * n, hipMemcpyHostToDevice, c->stream));
  hr_launch_md5((const Md5Plane *)c->out_dev, 3 * n, dig_dev, c->stream);
  HIPCHK(c, hipGetLastError());
  uint8_t *dig_host = (uint8_t *)c->dl_host + desc_bytes;
  HIPCHK(c, hipMemcpyAsync(dig_host, dig_dev, dig_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) {
    std::memcpy(digests + 48 * i, dig_host + 16 * i, 16);
    std::memcpy(digests + 48 * i + 16, dig_host + 16 * (n + 2 * i), 32);
  }
  return check_device_errors(c);
}

// The same hash without blocking: the kernel runs on a stream of its own behind everything queued on the context so
// far, beside the reconstruction of later pictures; the digests come back into pinned memory and are collected with
// hevcrecon_picture_md5_poll / _wait.  A picture that a pending ticket still reads is protected: whatever would
// overwrite it is ordered behind the hash kernel.
static int md5_submit_impl(hevcrecon_ctx *c, int n, const int *pic_ids, int64_t *ticket, bool copies) {
  if (!c || !pic_ids || !ticket) return 1;
  *ticket = -1;
  if (n <= 0 || n > HR_MD5_MAX_PICTURES) return fail(c, "number of pictures to digest out of range");
  for (int i = 0; i < n; i++)
    if (pic_ids[i] < 0 || pic_ids[i] >= (int)c->pics.size() || !c->pics[pic_ids[i]].valid) return fail(c, "picture is not resident");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->md5_stream) {
    // lowest priority: a hardware queue of its own (streams of one priority share a few queues, and a hash kernel
    // runs for milliseconds: nothing of the reconstruction may ever queue up behind one), served when the
    // reconstruction kernels leave room
    int least = 0, greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (const char *e = std::getenv("HEVCRECON_HASH_PRIORITY"))  // experiments: "normal" puts the hash streams on a par with the kernels' stream
      if (!std::strcmp(e, "normal")) least = 0;
    for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamCreateWithPriority(&c->md5_streams[i], hipStreamNonBlocking, least));
    HIPCHK(c, hipEventCreateWithFlags(&c->md5_fork, hipEventDisableTiming));
    c->md5_stream = true;
  }
  Md5Slot &m = c->md5_slots[c->md5_next % kMd5Slots];
  if (m.ticket >= 0) return fail(c, "too many picture hashes in flight: collect earlier tickets with hevcrecon_picture_md5_poll or _wait first");
  const size_t desc_bytes = align_up(sizeof(Md5Plane) * 3 * n, 256), dig_bytes = (size_t)48 * n;
  if (m.cap < (size_t)n) {
    if (m.dev) c->retired.push_back(m.dev);  // (freeing waits for the whole device)
    if (m.host) hipHostFree(m.host);
    m.dev = nullptr;
    m.host = nullptr;
    m.cap = 0;
    const size_t cap = std::max<size_t>(2 * (size_t)n, 64);
    const size_t bytes = align_up(sizeof(Md5Plane) * 3 * cap, 256) + 48 * cap + 64;
    HIPCHK(c, hipMalloc(&m.dev, bytes));
    HIPCHK(c, hipHostMalloc((void **)&m.host, bytes));
    m.cap = cap;
  }
  if (copies && (m.snap_cap < (size_t)n || m.snap_pic_bytes != c->pic_bytes)) {
    if (m.snap) c->retired.push_back(m.snap);
    m.snap = nullptr;
    m.snap_cap = 0;
    const size_t cap = std::max<size_t>(n, 4);
    HIPCHK(c, hipMalloc(&m.snap, cap * c->pic_bytes));
    m.snap_cap = cap;
    m.snap_pic_bytes = c->pic_bytes;
  }
  if (!m.done) HIPCHK(c, hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
  const SeqInfo &s = c->si;
  const size_t pb = c->pel_bytes;
  Md5Plane *d = (Md5Plane *)m.host;  // descriptors, then (after the copy back) the digests, in the slot's pinned buffer
  for (int i = 0; i < n; i++) {
    // the pictures as they are when everything queued so far has run, copied in the kernels' stream
    char *snap = (char *)c->pics[pic_ids[i]].base;
    if (copies) {
      snap = (char *)m.snap + (size_t)i * c->pic_bytes;
      HIPCHK(c, hipMemcpyAsync(snap, c->pics[pic_ids[i]].base, c->pic_bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    d[i] = Md5Plane{snap + c->plane_off[0], (uint32_t)(s.w * pb), (uint32_t)s.h, (uint32_t)(s.stride_y * pb), 0};
    d[n + 2 * i] = Md5Plane{snap + c->plane_off[1], (uint32_t)(s.cw * pb), (uint32_t)s.ch, (uint32_t)(s.stride_c * pb), 0};
    d[n + 2 * i + 1] = Md5Plane{snap + c->plane_off[2], (uint32_t)(s.cw * pb), (uint32_t)s.ch, (uint32_t)(s.stride_c * pb), 0};
  }
  hipStream_t hs = c->md5_streams[c->md5_next % 4];
  HIPCHK(c, hipEventRecord(c->md5_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(hs, c->md5_fork, 0));
  uint8_t *dig_dev = (uint8_t *)m.dev + desc_bytes;
  HIPCHK(c, hipMemcpyAsync(m.dev, d, sizeof(Md5Plane) * 3 * n, hipMemcpyHostToDevice, hs));
  hr_launch_md5((const Md5Plane *)m.dev, 3 * n, dig_dev, hs);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(m.host + desc_bytes, dig_dev, dig_bytes, hipMemcpyDeviceToHost, hs));
  // the error words of the kernels that wrote the pictures (they ran before the copies above) travel with the digests
  HIPCHK(c, hipMemcpyAsync(m.host + desc_bytes + dig_bytes, c->error_flag_dev, 64, hipMemcpyDeviceToHost, hs));
  HIPCHK(c, hipEventRecord(m.done, hs));
  m.ticket = c->md5_next++;
  m.n = n;
  *ticket = m.ticket;
  return 0;
}

extern "C" int hevcrecon_picture_md5_submit(hevcrecon_ctx *c, int n, const int *pic_ids, int64_t *ticket) {
  return md5_submit_impl(c, n, pic_ids, ticket, true);
}

extern "C" int hevcrecon_picture_md5_submit_in_place(hevcrecon_ctx *c, int n, const int *pic_ids, int64_t *ticket) {
  return md5_submit_impl(c, n, pic_ids, ticket, false);
}

static int md5_collect(hevcrecon_ctx *c, int64_t ticket, bool block, int *done, uint8_t *digests) {
  if (!c || !digests || ticket < 0) return 1;
  Md5Slot &m = c->md5_slots[ticket % kMd5Slots];
  if (m.ticket != ticket) return fail(c, "unknown picture hash ticket (never issued, or collected already)");
  HIPCHK(c, hipSetDevice(c->device));
  if (block) {
    HIPCHK(c, hipEventSynchronize(m.done));
  } else {
    const hipError_t e = hipEventQuery(m.done);
    if (e == hipErrorNotReady) {
      (void)hipGetLastError();
      if (done) *done = 0;
      return 0;
    }
    HIPCHK(c, e);
  }
  const int n = m.n;
  const uint8_t *dig_host = m.host + align_up(sizeof(Md5Plane) * 3 * n, 256);
  for (int i = 0; i < n; i++) {
    std::memcpy(digests + 48 * i, dig_host + 16 * i, 16);
    std::memcpy(digests + 48 * i + 16, dig_host + 16 * (n + 2 * i), 32);
  }
  m.ticket = -1;
  if (done) *done = 1;
  int32_t err0;
  std::memcpy(&err0, dig_host + (size_t)48 * n, 4);
  if (err0) return check_device_errors(c);  // reads the words again (waiting for the kernels' stream) and makes the message
  return 0;
}

extern "C" int hevcrecon_picture_md5_poll(hevcrecon_ctx *c, int64_t ticket, int *done, uint8_t *digests) {
  if (!done) return 1;
  *done = 0;
  return md5_collect(c, ticket, false, done, digests);
}

extern "C" int hevcrecon_picture_md5_wait(hevcrecon_ctx *c, int64_t ticket, uint8_t *digests) {
  return md5_collect(c, ticket, true, nullptr, digests);
}

extern "C" int hevcrecon_debug_keep_stages(hevcrecon_ctx *c, int on) {
  if (!c) return 1;
  c->keep_stages = on != 0;
  return 0;
}

extern "C" int hevcrecon_debug_download_stage(hevcrecon_ctx *c, int stage, int16_t *y, int16_t *cb, int16_t *cr, int stride_y,
                                              int stride_c) {
  if (!c || !y || !cb || !cr || stage < 0 || stage > 1) return 1;
  if (!c->keep_stages || !c->stage[stage].base) return fail(c, "stage pictures are kept only after hevcrecon_debug_keep_stages(ctx, 1)");
  HIPCHK(c, hipSetDevice(c->device));
  return download_planes(c, c->stage[stage], y, cb, cr, stride_y, stride_c);
}

extern "C" int hevcrecon_picture_upload(hevcrecon_ctx *c, int pic_id, int poc, const int16_t *y, const int16_t *cb, const int16_t *cr,
                                        int stride_y, int stride_c) {
  if (!c || !y || !cb || !cr) return 1;
  if (!c->have_seq) return fail(c, "no sequence parameters");
  if (pic_id < 0 || pic_id >= kMaxPics) return fail(c, "picture id out of range");
  HIPCHK(c, hipSetDevice(c->device));
  if ((int)c->pics.size() <= pic_id) c->pics.resize(pic_id + 1);
  DevPicture &p = c->pics[pic_id];
  if (alloc_picture(c, p)) return 1;
  const SeqInfo &s = c->si;
  const size_t pb = c->pel_bytes;
  const size_t ny = (size_t)s.w * s.h * pb, nc = (size_t)s.cw * s.ch * pb;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamSynchronize(c->md5_streams[i]));
  if (ensure_staging(c, ny + 2 * nc)) return 1;
  char *h = (char *)c->dl_host;
  const int16_t *src[3] = {y, cb, cr};
  char *dst[3] = {h, h + ny, h + ny + nc};
  for (int k = 0; k < 3; k++) {
    const int w = k ? s.cw : s.w, hh = k ? s.ch : s.h, st = k ? stride_c : stride_y;
    for (int r = 0; r < hh; r++) {
      const int16_t *q = src[k] + (size_t)r * st;
      if (pb == 1) {
        uint8_t *d = (uint8_t *)dst[k] + (size_t)r * w;
        for (int i = 0; i < w; i++) d[i] = (uint8_t)q[i];
      } else {
        uint16_t *d = (uint16_t *)dst[k] + (size_t)r * w;
        for (int i = 0; i < w; i++) d[i] = (uint16_t)q[i];
      }
    }
  }
  HIPCHK(c, hipMemcpy2DAsync(p.p[0], s.stride_y * pb, h, s.w * pb, s.w * pb, s.h, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(p.p[1], s.stride_c * pb, h + ny, s.cw * pb, s.cw * pb, s.ch, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(p.p[2], s.stride_c * pb, h + ny + nc, s.cw * pb, s.cw * pb, s.ch, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  p.valid = true;
  p.poc = poc;
  return 0;
}

extern "C" int hevcrecon_picture_release(hevcrecon_ctx *c, int pic_id) {
  if (!c) return 1;
  if (pic_id < 0 || pic_id >= (int)c->pics.size()) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamSynchronize(c->md5_streams[i]));
  free_picture(c->pics[pic_id]);
  return 0;
}

extern "C" int hevcrecon_synchronize(hevcrecon_ctx *c) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  for (int k = 0; k < kCopyStreams; k++) HIPCHK(c, hipStreamSynchronize(c->copy_stream[k]));
  c->upload_synced = c->upload_no - 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->md5_stream)
    for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamSynchronize(c->md5_streams[i]));
  for (size_t i = 0; i < c->retired.size(); i++) hipFree(c->retired[i]);
  c->retired.clear();
  return check_device_errors(c);
}

// ---------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------
extern "C" int hevcrecon_profile_enable(hevcrecon_ctx *c, int on) {
  if (!c) return 1;
  drain_intervals(c);
  c->profiling = on != 0;
  std::memset(c->ms, 0, sizeof(c->ms));
  std::memset(c->launches, 0, sizeof(c->launches));
  return 0;
}

extern "C" int hevcrecon_profile_read(hevcrecon_ctx *c, double ms_out[HEVCRECON_NUM_KERNELS], int64_t launches_out[HEVCRECON_NUM_KERNELS]) {
  if (!c || !ms_out || !launches_out) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_intervals(c);
  for (int i = 0; i < HEVCRECON_NUM_KERNELS; i++) {
    ms_out[i] = c->ms[i];
    launches_out[i] = c->launches[i];
  }
  return 0;
}

extern "C" const char *hevcrecon_kernel_name(int idx) {
  static const char *names[K_COUNT] = {"k_inter_tile64", "k_transform", "k_intra",        "k_deblock_v",    "k_deblock_h", "k_sao",
                                       "stage_copy",     "k_expand",    "k_inter_tile32", "k_inter_tile16", "k_inter_blk"};
  return idx >= 0 && idx < K_COUNT ? names[idx] : "";
}
