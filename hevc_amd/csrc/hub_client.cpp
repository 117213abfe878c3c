// hub_client.cpp — the decoder-process half of the hub protocol (hub_proto.h): entry points hevcrecon_hub_* of
// include/hevcrecon.h.  Plain host code: a process that uses only these and host-only syntax objects never opens the GPU.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "../../include/hevcrecon.h"
#include "hub_proto.h"

struct hevcrecon_hub {
  HubShared *sh = nullptr;
  HubClient *me = nullptr;
  int slot = -1;
  uint64_t blob_head = 0;  // where the next packed picture goes
  std::string err;
};

namespace {
void nap_us(long us) {
  timespec ts = {0, us * 1000};
  nanosleep(&ts, nullptr);
}
std::string shm_name(const char *name) { return std::string("/hevcrecon_hub_") + (name && *name ? name : "default"); }

int fail(hevcrecon_hub *h, const std::string &m) {
  if (h) h->err = m;
  return 1;
}

// Has the hub refused something of ours, or gone away?
int check(hevcrecon_hub *h) {
  if (h->me->failed.load(std::memory_order_acquire)) return fail(h, std::string("the hub refused a request: ") + h->me->error);
  if (h->sh->shutdown.load(std::memory_order_acquire)) return fail(h, "the hub is shutting down");
  return 0;
}

// Room for one more request (and `bytes` of packed picture, contiguous): returns the offset, waiting for the hub to
// catch up when there is none.
int reserve(hevcrecon_hub *h, uint64_t bytes, uint64_t *off_out) {
  HubClient *c = h->me;
  if (bytes > HR_HUB_BLOB_BYTES / 2) return fail(h, "packed picture larger than the hub's request ring takes");
  for (long spins = 0;; spins++) {
    if (int rc = check(h)) return rc;
    const uint64_t prod = c->produced.load(std::memory_order_relaxed), cons = c->consumed.load(std::memory_order_acquire);
    if (prod - cons < HR_HUB_ENTRIES) {
      if (prod == cons) {  // nothing outstanding: start over at the front
        h->blob_head = 0;
        *off_out = 0;
        return 0;
      }
      const uint64_t tail = c->entries[cons % HR_HUB_ENTRIES].blob_off, head = h->blob_head;
      if (head >= tail) {  // free: [head, end) and [0, tail)
        if (HR_HUB_BLOB_BYTES - head >= bytes) {
          *off_out = head;
          return 0;
        }
        if (tail > bytes) {
          *off_out = 0;
          return 0;
        }
      } else if (tail - head > bytes) {
        *off_out = head;
        return 0;
      }
    }
    nap_us(spins < 100 ? 20 : 200);
    if (spins > 5 * 60 * 5000L) return fail(h, "the hub has not taken a request for minutes");
  }
}
}  // namespace

extern "C" int hevcrecon_hub_connect(const char *name, const hevcrecon_seq_params *sp, hevcrecon_hub **out) {
  if (!out || !sp) return 1;
  *out = nullptr;
  const std::string path = shm_name(name);
  int fd = -1;
  for (int tries = 0; tries < 2500 && fd < 0; tries++) {  // the hub may still be starting (it opens the GPU first)
    fd = shm_open(path.c_str(), O_RDWR, 0600);
    if (fd < 0) nap_us(2000);
  }
  if (fd < 0) return 2;
  struct stat st;
  for (int tries = 0; tries < 2500; tries++) {
    if (fstat(fd, &st) != 0) {
      close(fd);
      return 2;
    }
    if ((size_t)st.st_size >= sizeof(HubShared)) break;
    nap_us(2000);
  }
  if ((size_t)st.st_size < sizeof(HubShared)) {
    close(fd);
    return 2;
  }
  void *p = mmap(nullptr, sizeof(HubShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  HubShared *sh = (HubShared *)p;
  for (int tries = 0; tries < 5000 && !sh->ready.load(std::memory_order_acquire); tries++) nap_us(2000);
  if (!sh->ready.load(std::memory_order_acquire) || sh->magic != HR_HUB_MAGIC || sh->version != HR_HUB_VERSION) {
    munmap(p, sizeof(HubShared));
    return 3;
  }
  for (int i = 0; i < HR_HUB_MAX_CLIENTS; i++) {
    uint32_t expect = HR_HUB_FREE;
    HubClient *c = &sh->clients[i];
    // a free slot has all counters at rest (the hub zeroes them before it frees the slot); the parameters are written
    // before the state that publishes them
    if (c->state.load(std::memory_order_acquire) != HR_HUB_FREE) continue;
    // claim with a value no other state uses, fill in, then publish
    if (!c->state.compare_exchange_strong(expect, 0x80000000u, std::memory_order_acq_rel)) continue;
    c->pid = (uint32_t)getpid();
    c->sp = *sp;
    c->state.store(HR_HUB_ATTACHED, std::memory_order_release);
    hevcrecon_hub *h = new hevcrecon_hub();
    h->sh = sh;
    h->me = c;
    h->slot = i;
    *out = h;
    return 0;
  }
  munmap(p, sizeof(HubShared));
  return 4;  // every slot is taken
}

extern "C" const char *hevcrecon_hub_error(const hevcrecon_hub *h) { return h ? h->err.c_str() : "no hub connection (is the hub running?)"; }

extern "C" int hevcrecon_hub_pic_id_base(const hevcrecon_hub *h) { return h ? h->slot * HR_HUB_ID_STRIDE : -1; }

extern "C" int hevcrecon_hub_submit(hevcrecon_hub *h, const hevcrecon_syntax *syn, int dst_pic_id) {
  if (!h || !syn) return 1;
  const int base = h->slot * HR_HUB_ID_STRIDE;
  if (dst_pic_id < base || dst_pic_id >= base + HR_HUB_ID_STRIDE) return fail(h, "picture id outside this connection's range (hevcrecon_hub_pic_id_base() .. + 255)");
  int64_t bytes = 0;
  if (hevcrecon_syntax_export(syn, nullptr, 0, &bytes)) return fail(h, "the picture is not finished");
  uint64_t off = 0;
  if (int rc = reserve(h, (uint64_t)bytes, &off)) return rc;
  HubClient *c = h->me;
  if (hevcrecon_syntax_export(syn, c->blob + off, bytes, &bytes)) return fail(h, "export failed");
  const uint64_t prod = c->produced.load(std::memory_order_relaxed);
  HubEntry &e = c->entries[prod % HR_HUB_ENTRIES];
  e.kind = HR_HUB_PICTURE;
  e.dst = dst_pic_id;
  e.src = -1;
  e.poc = 0;
  e.blob_off = off;
  e.blob_bytes = (uint64_t)bytes;
  h->blob_head = off + (((uint64_t)bytes + 63) & ~(uint64_t)63);
  c->produced.store(prod + 1, std::memory_order_release);
  return 0;
}

extern "C" int hevcrecon_hub_copy(hevcrecon_hub *h, int dst_pic_id, int src_pic_id, int dst_poc) {
  if (!h) return 1;
  const int base = h->slot * HR_HUB_ID_STRIDE;
  if (dst_pic_id < base || dst_pic_id >= base + HR_HUB_ID_STRIDE || src_pic_id < base || src_pic_id >= base + HR_HUB_ID_STRIDE)
    return fail(h, "picture id outside this connection's range");
  uint64_t off = 0;
  if (int rc = reserve(h, 0, &off)) return rc;
  HubClient *c = h->me;
  const uint64_t prod = c->produced.load(std::memory_order_relaxed);
  HubEntry &e = c->entries[prod % HR_HUB_ENTRIES];
  e.kind = HR_HUB_COPY;
  e.dst = dst_pic_id;
  e.src = src_pic_id;
  e.poc = dst_poc;
  e.blob_off = off;
  e.blob_bytes = 0;
  h->blob_head = off;
  c->produced.store(prod + 1, std::memory_order_release);
  return 0;
}

extern "C" int hevcrecon_hub_poll(hevcrecon_hub *h, int *got, uint8_t digest[48]) {
  if (!h || !got || !digest) return 1;
  *got = 0;
  if (int rc = check(h)) return rc;
  HubClient *c = h->me;
  const uint64_t have = c->hashed.load(std::memory_order_acquire), mine = c->collected.load(std::memory_order_relaxed);
  if (mine == have) return 0;
  std::memcpy(digest, c->digests[mine % HR_HUB_DIGESTS], 48);
  c->collected.store(mine + 1, std::memory_order_release);
  *got = 1;
  return 0;
}

extern "C" int hevcrecon_hub_wait(hevcrecon_hub *h, uint8_t digest[48]) {
  for (long spins = 0;; spins++) {
    int got = 0;
    if (int rc = hevcrecon_hub_poll(h, &got, digest)) return rc;
    if (got) return 0;
    nap_us(spins < 100 ? 20 : 200);
    if (spins > 5 * 60 * 5000L) return fail(h, "the hub has not answered for minutes");
  }
}

extern "C" void hevcrecon_hub_disconnect(hevcrecon_hub *h) {
  if (!h) return;
  h->me->state.store(HR_HUB_CLOSING, std::memory_order_release);  // the hub finishes what is queued, frees the pictures and the slot
  munmap((void *)h->sh, sizeof(HubShared));
  delete h;
}
