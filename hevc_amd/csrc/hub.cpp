// hub.cpp — hevcrecon_hub: the process that owns the GPU when many decoder processes share it (protocol and
// rationale: hub_proto.h).  It uses nothing but the engine's public C ABI.
//
//   hevcrecon_hub [--name NAME] [--device N] [--idle-exit SECONDS]
//
// Loop: take from every attached client the requests it has ready -- picture copies at once, and at most one picture
// per client per round, since consecutive pictures of a stream may refer to each other --, import the packed
// pictures into syntax objects of the context that belongs to the clients' sequence parameters, reconstruct them as
// one batch, queue the hash of the batch, and hand digests back as they arrive.  A client's picture ids arrive
// already offset into the client's own range, so streams never meet in the picture buffer.
#include <cerrno>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "../../include/hevcrecon.h"
#include "hub_proto.h"

namespace {
volatile sig_atomic_t g_stop = 0;
void on_signal(int) { g_stop = 1; }
double now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

const int kSyntaxPerClient = 8;  // packed pictures of one client in flight on the device

// A hash kernel runs for tens of milliseconds whatever it hashes (MD5 is one serial chain per plane), and only a few run
// at a time: requests must be few and large.  So a reconstructed picture is first copied aside (device to device, in
// order behind its reconstruction: the stream may overwrite the original with its next picture at once), and the
// copies that have gathered are hashed in one request whenever fewer than kHashRequests are in flight.
const int kHashRequests = 4;
const int kSnapshots = 768;                                       // copies waiting for their hash, per context
const int kSnapshotBase = HR_HUB_MAX_CLIENTS * HR_HUB_ID_STRIDE;  // their picture ids lie behind the clients' ranges

struct Context {  // one engine context per set of sequence parameters
  hevcrecon_ctx *ctx = nullptr;
  hevcrecon_seq_params sp;
  std::vector<int> snap_free;               // picture ids of free copies
  std::vector<std::pair<int, int>> waiting; // (copy, client slot) of pictures not yet in a hash request
  double waiting_since = 0;                 // when the oldest of them was set aside
  int in_flight = 0;                        // hash requests of this context in flight
};
struct Ticket {
  Context *cx;
  int64_t id;
  std::vector<int> client;  // slot of every picture of the request
  std::vector<int> snap;    // the copies it hashes (free again when the digests are back)
  double t_submit = 0;
};
struct ClientState {
  Context *cx = nullptr;
  hevcrecon_syntax *syn[kSyntaxPerClient] = {};
  int next_syn = 0;
  std::deque<std::vector<uint8_t>> ready;  // digests that have arrived but do not fit the client's ring yet
  uint64_t pictures = 0, digests_out = 0;
  bool used[HR_HUB_ID_STRIDE] = {};
};

void refuse(HubClient *c, const std::string &why) {
  std::snprintf(c->error, sizeof(c->error), "%s", why.c_str());
  c->failed.store(1, std::memory_order_release);
  std::fprintf(stderr, "hevcrecon_hub: client %u: %s\n", c->pid, why.c_str());
}
}  // namespace

int main(int argc, char **argv) {
  std::string name = "default";
  int device = 0;
  double idle_exit = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--name" && i + 1 < argc) name = argv[++i];
    else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
    else if (a == "--idle-exit" && i + 1 < argc) idle_exit = atof(argv[++i]);
    else {
      std::fprintf(stderr, "usage: %s [--name NAME] [--device N] [--idle-exit SECONDS]\n", argv[0]);
      return 2;
    }
  }
  std::signal(SIGTERM, on_signal);
  std::signal(SIGINT, on_signal);
  // the GPU first: clients wait for the segment to appear
  hevcrecon_ctx *probe = nullptr;
  if (hevcrecon_create(&probe, device)) {
    std::fprintf(stderr, "hevcrecon_hub: no usable GPU %d\n", device);
    return 1;
  }
  const std::string path = "/hevcrecon_hub_" + name;
  shm_unlink(path.c_str());
  const int fd = shm_open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(HubShared)) != 0) {
    std::perror("hevcrecon_hub: shared memory");
    return 1;
  }
  HubShared *sh = (HubShared *)mmap(nullptr, sizeof(HubShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (sh == (HubShared *)MAP_FAILED) {
    std::perror("hevcrecon_hub: mmap");
    shm_unlink(path.c_str());
    return 1;
  }
  sh->magic = HR_HUB_MAGIC;  // (a fresh segment is all zeros: every slot free, every counter at rest)
  sh->version = HR_HUB_VERSION;
  sh->ready.store(1, std::memory_order_release);

  std::vector<Context *> contexts;
  Context *first = new Context();
  first->ctx = probe;
  bool first_taken = false;
  std::vector<ClientState> cs(HR_HUB_MAX_CLIENTS);
  std::deque<Ticket> tickets;
  double last_active = now();
  bool ever_attached = false;
  uint64_t pictures = 0, batches = 0, hash_requests = 0, rounds = 0;
  double hash_latency = 0;
  double t_import = 0, t_recon = 0, t_copy = 0, t_submit = 0, t_collect = 0, t_sleep = 0;  // where the hub's own time goes
  const double t_start = now();
  int rc_exit = 0;

  auto context_for = [&](const hevcrecon_seq_params &sp) -> Context * {
    for (size_t i = 0; i < contexts.size(); i++)
      if (!std::memcmp(&contexts[i]->sp, &sp, sizeof(sp))) return contexts[i];
    Context *cx = first_taken ? new Context() : first;
    if (first_taken && hevcrecon_create(&cx->ctx, device)) return nullptr;
    first_taken = true;
    cx->sp = sp;
    for (int i = kSnapshots - 1; i >= 0; i--) cx->snap_free.push_back(kSnapshotBase + i);
    if (hevcrecon_sequence_init(cx->ctx, &sp)) {
      std::fprintf(stderr, "hevcrecon_hub: %s\n", hevcrecon_last_error(cx->ctx));
      return nullptr;
    }
    contexts.push_back(cx);
    return cx;
  };

  double last_roll_call = now();
  while (!g_stop && !sh->shutdown.load(std::memory_order_acquire)) {
    bool worked = false, any_attached = false;
    if (now() - last_roll_call > 0.5) {  // a decoder that died without saying so leaves like one that did
      last_roll_call = now();
      for (int i = 0; i < HR_HUB_MAX_CLIENTS; i++) {
        HubClient *c = &sh->clients[i];
        if (c->state.load(std::memory_order_acquire) == HR_HUB_ATTACHED && c->pid && kill((pid_t)c->pid, 0) != 0 && errno == ESRCH)
          c->state.store(HR_HUB_CLOSING, std::memory_order_release);
      }
    }
    // ---- requests ----
    std::map<Context *, std::pair<std::vector<hevcrecon_syntax *>, std::vector<int>>> batch;  // per context: syntax objects, destination ids
    std::map<Context *, std::vector<int>> batch_clients;
    for (int i = 0; i < HR_HUB_MAX_CLIENTS; i++) {
      HubClient *c = &sh->clients[i];
      const uint32_t st = c->state.load(std::memory_order_acquire);
      if (st != HR_HUB_ATTACHED && st != HR_HUB_CLOSING) continue;
      any_attached = ever_attached = true;
      ClientState &k = cs[i];
      if (c->failed.load(std::memory_order_relaxed)) {
        c->consumed.store(c->produced.load(std::memory_order_acquire), std::memory_order_release);  // drop what it still sends
      } else {
        if (!k.cx) {
          k.cx = context_for(c->sp);
          if (!k.cx) {
            refuse(c, "sequence parameters the engine does not take");
            continue;
          }
        }
        bool took_picture = false;
        while (!took_picture) {
          const uint64_t cons = c->consumed.load(std::memory_order_relaxed);
          if (cons == c->produced.load(std::memory_order_acquire)) break;
          const HubEntry e = c->entries[cons % HR_HUB_ENTRIES];
          const int base = i * HR_HUB_ID_STRIDE;
          std::string why;
          if (e.dst < base || e.dst >= base + HR_HUB_ID_STRIDE) why = "picture id outside the client's range";
          else if (e.kind == HR_HUB_COPY) {
            if (e.src < base || e.src >= base + HR_HUB_ID_STRIDE) why = "picture id outside the client's range";
            else if (hevcrecon_picture_copy(k.cx->ctx, e.dst, e.src, e.poc)) why = hevcrecon_last_error(k.cx->ctx);
          } else if (e.kind == HR_HUB_PICTURE) {
            if (e.blob_off > HR_HUB_BLOB_BYTES || e.blob_bytes > HR_HUB_BLOB_BYTES - e.blob_off) why = "request outside the ring";
            else {
              hevcrecon_syntax *&y = k.syn[k.next_syn % kSyntaxPerClient];
              if (!y && hevcrecon_syntax_create(k.cx->ctx, &y)) why = hevcrecon_last_error(k.cx->ctx);
              const double ti = now();
              const int irc = y ? hevcrecon_syntax_import(y, c->blob + e.blob_off, (int64_t)e.blob_bytes) : 1;
              t_import += now() - ti;
              if (!y) {
              } else if (irc) why = hevcrecon_last_error(k.cx->ctx);
              else {
                k.next_syn++;
                batch[k.cx].first.push_back(y);
                batch[k.cx].second.push_back(e.dst);
                batch_clients[k.cx].push_back(i);
                took_picture = true;
              }
            }
          } else {
            why = "unknown request";
          }
          if (!why.empty()) {
            refuse(c, why);
            break;
          }
          k.used[e.dst - base] = true;
          c->consumed.store(cons + 1, std::memory_order_release);
          worked = true;
        }
      }
      // a client that has left: when everything of it has run, its pictures and its slot are free again
      if (st == HR_HUB_CLOSING && c->consumed.load(std::memory_order_relaxed) == c->produced.load(std::memory_order_acquire)) {
        bool pending = false;
        if (k.cx) {  // a picture of it in this round's batch?
          const std::vector<int> &in_round = batch_clients[k.cx];
          for (size_t q = 0; q < in_round.size(); q++) pending = pending || in_round[q] == i;
        }
        if (k.cx)
          for (size_t q = 0; q < k.cx->waiting.size(); q++) pending = pending || k.cx->waiting[q].second == i;
        for (size_t t = 0; t < tickets.size(); t++)
          for (size_t q = 0; q < tickets[t].client.size(); q++) pending = pending || tickets[t].client[q] == i;
        if (!pending) {
          if (k.cx) {
            hevcrecon_synchronize(k.cx->ctx);
            for (int s = 0; s < kSyntaxPerClient; s++)
              if (k.syn[s]) hevcrecon_syntax_destroy(k.syn[s]);
            for (int p = 0; p < HR_HUB_ID_STRIDE; p++)
              if (k.used[p]) hevcrecon_picture_release(k.cx->ctx, i * HR_HUB_ID_STRIDE + p);
          }
          k = ClientState();
          c->produced.store(0, std::memory_order_relaxed);
          c->consumed.store(0, std::memory_order_relaxed);
          c->hashed.store(0, std::memory_order_relaxed);
          c->collected.store(0, std::memory_order_relaxed);
          c->failed.store(0, std::memory_order_relaxed);
          c->state.store(HR_HUB_FREE, std::memory_order_release);
          worked = true;
        }
      }
    }
    // ---- one batch per context, and the hash of it ----
    for (auto it = batch.begin(); it != batch.end(); ++it) {
      Context *cx = it->first;
      std::vector<hevcrecon_syntax *> &syn = it->second.first;
      std::vector<int> &ids = it->second.second;
      std::vector<int> &who = batch_clients[cx];
      if (syn.empty()) continue;
      double tr = now();
      const int brc = hevcrecon_reconstruct_batch(cx->ctx, (int)syn.size(), syn.data(), ids.data());
      t_recon += now() - tr;
      tr = now();
      if (brc) {
        // one picture of the batch is at fault (a reference that is not resident, ...): one by one, the culprit's client hears of it
        for (size_t q = 0; q < syn.size(); q++)
          if (hevcrecon_reconstruct(cx->ctx, syn[q], ids[q])) refuse(&sh->clients[who[q]], hevcrecon_last_error(cx->ctx));
      }
      for (size_t q = 0; q < ids.size(); q++) {
        if (cx->snap_free.empty()) {  // (cannot happen while requests are collected below: every copy belongs to one)
          refuse(&sh->clients[who[q]], "no room for another picture waiting for its hash");
          continue;
        }
        const int sid = cx->snap_free.back();
        if (hevcrecon_picture_copy(cx->ctx, sid, ids[q], 0)) {
          refuse(&sh->clients[who[q]], hevcrecon_last_error(cx->ctx));
          continue;
        }
        cx->snap_free.pop_back();
        if (cx->waiting.empty()) cx->waiting_since = now();
        cx->waiting.push_back(std::make_pair(sid, who[q]));
      }
      t_copy += now() - tr;
      pictures += syn.size();
      batches++;
      sh->pictures.store(pictures, std::memory_order_relaxed);
      sh->batches.store(batches, std::memory_order_relaxed);
    }
    // ---- hash requests: everything that has gathered, whenever a request slot is free ----
    for (size_t ci = 0; ci < contexts.size(); ci++) {
      Context *cx = contexts[ci];
      // at once when nothing is being hashed (a lone stream waits for nothing); else when enough has gathered to be
      // worth a kernel's tens of milliseconds, or the oldest copy has waited that long itself
      const bool must = cx->waiting.size() >= (size_t)kSnapshots / 2;
      const bool worth = cx->in_flight == 0 || cx->waiting.size() >= 32 || now() - cx->waiting_since > 0.02;
      if (cx->waiting.empty() || ((cx->in_flight >= kHashRequests || !worth) && !must)) continue;
      Ticket t;
      t.cx = cx;
      std::vector<int> ids;
      for (size_t q = 0; q < cx->waiting.size(); q++) {
        t.snap.push_back(cx->waiting[q].first);
        t.client.push_back(cx->waiting[q].second);
      }
      const double tsb = now();
      const int src = hevcrecon_picture_md5_submit_in_place(cx->ctx, (int)t.snap.size(), t.snap.data(), &t.id);
      t_submit += now() - tsb;
      if (src) {
        std::fprintf(stderr, "hevcrecon_hub: %s\n", hevcrecon_last_error(cx->ctx));
        rc_exit = 1;
        g_stop = 1;
        break;
      }
      t.t_submit = now();
      cx->waiting.clear();
      cx->in_flight++;
      tickets.push_back(t);
      hash_requests++;
      worked = true;
    }
    // ---- digests ----
    const double tc0 = now();
    while (!tickets.empty()) {
      Ticket &t = tickets.front();
      std::vector<uint8_t> dig(48 * t.client.size());
      int done = 0;
      const bool crowded = tickets.size() >= 48 || t.cx->snap_free.size() < 64;  // the engine keeps 64 requests, the hub so many copies: wait rather than overrun
      const int rc = crowded ? hevcrecon_picture_md5_wait(t.cx->ctx, t.id, dig.data()) : hevcrecon_picture_md5_poll(t.cx->ctx, t.id, &done, dig.data());
      if (rc) {  // a kernel refused a work list: whose it was cannot be told apart, every client of the request hears of it
        for (size_t q = 0; q < t.client.size(); q++) refuse(&sh->clients[t.client[q]], hevcrecon_last_error(t.cx->ctx));
        for (size_t q = 0; q < t.snap.size(); q++) t.cx->snap_free.push_back(t.snap[q]);
        t.cx->in_flight--;
        tickets.pop_front();
        continue;
      }
      if (!crowded && !done) break;
      for (size_t q = 0; q < t.client.size(); q++) cs[t.client[q]].ready.push_back(std::vector<uint8_t>(dig.begin() + 48 * q, dig.begin() + 48 * q + 48));
      for (size_t q = 0; q < t.snap.size(); q++) t.cx->snap_free.push_back(t.snap[q]);
      t.cx->in_flight--;
      hash_latency += now() - t.t_submit;
      tickets.pop_front();
      worked = true;
    }
    t_collect += now() - tc0;
    rounds++;
    for (int i = 0; i < HR_HUB_MAX_CLIENTS; i++) {
      ClientState &k = cs[i];
      HubClient *c = &sh->clients[i];
      while (!k.ready.empty()) {
        const uint64_t h = c->hashed.load(std::memory_order_relaxed);
        if (c->state.load(std::memory_order_acquire) == HR_HUB_ATTACHED && h - c->collected.load(std::memory_order_acquire) >= HR_HUB_DIGESTS) break;
        std::memcpy(c->digests[h % HR_HUB_DIGESTS], k.ready.front().data(), 48);
        c->hashed.store(h + 1, std::memory_order_release);
        k.ready.pop_front();
        worked = true;
      }
    }
    if (worked || any_attached) last_active = now();
    if (!worked) {
      const double tz = now();
      timespec ts = {0, 50 * 1000};
      nanosleep(&ts, nullptr);
      t_sleep += now() - tz;
    }
    if (idle_exit > 0 && ever_attached && !any_attached && tickets.empty() && now() - last_active > idle_exit) break;
  }
  sh->shutdown.store(1, std::memory_order_release);
  std::fprintf(stderr, "hevcrecon_hub: %llu pictures in %llu batches (%.1f per batch), %llu hash requests, %.1f s\n", (unsigned long long)pictures,
               (unsigned long long)batches, batches ? (double)pictures / batches : 0.0, (unsigned long long)hash_requests, now() - t_start);
  std::fprintf(stderr, "hevcrecon_hub: a hash request took %.1f ms on average\n", hash_requests ? hash_latency / hash_requests * 1e3 : 0.0);
  std::fprintf(stderr, "hevcrecon_hub: %llu rounds; seconds in import %.3f, reconstruct_batch %.3f, copies aside %.3f, hash requests %.3f, collecting %.3f, asleep %.3f\n",
               (unsigned long long)rounds, t_import, t_recon, t_copy, t_submit, t_collect, t_sleep);
  for (size_t i = 0; i < cs.size(); i++)
    for (int s = 0; s < kSyntaxPerClient; s++)
      if (cs[i].syn[s]) hevcrecon_syntax_destroy(cs[i].syn[s]);
  for (size_t i = 0; i < contexts.size(); i++) hevcrecon_destroy(contexts[i]->ctx);
  if (!first_taken) hevcrecon_destroy(probe);
  munmap((void *)sh, sizeof(HubShared));
  shm_unlink(path.c_str());
  return rc_exit;
}
