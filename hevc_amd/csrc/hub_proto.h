// hub_proto.h — shared-memory protocol between decoder processes and the process that owns the GPU (hub.cpp).
//
// Why there is a hub: the reference decoder is single-threaded with global state, so a host that decodes many
// streams runs one decoder process per stream; a GPU, though, serves only a handful of processes at a time (16
// decoder processes each with a context of their own decode fewer frames per second than 8 do), and pictures of
// different streams share one set of kernel launches only inside one context.  With a hub the decoder processes never
// open the GPU: each packs its pictures on its own core (host-only syntax objects), exports them into its part of a
// shared-memory segment, and one process imports them, batches what the streams have ready into
// hevcrecon_reconstruct_batch calls, hashes the pictures on the device and hands the digests back.
//
// One segment (POSIX shared memory "/hevcrecon_hub_<name>") holds a fixed table of client slots.  Each slot is a
// single-producer / single-consumer pair of rings: requests (packed pictures and picture copies, in decoding order)
// from the client to the hub, digests (one per picture, in the same order) back.  All counters only ever grow.
#ifndef HEVCRECON_HUB_PROTO_H
#define HEVCRECON_HUB_PROTO_H

#include <atomic>
#include <stdint.h>
#include "../../include/hevcrecon.h"

#define HR_HUB_MAGIC 0x42554848u /* "HHUB" */
#define HR_HUB_VERSION 1u
#define HR_HUB_MAX_CLIENTS 32
#define HR_HUB_ENTRIES 32                /* requests a client may have outstanding */
#define HR_HUB_BLOB_BYTES (32u << 20)    /* bytes of packed pictures a client may have outstanding */
#define HR_HUB_DIGESTS 128               /* digests waiting to be collected by a client */
#define HR_HUB_ID_STRIDE 256             /* the engine's picture ids of client slot i are i * 256 + (0 .. 255) */

enum { HR_HUB_FREE = 0, HR_HUB_ATTACHED = 1, HR_HUB_CLOSING = 2 };
enum { HR_HUB_PICTURE = 1, HR_HUB_COPY = 2 };

struct HubEntry {
  uint32_t kind;       /* HR_HUB_PICTURE: reconstruct the packed picture at blob_off into picture dst; HR_HUB_COPY: dst = copy of src */
  int32_t dst, src, poc;
  uint64_t blob_off, blob_bytes;
};

struct HubClient {
  std::atomic<uint32_t> state;
  uint32_t pid;
  hevcrecon_seq_params sp;
  std::atomic<uint64_t> produced;   /* requests written by the client */
  std::atomic<uint64_t> consumed;   /* requests the hub is done with (their bytes in blob[] may be reused) */
  std::atomic<uint64_t> hashed;     /* pictures whose digests the hub has written */
  std::atomic<uint64_t> collected;  /* digests the client has taken */
  std::atomic<uint32_t> failed;     /* the hub refused a request of this client: message in error[] */
  char error[256];
  HubEntry entries[HR_HUB_ENTRIES];
  uint8_t digests[HR_HUB_DIGESTS][48];
  alignas(64) unsigned char blob[HR_HUB_BLOB_BYTES];
};

struct HubShared {
  uint32_t magic, version;
  std::atomic<uint32_t> ready;     /* set by the hub once the table is initialised */
  std::atomic<uint32_t> shutdown;  /* set by whoever wants the hub to leave */
  std::atomic<uint64_t> pictures;  /* pictures reconstructed so far, all clients */
  std::atomic<uint64_t> batches;   /* hevcrecon_reconstruct_batch calls so far */
  HubClient clients[HR_HUB_MAX_CLIENTS];
};

#endif
