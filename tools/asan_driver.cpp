// asan_driver.cpp — the host side of the engine (packer.cpp, engine.cpp, hub_client.cpp) and the kernels' CPU
// emulation (kernels.hip against tests/emu/hip_emu.h) in ONE executable built with AddressSanitizer and
// UndefinedBehaviorSanitizer, driven through the public C ABI by raw syntax dumps (tools/syn_dump.h):
//
//   tests/emu/build_emu.sh --asan      ->  tests/emu/asan_driver
//   tests/emu/asan_driver a.syn b.syn ...   [HEVCRECON_ASAN_PICTURES=n: pictures per dump, default 3]
//
// Per picture: begin / add_ctu per CTU in coding order / the picture's scaling lists, weights and slice table / finish /
// reconstruct, the device hash through a ticket, and the digests compared with the reference decoder's (in the dump);
// then the same picture exported, imported into a second object and reconstructed by the entry point that allocates
// nothing into a copy (hevcrecon_picture_copy) of itself, hashed again.  Refused input (a CTU out of order, a damaged
// block) goes through the error paths.  Exit status 0 only if every digest matched and no sanitizer complained.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/hevcrecon.h"
#include "syn_dump.h"

#define CHECK(call)                                                                                 \
  do {                                                                                              \
    if ((call) != 0) {                                                                              \
      fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, hevcrecon_last_error(ctx)); \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

static int run_dump(const char *path, size_t max_pictures) {
  syn_dump::Dump d;
  const std::string why = syn_dump::read(path, d, max_pictures);
  if (!why.empty()) {
    fprintf(stderr, "%s\n", why.c_str());
    return 2;
  }
  hevcrecon_ctx *ctx = nullptr;
  if (hevcrecon_create(&ctx, 0)) return 2;
  CHECK(hevcrecon_sequence_init(ctx, &d.sp));
  hevcrecon_syntax *syn = nullptr, *second = nullptr, *host = nullptr;
  CHECK(hevcrecon_syntax_create(ctx, &syn));
  CHECK(hevcrecon_syntax_create(ctx, &second));
  if (hevcrecon_syntax_create_detached(&d.sp, &host)) return 2;
  const int kScratchCopy = 200;
  int mismatches = 0;
  for (size_t k = 0; k < d.pics.size(); k++) {
    const syn_dump::Picture &p = d.pics[k];
    const std::vector<int> order = syn_dump::coding_order(p);
    for (int pass = 0; pass < 2; pass++) {  // into an object of the context, and into a host-only one
      hevcrecon_syntax *y = pass ? host : syn;
      CHECK(hevcrecon_syntax_begin(y));
      if (p.has_scaling) CHECK(hevcrecon_syntax_set_scaling_lists(y, &p.scaling));
      if (p.has_weights) CHECK(hevcrecon_syntax_set_pred_weights(y, &p.weights));
      for (int i = 0; i < p.nctu; i++) {
        hevcrecon_ctu_syntax cs;
        syn_dump::ctu(p, order[i], cs);
        if (hevcrecon_syntax_add_ctu(y, order[i], &cs)) {
          fprintf(stderr, "%s picture %zu CTU %d: %s\n", path, k, order[i], hevcrecon_syntax_last_error(y));
          return 1;
        }
      }
      if (!p.slices.empty()) CHECK(hevcrecon_syntax_set_slices(y, (int)p.slices.size(), p.slices.data()));
      if (hevcrecon_syntax_finish(y, &p.pp)) {
        fprintf(stderr, "%s picture %zu: %s\n", path, k, hevcrecon_syntax_last_error(y));
        return 1;
      }
    }
    CHECK(hevcrecon_reconstruct(ctx, syn, p.pic_id));
    int64_t ticket = -1;
    CHECK(hevcrecon_picture_md5_submit(ctx, 1, &p.pic_id, &ticket));
    // the packed picture as a block of bytes, into the second object; reconstructed without allocating into a copy
    int64_t n = 0;
    CHECK(hevcrecon_syntax_export(host, nullptr, 0, &n));
    std::vector<unsigned char> blob((size_t)n);
    CHECK(hevcrecon_syntax_export(host, blob.data(), n, &n));
    CHECK(hevcrecon_syntax_import(second, blob.data(), n));
    CHECK(hevcrecon_picture_copy(ctx, kScratchCopy, p.pic_id, p.pp.poc));
    CHECK(hevcrecon_reconstruct_noalloc(ctx, second, kScratchCopy));
    unsigned char got[2][48];
    CHECK(hevcrecon_picture_md5_wait(ctx, ticket, got[0]));
    CHECK(hevcrecon_picture_md5_batch(ctx, 1, &kScratchCopy, got[1]));
    for (int v = 0; v < 2; v++)
      if (memcmp(got[v], p.md5[2], 48)) {
        fprintf(stderr, "%s picture %zu (POC %d): %s differs from the reference decoder's digest\n", path, k, p.pp.poc, v ? "the imported copy" : "the picture");
        mismatches++;
      }
    // error paths: a damaged block, a CTU out of order
    if (k == 0) {
      std::vector<unsigned char> bad(blob);
      bad[bad.size() / 2] ^= 0x5a;
      bad[40] ^= 0xff;
      (void)hevcrecon_syntax_import(host, bad.data(), n);
      (void)hevcrecon_syntax_import(host, bad.data(), n / 2);
      hevcrecon_ctu_syntax cs;
      syn_dump::ctu(p, 0, cs);
      CHECK(hevcrecon_syntax_begin(host));
      if (p.nctu > 1 && hevcrecon_syntax_add_ctu(host, 1, &cs) == 0 && p.tiles.empty()) {
        fprintf(stderr, "%s: a CTU out of order was accepted\n", path);
        return 1;
      }
    }
  }
  CHECK(hevcrecon_synchronize(ctx));
  hevcrecon_syntax_destroy(syn);
  hevcrecon_syntax_destroy(second);
  hevcrecon_syntax_destroy(host);
  hevcrecon_destroy(ctx);
  printf("%s: %zu pictures, %d mismatches\n", path, d.pics.size(), mismatches);
  fflush(stdout);
  return mismatches ? 1 : 0;
}

int main(int argc, char **argv) {
  const size_t n = getenv("HEVCRECON_ASAN_PICTURES") ? (size_t)atoi(getenv("HEVCRECON_ASAN_PICTURES")) : 3;
  int rc = 0;
  for (int i = 1; i < argc; i++) rc |= run_dump(argv[i], n);
  return rc;
}
