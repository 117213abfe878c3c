// pack_bench.cpp — host-side cost of the hand-over, in C++ and without a GPU: reads a raw syntax dump (the ".syn" that
// oracle/hm_bridge.cpp writes in dump mode; `xz -dc x.syn.xz > x.syn` for a fixture), and packs every picture with the
// engine's packer (hevc_amd/csrc/packer.cpp, linked in directly) the way a decoder would: begin, one add_ctu per CTU
// in coding order, finish.  Prints microseconds per picture for the two halves.
//
//   g++ -O2 -std=c++17 -Ihevc_amd/csrc tools/pack_bench.cpp hevc_amd/csrc/packer.cpp -o /tmp/pack_bench
//   /tmp/pack_bench x.syn [repeats]
//
// (-pg on both sources gives a gprof profile of the packer.)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../hevc_amd/csrc/packer.h"

namespace {
const int P = HEVCRECON_PARTS_PER_CTU;

struct Picture {
  int nctu = 0;
  hevcrecon_pic_params pp;
  std::vector<uint8_t> u8[13];  // depth, pred_mode, part_size, tr_idx, cbf0..2, qp, luma_dir, chroma_dir, tskip0..2
  std::vector<int16_t> mv[2];
  std::vector<int8_t> ref[2];
  std::vector<uint8_t> bypass, ipcm;
  std::vector<int32_t> coeff[3];
  std::vector<int32_t> sao;
  std::vector<uint32_t> slice_start;
  std::vector<int32_t> tiles;
};

template <class T> bool get(FILE *f, std::vector<T> &v, size_t n) {
  v.resize(n);
  return fread(v.data(), sizeof(T), n, f) == n;
}

SeqInfo seq_info(const hevcrecon_seq_params &sp) {
  SeqInfo s;
  s.w = sp.width, s.h = sp.height, s.cw = sp.width / 2, s.ch = sp.height / 2;
  s.uw = sp.width / 4, s.uh = sp.height / 4;
  s.ctus_w = (sp.width + 63) / 64, s.ctus_h = (sp.height + 63) / 64;
  s.bd_y = sp.bit_depth_luma, s.bd_c = sp.bit_depth_chroma;
  s.strong_intra = sp.strong_intra_smoothing;
  s.stride_y = (sp.width + 127) & ~127, s.stride_c = (sp.width / 2 + 127) & ~127;
  s.pcm_bd_y = sp.pcm_bit_depth_luma, s.pcm_bd_c = sp.pcm_bit_depth_chroma, s.pcm_lf_disable = sp.pcm_loop_filter_disable;
  return s;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s dump.syn [repeats [first_picture [pictures]]]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 2 ? atoi(argv[2]) : 5;
  // a few pictures packed over and over stay in the caches, as a CTU's arrays are when the parser has just written them
  const size_t first_pic = argc > 3 ? (size_t)atoi(argv[3]) : 0, max_pics = argc > 4 ? (size_t)atoi(argv[4]) : (size_t)-1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  char magic[8];
  hevcrecon_seq_params sp;
  if (fread(magic, 1, 8, f) != 8 || (memcmp(magic, "HVSYN2\0\0", 8) && memcmp(magic, "HVSYN1\0\0", 8)) || fread(&sp, sizeof(sp), 1, f) != 1) {
    fprintf(stderr, "%s: not a raw syntax dump (decompress .xz first)\n", argv[1]);
    return 2;
  }
  // dumps of format 1 hold shorter picture parameters (up to and including the reference POCs)
  const size_t pp_size = magic[5] == '2' ? sizeof(hevcrecon_pic_params) : (13 + 2 * HEVCRECON_MAX_REFS * 2) * 4;
  std::vector<Picture> pics;
  std::vector<uint32_t> pending_slices;
  std::vector<int32_t> pending_tiles;
  for (;;) {
    int32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) break;
    const int32_t tag = hdr[0], n = hdr[1];
    if (tag == 0x4C414353) { fseek(f, sizeof(hevcrecon_scaling_lists), SEEK_CUR); continue; }
    if (tag == 0x44525057) { fseek(f, sizeof(hevcrecon_pred_weights), SEEK_CUR); continue; }
    if (tag == 0x54534C53) { get(f, pending_slices, (size_t)n * P); continue; }
    if (tag == 0x454C4954) { get(f, pending_tiles, (size_t)n * 2); continue; }
    if (tag == 0x52504C53) { fseek(f, (long)n * sizeof(hevcrecon_slice_params), SEEK_CUR); continue; }
    if (tag == 0x534D4350) { fseek(f, (long)n * (4 + (16 + 4 + 4) * P * 2), SEEK_CUR); continue; }
    if (tag != 0x54434950) {
      fprintf(stderr, "bad record tag %08x\n", tag);
      return 2;
    }
    pics.emplace_back();
    Picture &p = pics.back();
    p.nctu = n;
    p.slice_start.swap(pending_slices);
    p.tiles.swap(pending_tiles);
    memset(&p.pp, 0, sizeof(p.pp));
    bool ok = fread(&p.pp, pp_size, 1, f) == 1;
    for (int i = 0; i < 13 && ok; i++) ok = get(f, p.u8[i], (size_t)n * P);
    for (int l = 0; l < 2 && ok; l++) ok = get(f, p.mv[l], (size_t)n * P * 2);
    for (int l = 0; l < 2 && ok; l++) ok = get(f, p.ref[l], (size_t)n * P);
    ok = ok && get(f, p.bypass, (size_t)n * P) && get(f, p.ipcm, (size_t)n * P);
    for (int c = 0; c < 3 && ok; c++) {
      std::vector<int16_t> c16;
      ok = get(f, c16, (size_t)n * P * (c ? 4 : 16));
      p.coeff[c].assign(c16.begin(), c16.end());  // the decoder holds them as 32-bit values
    }
    ok = ok && get(f, p.sao, (size_t)n * 18);
    fseek(f, 144, SEEK_CUR);
    if (hdr[3]) fseek(f, 3L * (sp.width * sp.height * 3 / 2) * 2, SEEK_CUR);
    if (!ok) {
      fprintf(stderr, "truncated dump\n");
      return 2;
    }
  }
  fclose(f);
  if (first_pic < pics.size()) pics.erase(pics.begin(), pics.begin() + first_pic);
  if (pics.size() > max_pics) pics.resize(max_pics);
  const SeqInfo si = seq_info(sp);
  PackedPicture pk;
  double t_add = 0, t_fin = 0, t_begin = 0;
  long packed = 0;
  size_t bytes = 0;
  for (int rep = 0; rep < repeats; rep++)
    for (size_t k = 0; k < pics.size(); k++) {
      const Picture &p = pics[k];
      std::vector<int> order(p.nctu);
      for (int a = 0; a < p.nctu; a++) order[p.tiles.empty() ? a : p.tiles[2 * a] - 1] = a;
      auto t0 = std::chrono::steady_clock::now();
      pk.begin(si);
      auto t1 = std::chrono::steady_clock::now();
      for (int i = 0; i < p.nctu; i++) {
        const int a = order[i];
        hevcrecon_ctu_syntax cs;
        memset(&cs, 0, sizeof(cs));
        const size_t o = (size_t)a * P;
        cs.depth = p.u8[0].data() + o;
        cs.pred_mode = (const int8_t *)p.u8[1].data() + o;
        cs.part_size = (const int8_t *)p.u8[2].data() + o;
        cs.tr_idx = p.u8[3].data() + o;
        for (int c = 0; c < 3; c++) {
          cs.cbf[c] = p.u8[4 + c].data() + o;
          cs.transform_skip[c] = p.u8[10 + c].data() + o;
          cs.coeff[c] = p.coeff[c].data() + o * (c ? 4 : 16);
          cs.sao_type[c] = p.sao[(size_t)a * 18 + c * 6];
          cs.sao_subtype[c] = p.sao[(size_t)a * 18 + c * 6 + 1];
          for (int j = 0; j < 4; j++) cs.sao_offset[c][j] = p.sao[(size_t)a * 18 + c * 6 + 2 + j];
        }
        cs.qp = (const int8_t *)p.u8[7].data() + o;
        cs.luma_dir = p.u8[8].data() + o;
        cs.chroma_dir = p.u8[9].data() + o;
        for (int l = 0; l < 2; l++) {
          cs.mv[l] = p.mv[l].data() + 2 * o;
          cs.ref_idx[l] = p.ref[l].data() + o;
        }
        cs.transquant_bypass = p.bypass.data() + o;
        cs.ipcm = p.ipcm.data() + o;
        cs.slice_start = p.slice_start.empty() ? nullptr : p.slice_start.data() + o;
        if (!p.tiles.empty()) {
          cs.coding_addr_plus1 = p.tiles[2 * a];
          cs.tile_id = p.tiles[2 * a + 1];
        }
        if (!pk.add_ctu(a, &cs)) {
          fprintf(stderr, "picture %zu CTU %d: %s\n", k, a, pk.error.c_str());
          return 1;
        }
      }
      auto t2 = std::chrono::steady_clock::now();
      if (!pk.finish(&p.pp)) {
        fprintf(stderr, "picture %zu: %s\n", k, pk.error.c_str());
        return 1;
      }
      auto t3 = std::chrono::steady_clock::now();
      t_begin += std::chrono::duration<double>(t1 - t0).count();
      t_add += std::chrono::duration<double>(t2 - t1).count();
      t_fin += std::chrono::duration<double>(t3 - t2).count();
      packed++;
      bytes += pk.runs.size() * 16 + pk.coeff.size() * 4 + pk.intra.size() * 16 + (pk.mc_tile[0].size() + pk.mc_tile[1].size() + pk.mc_tile[2].size()) * 16 +
               pk.mc_blk.size() * 4 + (pk.tu[0].size() + pk.tu[1].size() + pk.tu[2].size() + pk.tu[3].size()) * 12;
    }
  printf("{\"pictures\": %ld, \"begin_us\": %.1f, \"add_ctu_us\": %.1f, \"finish_us\": %.1f, \"total_us\": %.1f, \"list_bytes_per_picture\": %zu}\n", packed,
         t_begin / packed * 1e6, t_add / packed * 1e6, t_fin / packed * 1e6, (t_begin + t_add + t_fin) / packed * 1e6, bytes / packed);
  return 0;
}
