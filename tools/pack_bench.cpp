// pack_bench.cpp — host-side cost of the hand-over, in C++ and without a GPU: reads a raw syntax dump (tools/syn_dump.h)
// and packs every picture with the engine's packer (hevc_amd/csrc/packer.cpp, linked in directly) the way a decoder
// would: begin, one add_ctu per CTU in coding order, finish.  Prints microseconds per picture for the two halves.
//
//   g++ -O2 -std=c++17 -Ihevc_amd/csrc tools/pack_bench.cpp hevc_amd/csrc/packer.cpp -o /tmp/pack_bench
//   /tmp/pack_bench x.syn [repeats [first_picture [pictures]]]
//
// A few pictures packed over and over stay in the caches, as a CTU's arrays are when the parser has just written them.
// (-pg on both sources gives a gprof profile of the packer.)
#include <chrono>
#include "../hevc_amd/csrc/packer.h"
#include "syn_dump.h"

static SeqInfo seq_info(const hevcrecon_seq_params &sp) {
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

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s dump.syn [repeats [first_picture [pictures]]]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 2 ? atoi(argv[2]) : 5;
  const size_t first_pic = argc > 3 ? (size_t)atoi(argv[3]) : 0, max_pics = argc > 4 ? (size_t)atoi(argv[4]) : (size_t)-1;
  syn_dump::Dump d;
  const std::string why = syn_dump::read(argv[1], d);
  if (!why.empty()) {
    fprintf(stderr, "%s\n", why.c_str());
    return 2;
  }
  std::vector<syn_dump::Picture> &pics = d.pics;
  if (first_pic < pics.size()) pics.erase(pics.begin(), pics.begin() + first_pic);
  if (pics.size() > max_pics) pics.resize(max_pics);
  const SeqInfo si = seq_info(d.sp);
  PackedPicture pk;
  double t_add = 0, t_fin = 0, t_begin = 0;
  long packed = 0;
  size_t bytes = 0;
  for (int rep = 0; rep < repeats; rep++)
    for (size_t k = 0; k < pics.size(); k++) {
      const syn_dump::Picture &p = pics[k];
      const std::vector<int> order = syn_dump::coding_order(p);
      auto t0 = std::chrono::steady_clock::now();
      pk.begin(si);
      auto t1 = std::chrono::steady_clock::now();
      for (int i = 0; i < p.nctu; i++) {
        hevcrecon_ctu_syntax cs;
        syn_dump::ctu(p, order[i], cs);
        if (!pk.add_ctu(order[i], &cs)) {
          fprintf(stderr, "picture %zu CTU %d: %s\n", k, order[i], pk.error.c_str());
          return 1;
        }
      }
      auto t2 = std::chrono::steady_clock::now();
      if (!p.slices.empty()) pk.set_slices((int)p.slices.size(), p.slices.data());
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
               (pk.mc_blk.size() + pk.mc_blk8.size()) * 4 + (pk.tu[0].size() + pk.tu[1].size() + pk.tu[2].size() + pk.tu[3].size()) * 12;
    }
  printf("{\"last_picture_regions\": {\"tile64\": %zu, \"tile32\": %zu, \"tile16\": %zu, \"blk8\": %zu, \"blk8_by_unit\": %zu}, ", pk.mc_tile[0].size(), pk.mc_tile[1].size(),
         pk.mc_tile[2].size(), pk.mc_blk8.size(), pk.mc_blk.size());
  printf("\"pictures\": %ld, \"begin_us\": %.1f, \"add_ctu_us\": %.1f, \"finish_us\": %.1f, \"total_us\": %.1f, \"list_bytes_per_picture\": %zu}\n", packed,
         t_begin / packed * 1e6, t_add / packed * 1e6, t_fin / packed * 1e6, (t_begin + t_add + t_fin) / packed * 1e6, bytes / packed);
  return 0;
}
