// syn_dump.h — reader of raw syntax dumps (the ".syn" that oracle/hm_bridge.cpp writes in dump mode; `xz -dc x.syn.xz`
// gives one from a fixture) for the C++ tools: every picture's per-CTU arrays as the decoder held them, the
// reference decoder's digests of the final picture, and a helper that points a hevcrecon_ctu_syntax at one CTU.
#ifndef HEVCRECON_TOOLS_SYN_DUMP_H
#define HEVCRECON_TOOLS_SYN_DUMP_H
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/hevcrecon.h"

namespace syn_dump {
const int P = HEVCRECON_PARTS_PER_CTU;

struct Picture {
  int nctu = 0, pic_id = 0;
  hevcrecon_pic_params pp;
  std::vector<uint8_t> u8[13];  // depth, pred_mode, part_size, tr_idx, cbf0..2, qp, luma_dir, chroma_dir, tskip0..2
  std::vector<int16_t> mv[2];
  std::vector<int8_t> ref[2];
  std::vector<uint8_t> bypass, ipcm;
  std::vector<int32_t> coeff[3];
  std::vector<int32_t> sao;
  std::vector<uint32_t> slice_start;
  std::vector<int32_t> tiles;
  std::vector<hevcrecon_slice_params> slices;
  bool has_scaling = false, has_weights = false;
  hevcrecon_scaling_lists scaling;
  hevcrecon_pred_weights weights;
  std::vector<std::vector<int16_t>> pcm[3];  // per CTU (empty: none)
  unsigned char md5[3][3][16];               // [stage][component]
};

struct Dump {
  hevcrecon_seq_params sp;
  std::vector<Picture> pics;
};

template <class T> inline bool get(FILE *f, std::vector<T> &v, size_t n) {
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

// Reads a whole dump; returns an empty string, or what is wrong with the file.
inline std::string read(const char *path, Dump &d, size_t max_pictures = (size_t)-1) {
  FILE *f = fopen(path, "rb");
  if (!f) return std::string("cannot open ") + path;
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || (memcmp(magic, "HVSYN2\0\0", 8) && memcmp(magic, "HVSYN1\0\0", 8)) || fread(&d.sp, sizeof(d.sp), 1, f) != 1) {
    fclose(f);
    return std::string(path) + ": not a raw syntax dump (decompress .xz first)";
  }
  // dumps of format 1 hold shorter picture parameters (up to and including the reference POCs)
  const size_t pp_size = magic[5] == '2' ? sizeof(hevcrecon_pic_params) : (13 + 2 * HEVCRECON_MAX_REFS * 2) * 4;
  Picture next;
  while (d.pics.size() < max_pictures) {
    int32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) break;
    const int32_t tag = hdr[0], n = hdr[1];
    bool ok = true;
    if (tag == 0x4C414353) { ok = fread(&next.scaling, sizeof(next.scaling), 1, f) == 1; next.has_scaling = true; }
    else if (tag == 0x44525057) { ok = fread(&next.weights, sizeof(next.weights), 1, f) == 1; next.has_weights = true; }
    else if (tag == 0x54534C53) ok = get(f, next.slice_start, (size_t)n * P);
    else if (tag == 0x454C4954) ok = get(f, next.tiles, (size_t)n * 2);
    else if (tag == 0x52504C53) ok = get(f, next.slices, (size_t)n);
    else if (tag == 0x534D4350) {
      for (int i = 0; i < n && ok; i++) {
        int32_t addr;
        ok = fread(&addr, 4, 1, f) == 1;
        for (int c = 0; c < 3 && ok; c++) {
          if ((int)next.pcm[c].size() <= addr) next.pcm[c].resize(addr + 1);
          ok = get(f, next.pcm[c][addr], (size_t)(c ? 4 : 16) * P);
        }
      }
    } else if (tag == 0x54434950) {
      Picture &p = next;
      p.nctu = n;
      p.pic_id = hdr[2];
      memset(&p.pp, 0, sizeof(p.pp));
      ok = fread(&p.pp, pp_size, 1, f) == 1;
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
      ok = ok && fread(p.md5, 1, 144, f) == 144;
      if (ok && hdr[3]) fseek(f, 3L * (d.sp.width * d.sp.height * 3 / 2) * 2, SEEK_CUR);
      if (ok) {
        d.pics.push_back(p);
        next = Picture();
      }
    } else {
      fclose(f);
      return "bad record tag";
    }
    if (!ok) {
      fclose(f);
      return "truncated dump";
    }
  }
  fclose(f);
  return "";
}

// CTU addresses of a picture in coding order (tile scan when the picture has tiles)
inline std::vector<int> coding_order(const Picture &p) {
  std::vector<int> order(p.nctu);
  for (int a = 0; a < p.nctu; a++) order[p.tiles.empty() ? a : p.tiles[2 * a] - 1] = a;
  return order;
}

inline void ctu(const Picture &p, int a, hevcrecon_ctu_syntax &cs) {
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
    cs.pcm[c] = (int)p.pcm[c].size() > a && !p.pcm[c][a].empty() ? p.pcm[c][a].data() : nullptr;
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
}
}  // namespace syn_dump
#endif
