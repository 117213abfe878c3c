// packer.cpp — see packer.h.
//
// The walk below mirrors what the reference does per CTU when it reconstructs
// (TDecCu::xDecompressCU, TLibDecoder/TDecCu.cpp:404-466; TComTrQuant::invRecurTransformNxN,
// TLibCommon/TComTrQuant.cpp:1292-1335; TDecCu::xIntraLumaRecQT / xIntraChromaRecQT,
// TLibDecoder/TDecCu.cpp:686-743) but instead of computing samples it only records WHAT has to be
// computed, so that the GPU can do all blocks of a kind at once.
#include "packer.h"
#include <algorithm>
#include <cstring>
#include <mutex>

namespace {

// z-scan <-> raster inside a 64x64 CTU of 4x4 partitions (x is the low bit of each pair).
struct ZScan {
  uint8_t x[256], y[256], z[16][16];
  ZScan() {
    for (int i = 0; i < 256; i++) {
      int xx = 0, yy = 0;
      for (int b = 0; b < 4; b++) {
        xx |= ((i >> (2 * b)) & 1) << b;
        yy |= ((i >> (2 * b + 1)) & 1) << b;
      }
      x[i] = (uint8_t)xx;
      y[i] = (uint8_t)yy;
      z[yy][xx] = (uint8_t)i;
    }
  }
};
const ZScan kZ;

// Working records of pictures being packed (2 MB at 1080p): handed out by begin(), given back by finish().  A decoder
// packs one picture at a time and a host that keeps thousands of packed pictures does not keep thousands of these.
struct RecPool {
  std::mutex m;
  std::vector<std::vector<Part4>> free_list;
  void take(std::vector<Part4> &v, size_t n) {
    {
      std::lock_guard<std::mutex> lock(m);
      for (size_t i = 0; i < free_list.size(); i++)
        if (free_list[i].size() == n) {
          v.swap(free_list[i]);
          free_list.erase(free_list.begin() + i);
          return;
        }
    }
    v.assign(n, Part4());
  }
  void give(std::vector<Part4> &v) {
    if (v.empty()) return;
    std::lock_guard<std::mutex> lock(m);
    if (free_list.size() < 8) {
      free_list.emplace_back();
      free_list.back().swap(v);
    }
    std::vector<Part4>().swap(v);
  }
};
RecPool g_rec_pool;

inline bool same_record(const Part4 &a, const Part4 &b) {
  uint64_t x[2], y[2];
  std::memcpy(x, &a, 16);
  std::memcpy(y, &b, 16);
  return x[0] == y[0] && x[1] == y[1];
}

inline int16_t sat16(int32_t v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline int16_t sat16(int16_t v) { return v; }
inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}

}  // namespace

void PackedPicture::begin(const SeqInfo &s) {
  seq = s;
  // per CTU, z-order inside (the order the CTUs' partitions arrive in): every record inside the picture is written
  // by add_ctu before anything reads it; CTUs that reach over the picture's edge are cleared there first
  if (rec4.size() != (size_t)s.ctus_w * s.ctus_h * 256) {
    g_rec_pool.give(rec4);
    g_rec_pool.take(rec4, (size_t)s.ctus_w * s.ctus_h * 256);
  }
  for (int i = 0; i < 4; i++) tu[i].clear();
  coeff_len = 0;
  coded_coeffs = 0;
  intra.clear();
  intra_nb.clear();
  ictu.clear();
  ictu_wave.clear();
  lvl_counts.clear();
  rounds.clear();
  wave_off.clear();
  ctu_wave.assign((size_t)s.ctus_w * s.ctus_h, 0);
  sao.assign((size_t)s.ctus_w * s.ctus_h * 3, SaoCtu());
  use_scaling = false;
  use_wp = false;
  has_bypass = false;
  has_pcm = false;
  n_rec4 = (int64_t)s.uw * s.uh;
  n_rec4_intra = 0;
  mc_ref_samples = 0;
  max_ref_idx[0] = max_ref_idx[1] = -1;
  inter_without_ref = false;
  next_ctu = 0;
  ctu_order.assign((size_t)s.ctus_w * s.ctus_h, -1);
  ctu_tile.assign((size_t)s.ctus_w * s.ctus_h, 0);
  slices.clear();
  ctu_in_two_slices = false;
  ctu_slice.assign((size_t)s.ctus_w * s.ctus_h, 0);
  ctu_info.clear();
  finished = false;
  error.clear();
}

void PackedPicture::release_work() { g_rec_pool.give(rec4); }

bool PackedPicture::set_slices(int n, const hevcrecon_slice_params *sl) {
  if (finished) {
    error = "the slice table must be set between hevcrecon_syntax_begin and hevcrecon_syntax_finish";
    return false;
  }
  if (n < 0 || (n > 0 && !sl)) {
    error = "bad slice table";
    return false;
  }
  for (int i = 0; i < n; i++)
    if (sl[i].beta_offset_div2 < -6 || sl[i].beta_offset_div2 > 6 || sl[i].tc_offset_div2 < -6 || sl[i].tc_offset_div2 > 6 ||
        sl[i].cb_qp_delta_slice < -12 || sl[i].cb_qp_delta_slice > 12 || sl[i].cr_qp_delta_slice < -12 || sl[i].cr_qp_delta_slice > 12) {
      error = "slice parameter out of range";
      return false;
    }
  slices.assign(sl, sl + n);
  return true;
}

const hevcrecon_slice_params *PackedPicture::slice_at(uint32_t start) const {
  for (size_t i = 0; i < slices.size(); i++)
    if (slices[i].start_addr == start) return &slices[i];
  return nullptr;
}

// What the in-loop filter kernels need to know per CTU when slices carry their own parameters or boundaries are not
// crossed.  Deblocking: TComLoopFilter::xSetLoopfilterParam (TLibCommon/TComLoopFilter.cpp:363-419) over
// TComDataCU::getPULeft / getPUAbove (TLibCommon/TComDataCU.cpp:1141-1238): the edge towards a neighbouring CTU is
// filtered unless the CTU's slice does not filter across slice boundaries and the neighbour precedes that slice, or
// tiles are not crossed and the neighbour belongs to another tile; parameters are those of the CTU's slice.
// SAO: TComDataCU::setNDBFilterBlockBorderAvailability (TLibCommon/TComDataCU.cpp:3813-4166, one block per CTU) with
// the tile sides of TComPic::createNonDBFilterInfo (TLibCommon/TComPic.cpp:301-327): a neighbouring block is usable
// when it lies in the picture and either belongs to the same slice or the later of the two slices filters across its
// boundaries, and no uncrossed tile boundary lies on the sides of the CTU it is reached over.
bool PackedPicture::build_ctu_info() {
  ctu_info.clear();
  if (!slices.empty()) {
    if (ctu_in_two_slices) {
      error = "a slice table needs slices that end at CTU boundaries";
      return false;
    }
    for (size_t a = 0; a < ctu_slice.size(); a++)
      if (!slice_at(ctu_slice[a])) {
        error = "CTU of a slice that is not in the slice table";
        return false;
      }
  }
  if (slices.empty() && !pp.no_loop_filter_across_tiles) return true;
  const int W = seq.ctus_w, H = seq.ctus_h;
  ctu_info.resize((size_t)W * H);
  hevcrecon_slice_params dflt;
  std::memset(&dflt, 0, sizeof(dflt));
  dflt.deblocking_disable = pp.deblocking_disable;
  dflt.beta_offset_div2 = pp.beta_offset_div2;
  dflt.tc_offset_div2 = pp.tc_offset_div2;
  dflt.cb_qp_delta_slice = pp.cb_qp_delta_slice;
  dflt.cr_qp_delta_slice = pp.cr_qp_delta_slice;
  dflt.loop_filter_across_slices = 1;
  auto slice_of = [&](int a) -> const hevcrecon_slice_params & {
    const hevcrecon_slice_params *sl = slices.empty() ? nullptr : slice_at(ctu_slice[a]);
    return sl ? *sl : dflt;
  };
  const bool tiles_apart = pp.no_loop_filter_across_tiles != 0;
  for (int a = 0; a < W * H; a++) {
    const int cx = a % W, cy = a / W;
    const hevcrecon_slice_params &me = slice_of(a);
    CtuInfo ci;
    ci.beta_off_div2 = (int8_t)me.beta_offset_div2;
    ci.tc_off_div2 = (int8_t)me.tc_offset_div2;
    ci.flags = me.deblocking_disable ? CI_DEBLOCK_OFF : 0;
    auto edge_ok = [&](int n) {
      if (!me.loop_filter_across_slices && (uint32_t)ctu_order[n] * 256u + 255u < ctu_slice[a]) return false;
      if (tiles_apart && ctu_tile[n] != ctu_tile[a]) return false;
      return true;
    };
    if (cx > 0 && edge_ok(a - 1)) ci.flags |= CI_EDGE_LEFT;
    if (cy > 0 && edge_ok(a - W)) ci.flags |= CI_EDGE_TOP;
    static const int kOx[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, kOy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
    ci.sao_avail = 0;
    for (int b = 0; b < 8; b++) {
      const int nx = cx + kOx[b], ny = cy + kOy[b];
      if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
      const int n = ny * W + nx;
      if (ctu_slice[n] != ctu_slice[a]) {
        const hevcrecon_slice_params &later = ctu_slice[n] > ctu_slice[a] ? slice_of(n) : me;
        if (!later.loop_filter_across_slices) continue;
      }
      if (tiles_apart && ((kOx[b] && ctu_tile[a + kOx[b]] != ctu_tile[a]) || (kOy[b] && ctu_tile[a + kOy[b] * W] != ctu_tile[a]))) continue;
      ci.sao_avail |= (uint8_t)(1 << b);
    }
    ctu_info[a] = ci;
  }
  // De-quantisation adds the picture's chroma offsets (PPS offset + the slice delta of the picture parameters) to the
  // QP a transform block carries: blocks of a slice with other deltas carry the difference.
  if (!slices.empty())
    for (int i = 0; i < 4; i++)
      for (size_t k = 0; k < tu[i].size(); k++) {
        TuItem &t = tu[i][k];
        if (!t.comp) continue;
        const int a = (t.y >> 5) * W + (t.x >> 5);
        const hevcrecon_slice_params &sl = slice_of(a);
        t.qp = (int8_t)(t.qp + (t.comp == 1 ? sl.cb_qp_delta_slice - pp.cb_qp_delta_slice : sl.cr_qp_delta_slice - pp.cr_qp_delta_slice));
      }
  return true;
}

// Position in coding order of the 4x4 unit (ux, uy): CTUs in the order they were added (raster, or tile scan),
// z-order inside.  Units of CTUs that have not been added yet come after everything.
const Part4 &PackedPicture::unit(int ux, int uy) const {
  return rec4[((size_t)(uy >> 4) * seq.ctus_w + (ux >> 4)) * 256 + kZ.z[uy & 15][ux & 15]];
}

int PackedPicture::coding_order(int ux, int uy) const {
  const int o = ctu_order[(size_t)(uy >> 4) * seq.ctus_w + (ux >> 4)];
  return o < 0 ? 0x7fffffff : o * 256 + kZ.z[uy & 15][ux & 15];
}

bool PackedPicture::add_ctu(int addr, const hevcrecon_ctu_syntax *ctu) {
  if (!error.empty()) return false;  // a refused CTU spoils the picture: begin again
  if (finished || addr < 0 || addr >= seq.ctus_w * seq.ctus_h || ctu_order[addr] >= 0 ||
      (ctu->coding_addr_plus1 ? ctu->coding_addr_plus1 - 1 : addr) != next_ctu) {
    error = "CTUs must be added once each, in coding order";
    return false;
  }
  ctu_order[addr] = next_ctu++;
  ctu_tile[addr] = ctu->tile_id;
  ctu_slice[addr] = ctu->slice_start ? ctu->slice_start[0] : 0;
  if (ctu->slice_start) {
    // per-slice parameters are kept per CTU: should a slice table arrive (any time before the picture closes), every
    // partition of the CTU inside the picture has to belong to the same slice
    const int cx0 = (addr % seq.ctus_w) * 16, cy0 = (addr / seq.ctus_w) * 16;
    for (int z = 1; z < 256; z++)
      if (ctu->slice_start[z] != ctu_slice[addr] && cx0 + kZ.x[z] < seq.uw && cy0 + kZ.y[z] < seq.uh) {
        ctu_in_two_slices = true;
        break;
      }
  }
  for (int c = 0; c < 3; c++) {
    // what the SAO kernel's offset table holds in a byte: types -1..4, offsets of at most 31 in magnitude (the
    // syntax allows (1 << (min(bit depth, 10) - 5)) - 1)
    bool ok = ctu->sao_type[c] >= -1 && ctu->sao_type[c] <= 4;
    for (int i = 0; i < 4 && ok && ctu->sao_type[c] >= 0; i++) ok = ctu->sao_offset[c][i] >= -31 && ctu->sao_offset[c][i] <= 31;
    if (!ok) {
      error = "SAO type or offset out of range";
      return false;
    }
    SaoCtu &s = sao[(size_t)addr * 3 + c];
    s.type = (int8_t)ctu->sao_type[c];
    s.band = (int8_t)ctu->sao_subtype[c];
    for (int i = 0; i < 4; i++) s.offset[i] = (int8_t)ctu->sao_offset[c][i];
  }
  ctu_items.clear();
  ctu_item_nb.clear();
  ctu_item_lvl.clear();
  ctu_dep_wave = 0;
  ctu_dep_mask = 0;
  const int cx = addr % seq.ctus_w, cy = addr / seq.ctus_w;
  cur_rec = &rec4[(size_t)addr * 256];
  if (cx * 64 + 64 > seq.w || cy * 64 + 64 > seq.h) std::memset(cur_rec, 0, 256 * sizeof(Part4));  // units outside the picture stay invalid
  ctu_has_intra = false;
  walk_cu(ctu, cx * 64, cy * 64, 0, 0);
  if (!error.empty()) return false;
  if (!ctu_items.empty()) {
    // blocks of one local level are independent of each other: order the CTU's blocks by level
    int nlev = 0;
    for (size_t i = 0; i < ctu_item_lvl.size(); i++) nlev = std::max<int>(nlev, ctu_item_lvl[i]);
    if (nlev > HR_INTRA_MAX_LEVELS || ctu_items.size() > HR_INTRA_MAX_ITEMS) {
      error = "CTU holds more intra blocks or dependency levels than a 64x64 CTU can";
      return false;
    }
    std::vector<int> start(nlev + 2, 0);
    for (size_t i = 0; i < ctu_item_lvl.size(); i++) start[ctu_item_lvl[i] + 1]++;
    IntraCtu ic;
    ic.item_start = (uint32_t)intra.size();
    ic.lvl_start = (uint32_t)lvl_counts.size();
    ic.n_items = (uint16_t)ctu_items.size();
    ic.n_lvl = (uint16_t)nlev;
    ic.cx = (uint8_t)cx;
    ic.cy = (uint8_t)cy;
    ic.deps = (uint8_t)ctu_dep_mask;
    ic.pad = 0;
    ic.pad2 = 0;
    for (int l = 1; l <= nlev; l++) lvl_counts.push_back((uint16_t)start[l + 1]);
    for (int l = 1; l <= nlev + 1; l++) start[l] += start[l - 1];
    const std::vector<int> first(start);  // first[l] = first block of level l, relative to the CTU's first
    intra.resize(ic.item_start + ctu_items.size());
    intra_nb.resize(intra.size());
    for (size_t i = 0; i < ctu_items.size(); i++) {
      const size_t at = ic.item_start + start[ctu_item_lvl[i]]++;
      intra[at] = ctu_items[i];
      intra_nb[at] = ctu_item_nb[i];
    }
    // rounds: the blocks of a level, HR_INTRA_ROUND_BLOCKS at a time
    ic.round_start = (uint32_t)rounds.size();
    for (int l = 1; l <= nlev; l++)
      for (int b = first[l]; b < first[l + 1]; b += HR_INTRA_ROUND_BLOCKS) {
        const int nb = std::min<int>(HR_INTRA_ROUND_BLOCKS, first[l + 1] - b);
        rounds.push_back((uint16_t)(b | (nb << IR_COUNT_SHIFT)));
      }
    if (rounds.size() - ic.round_start > HR_INTRA_MAX_ROUNDS) {
      error = "CTU holds more intra blocks or dependency levels than a 64x64 CTU can";
      return false;
    }
    ic.n_rounds = (uint16_t)(rounds.size() - ic.round_start);
    // wave: after every neighbouring CTU whose intra blocks this CTU's blocks actually read (found while the
    // blocks' neighbour masks were built)
    const int wave = ctu_dep_wave;
    ctu_wave[addr] = (uint16_t)(wave + 1);
    ictu.push_back(ic);
    ictu_wave.push_back((uint16_t)(wave + 1));
  }
  return error.empty();
}

void PackedPicture::walk_cu(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int depth) {
  const int size = 64 >> depth;
  const int x = ctux + 4 * kZ.x[z], y = ctuy + 4 * kZ.y[z];
  const bool boundary = x + size > seq.w || y + size > seq.h;
  if ((depth < ctu->depth[z] && depth < 3) || boundary) {
    if (depth >= 3) {
      error = "coding unit crosses the picture boundary";
      return;
    }
    const int np = 256 >> (2 * (depth + 1));
    for (int q = 0; q < 4; q++) {
      const int zq = z + q * np;
      if (ctux + 4 * kZ.x[zq] < seq.w && ctuy + 4 * kZ.y[zq] < seq.h) walk_cu(ctu, ctux, ctuy, zq, depth + 1);
    }
    return;
  }
  const int np = (size >> 2) * (size >> 2);
  const bool intra = ctu->pred_mode[z] == 1;
  // values that index tables on the device are checked on the way (a parser bug must not become a stray access):
  // the QP range of TComDataCU::setQPSubParts is -QpBDOffset .. 51
  const int qp_min = -6 * (seq.bd_y - 8);
  bool qp_ok = true;
  for (int zz = z; zz < z + np; zz++) qp_ok = qp_ok && ctu->qp[zz] >= qp_min && ctu->qp[zz] <= 51;
  if (!qp_ok) {
    error = "QP out of range";
    return;
  }
  const bool pcm = intra && ctu->ipcm && ctu->ipcm[z];
  // TComLoopFilter::xEdgeFilterLuma (TLibCommon/TComLoopFilter.cpp:633-641) and xPCMCURestoration
  // (TLibCommon/TComSampleAdaptiveOffset.cpp:1435): lossless units always, IPCM units when the SPS says so
  const bool unfiltered = (ctu->transquant_bypass && ctu->transquant_bypass[z]) || (pcm && seq.pcm_lf_disable);
  const uint8_t cu_flags = (uint8_t)((unfiltered ? P4_BYPASS : 0) | (pcm ? P4_IPCM : 0));
  const uint8_t *cu_depth = ctu->depth, *tr_idx = ctu->tr_idx, *cbf_y = ctu->cbf[0];
  const int8_t *pred_mode = ctu->pred_mode, *part_size = ctu->part_size, *qp = ctu->qp, *ref0 = ctu->ref_idx[0], *ref1 = ctu->ref_idx[1];
  const int16_t *mv0 = ctu->mv[0], *mv1 = ctu->mv[1];
  Part4 *out = cur_rec;
  int nref_units = 0, max0 = max_ref_idx[0], max1 = max_ref_idx[1];
  for (int zz = z; zz < z + np; zz++) {
    Part4 p;
    p.flags = (uint8_t)((cu_depth[zz] & P4_DEPTH_MASK) | (pred_mode[zz] == 1 ? P4_INTRA : 0) | ((part_size[zz] & 7) << P4_PARTSIZE_SHIFT) | cu_flags);
    p.tu = (uint8_t)((tr_idx[zz] & P4_TRIDX_MASK) | (((cbf_y[zz] >> tr_idx[zz]) & 1) ? P4_CBF_Y : 0) | P4_VALID);
    p.qp = qp[zz];
    p.pad = 0;
    p.ref_idx[0] = ref0[zz];
    p.ref_idx[1] = ref1[zz];
    p.pad2[0] = p.pad2[1] = 0;
    p.mv[0][0] = mv0[2 * zz];
    p.mv[0][1] = mv0[2 * zz + 1];
    p.mv[1][0] = mv1[2 * zz];
    p.mv[1][1] = mv1[2 * zz + 1];
    out[zz] = p;
    if (!intra) {
      const int nref = (p.ref_idx[0] >= 0) + (p.ref_idx[1] >= 0);
      nref_units += nref;
      if (nref == 0) inter_without_ref = true;
      max0 = p.ref_idx[0] > max0 ? p.ref_idx[0] : max0;
      max1 = p.ref_idx[1] > max1 ? p.ref_idx[1] : max1;
    }
  }
  mc_ref_samples += 24 * (int64_t)nref_units;  // 16 luma + 2 x 4 chroma samples per reference read
  max_ref_idx[0] = max0;
  max_ref_idx[1] = max1;
  if (intra) n_rec4_intra += np;
  if (unfiltered) has_bypass = true;
  if (pcm) {
    // TDecCu::xReconPCM (TLibDecoder/TDecCu.cpp:845-871): no prediction, no residual; the samples travel in the
    // coefficient stream and the transform kernel stores them
    if (!ctu->pcm[0] || !ctu->pcm[1] || !ctu->pcm[2] || seq.pcm_bd_y <= 0 || seq.pcm_bd_c <= 0 || size > 32) {
      error = "IPCM coding unit without PCM samples, PCM bit depths, or larger than 32x32";
      return;
    }
    has_pcm = true;
    for (int comp = 0; comp < 3; comp++) {
      const int log2n = ilog2(size) - (comp ? 1 : 0), nn = 1 << (2 * log2n);
      const int16_t *src = ctu->pcm[comp] + (comp ? 4 * z : 16 * z);
      const uint32_t off = append_block(src, nn);
      TuItem t;
      std::memset(&t, 0, sizeof(t));
      t.x = (uint16_t)(comp ? x >> 1 : x);
      t.y = (uint16_t)(comp ? y >> 1 : y);
      t.comp = (uint8_t)comp;
      t.flags = TU_PCM;
      t.qp = ctu->qp[z];
      t.coeff_off = off;
      tu[log2n - 2].push_back(t);
    }
    return;
  }
  if (intra) {
    // TDecCu::xReconIntraQT, TLibDecoder/TDecCu.cpp:652-674: all luma blocks of the CU, then all chroma blocks.
    const int parts = ctu->part_size[z] == 3 ? 4 : 1;
    const int init = parts == 4 ? 1 : 0;
    for (int chroma = 0; chroma < 2; chroma++)
      for (int pu = 0; pu < parts; pu++) intra_tus(ctu, ctux, ctuy, z, z + (parts == 4 ? pu * (np / 4) : 0), size >> init, init, chroma);
  } else {
    for (int comp = 0; comp < 3; comp++) inter_tus(ctu, ctux, ctuy, z, comp, z, size, 0);
  }
}

// Appends the non-zero values of a block to the sparse coefficient stream (values saturated to 16 bits, which
// loses nothing because de-quantisation clips its input to 16 bits first: TLibCommon/TComTrQuant.cpp:1181).
// Returns the index of the block's header word.
template <typename T> uint32_t PackedPicture::append_block(const T *src, int nn) {
  // (the vector is kept larger than what has been written, coeff_len, so that appending neither checks capacity nor
  // clears memory per coefficient; finish() cuts it to size)
  const size_t off = coeff_len;
  if (coeff.size() < off + 1 + (size_t)nn) coeff.resize(std::max(coeff.size() * 2, off + 1 + (size_t)nn + 4096));
  uint32_t *dst = coeff.data() + off + 1;
  uint32_t count = 0;
  // most coefficients of a coded block are zero: 16 bytes of zeros at a time are skipped (nn is a multiple of 16)
  constexpr int G = 16 / (int)sizeof(T);
  for (int i = 0; i < nn; i += G) {
    uint64_t w[2];
    std::memcpy(w, src + i, 16);
    if (!(w[0] | w[1])) continue;
    for (int k = i; k < i + G; k++) {
      const int16_t v = sat16(src[k]);
      dst[count] = (uint32_t)k | ((uint32_t)(uint16_t)v << 16);
      count += v != 0;
    }
  }
  coeff[off] = count;
  coeff_len = off + 1 + count;
  coded_coeffs += nn;
  return (uint32_t)off;
}

// Appends one coded transform block: its work item and its coefficients.
void PackedPicture::emit_tu(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int flags,
                            uint32_t *off_out) {
  if (log2n < 2 || log2n > 5) {
    error = "transform block size out of range";
    if (off_out) *off_out = 0;
    return;
  }
  const int n = 1 << log2n, nn = n * n;
  const int32_t *src = ctu->coeff[comp] + (comp ? 4 * z : 16 * z);
  const uint32_t off = append_block(src, nn);
  TuItem t;
  std::memset(&t, 0, sizeof(t));
  t.x = (uint16_t)x;
  t.y = (uint16_t)y;
  t.comp = (uint8_t)comp;
  t.flags = (uint8_t)(flags | (ctu->transform_skip[comp][z] ? TU_TSKIP : 0) |
                      ((ctu->transquant_bypass && ctu->transquant_bypass[z]) ? TU_BYPASS : 0));
  t.qp = ctu->qp[zcu];
  t.coeff_off = off;
  tu[log2n - 2].push_back(t);
  if (off_out) *off_out = off;
}

void PackedPicture::inter_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int comp, int z, int size, int trdepth) {
  if (!((ctu->cbf[comp][z] >> trdepth) & 1)) return;
  if (trdepth < ctu->tr_idx[z]) {
    if (size <= 4) {
      error = "transform block size out of range";
      return;
    }
    const int np = (size >> 2) * (size >> 2) / 4;
    for (int q = 0; q < 4; q++) inter_tus(ctu, ctux, ctuy, zcu, comp, z + q * np, size >> 1, trdepth + 1);
    return;
  }
  int log2n = ilog2(size);
  int x = ctux + 4 * kZ.x[z], y = ctuy + 4 * kZ.y[z];
  if (comp) {
    if (log2n == 2) {
      if (z & 3) return;  // one 4x4 chroma block per 8x8 luma area, carried by its first quadrant
    } else {
      log2n--;
    }
    x >>= 1;
    y >>= 1;
  }
  emit_tu(ctu, comp, z, zcu, x, y, log2n, 0, NULL);
}

void PackedPicture::emit_intra(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int mode) {
  const int n = 1 << log2n;
  const int sh = comp ? 1 : 0;
  const int us = 4 >> sh;           // samples per availability unit in this plane
  const int nu = n / us;            // units per block side
  const int ux0 = (x << sh) >> 2, uy0 = (y << sh) >> 2;
  if (!ctu_has_intra) {  // the level maps cover one CTU: cleared when its first intra block arrives
    std::memset(ctu_lvl, 0, sizeof(ctu_lvl));
    ctu_has_intra = true;
  }
  uint16_t *lvl = ctu_lvl[comp ? 1 : 0];
  const int cur = coding_order(ux0, uy0);
  // A neighbouring unit can be used iff it is inside the picture and precedes this block in coding order
  // (TComPattern::is*Available over TComDataCU::getPU*, TLibCommon/TComPattern.cpp:653-830,
  // TLibCommon/TComDataCU.cpp:1141-1589, for one slice / one tile); with constrained intra prediction it must
  // also be intra itself.
  // Local dependency level: one more than the highest level among the intra blocks OF THE SAME CTU that cover
  // an available neighbouring unit (blocks of other CTUs are finished before this CTU starts: CTU waves).
  const int my_cx = ux0 >> 4, my_cy = uy0 >> 4;
  // A unit of ANOTHER CTU matters only when an intra block covers it: then this CTU has to wait for that CTU's
  // wave (inter samples are final before any intra kernel runs).
  // ... and, when it lies in another CTU, not before the start of this block's slice and in the same tile
  // (TComDataCU::getPULeft and friends check slice and tile only when they leave the CTU)
  const int slice_start = ctu->slice_start ? (int)ctu->slice_start[z] : 0;
  const int my_tile = ctu->tile_id;
  auto usable = [&](int ux, int uy) -> bool {
    const int o = coding_order(ux, uy);
    if (o >= cur) return false;
    if ((o >> 8) == (cur >> 8)) return true;
    return o >= slice_start && ctu_tile[(size_t)(uy >> 4) * seq.ctus_w + (ux >> 4)] == my_tile;
  };
  auto is_intra = [&](int ux, int uy) -> bool { return (unit(ux, uy).flags & P4_INTRA) != 0; };
  IntraNeighbours nb = {0, 0, 0};
  auto level_at = [&](int ux, int uy) -> int {
    if ((ux >> 4) == my_cx && (uy >> 4) == my_cy) return lvl[(uy & 15) * 16 + (ux & 15)];
    if (unit(ux, uy).flags & P4_INTRA) {
      ctu_dep_wave = std::max<int>(ctu_dep_wave, ctu_wave[(size_t)(uy >> 4) * seq.ctus_w + (ux >> 4)]);
      // neighbours of a block lie in the CTU to the left, above-left, above or above-right of its own
      const int dx = (ux >> 4) - my_cx, dy = (uy >> 4) - my_cy;
      ctu_dep_mask |= dy == 0 ? IC_DEP_LEFT : dx < 0 ? IC_DEP_ABOVE_LEFT : dx == 0 ? IC_DEP_ABOVE : IC_DEP_ABOVE_RIGHT;
    }
    return 0;
  };
  int level = 0;
  uint16_t left = 0, above = 0;
  int corner = 0;
  for (int j = 0; j < 2 * nu; j++) {
    const int ux = ux0 - 1, uy = uy0 + j;
    if (ux >= 0 && uy < seq.uh && usable(ux, uy)) {
      left |= (uint16_t)(1u << j);
      if (is_intra(ux, uy)) nb.left |= (uint16_t)(1u << j);
      level = std::max<int>(level, level_at(ux, uy));
    }
  }
  for (int i = 0; i < 2 * nu; i++) {
    const int ux = ux0 + i, uy = uy0 - 1;
    if (uy >= 0 && ux < seq.uw && usable(ux, uy)) {
      above |= (uint16_t)(1u << i);
      if (is_intra(ux, uy)) nb.above |= (uint16_t)(1u << i);
      level = std::max<int>(level, level_at(ux, uy));
    }
  }
  if (ux0 > 0 && uy0 > 0 && usable(ux0 - 1, uy0 - 1)) {
    corner = 1;
    nb.corner = is_intra(ux0 - 1, uy0 - 1) ? 1 : 0;
    level = std::max<int>(level, level_at(ux0 - 1, uy0 - 1));
  }
  level += 1;
  if (comp < 2)  // Cb and Cr share one map; Cr arrives right after Cb with identical geometry
    for (int j = 0; j < nu; j++)
      for (int i = 0; i < nu; i++) lvl[((uy0 + j) & 15) * 16 + ((ux0 + i) & 15)] = (uint16_t)level;
  if (mode < 0 || mode > 34 || log2n < 2 || log2n > 5) {
    error = mode < 0 || mode > 34 ? "intra prediction mode out of range" : "transform block size out of range";
    return;
  }
  IntraItem it;
  std::memset(&it, 0, sizeof(it));
  it.x = (uint16_t)x;
  it.y = (uint16_t)y;
  it.log2n = (uint8_t)log2n;
  it.comp = (uint8_t)comp;
  it.mode = (uint8_t)mode;
  it.flags = (uint8_t)(corner ? IN_CORNER : 0);
  it.avail_left = left;
  it.avail_above = above;
  it.prep = 0;
  if ((ctu->cbf[comp][z] >> ctu->tr_idx[z]) & 1) {
    emit_tu(ctu, comp, z, zcu, x, y, log2n, TU_TO_RESI | ((comp == 0 && log2n == 2) ? TU_DST : 0), NULL);
    it.flags |= IN_HAS_RESI;
  }
  const int parts = log2n == 5 ? HR_INTRA_SPLIT32 : 1;
  for (int part = 0; part < parts; part++) {
    it.prep = (uint32_t)part << IP_PART_SHIFT;  // the rest of the word is filled in when the picture closes
    ctu_items.push_back(it);
    ctu_item_nb.push_back(nb);
    ctu_item_lvl.push_back((uint16_t)level);
  }
}

void PackedPicture::intra_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int z, int size, int trdepth, int chroma) {
  if (trdepth < ctu->tr_idx[z]) {
    if (size <= 4) {
      error = "transform block size out of range";
      return;
    }
    const int np = (size >> 2) * (size >> 2) / 4;
    for (int q = 0; q < 4; q++) intra_tus(ctu, ctux, ctuy, zcu, z + q * np, size >> 1, trdepth + 1, chroma);
    return;
  }
  const int x = ctux + 4 * kZ.x[z], y = ctuy + 4 * kZ.y[z];
  const int log2n = ilog2(size);
  if (!chroma) {
    emit_intra(ctu, 0, z, zcu, x, y, log2n, ctu->luma_dir[z]);
    return;
  }
  int clog2 = log2n - 1;
  if (log2n == 2) {
    if (z & 3) return;
    clog2 = 2;
  }
  int mode = ctu->chroma_dir[zcu];
  if (mode == 36) mode = ctu->luma_dir[zcu];  // derived mode (TLibDecoder/TDecCu.cpp:608-611)
  emit_intra(ctu, 1, z, zcu, x >> 1, y >> 1, clog2, mode);
  emit_intra(ctu, 2, z, zcu, x >> 1, y >> 1, clog2, mode);
}

// What the intra kernel would otherwise derive per block from the item's other fields (IP_*).
uint32_t PackedPicture::intra_prep(const IntraItem &it) {
  const int n = 1 << it.log2n, us = it.comp ? 2 : 4, nu = n / us;
  // availability of the 4*nu + 1 units in sample order: left column bottom-up, corner, row above left to right
  uint64_t m = 0;
  for (int u = 0; u < 2 * nu; u++)
    if ((it.avail_left >> (2 * nu - 1 - u)) & 1) m |= 1ull << u;
  if (it.flags & IN_CORNER) m |= 1ull << (2 * nu);
  for (int u = 0; u < 2 * nu; u++)
    if ((it.avail_above >> u) & 1) m |= 1ull << (2 * nu + 1 + u);
  auto first_of = [&](int u) { return u < 2 * nu ? u * us : u == 2 * nu ? 2 * n : 2 * n + 1 + (u - 2 * nu - 1) * us; };
  auto last_of = [&](int u) { return u < 2 * nu ? u * us + us - 1 : u == 2 * nu ? 2 * n : 2 * n + 1 + (u - 2 * nu - 1) * us + us - 1; };
  uint32_t prep = 0;
  if (m) {
    const int ua = __builtin_ctzll(m), ub = 63 - __builtin_clzll(m);
    const bool run = (m >> ua) == (ub - ua == 63 ? ~0ull : (1ull << (ub - ua + 1)) - 1);
    prep |= (uint32_t)first_of(ua) << IP_FIRST_SHIFT | (uint32_t)last_of(ub) << IP_LAST_SHIFT |
            (uint32_t)(run ? IP_KIND_RUN : IP_KIND_HOLES) << IP_KIND_SHIFT;
  }
  // smoothing of the reference samples (TComPattern::getPredictorPtr, TLibCommon/TComPattern.cpp:627-651): luma only,
  // not for DC, by the distance of the mode from pure horizontal / vertical
  const int dist = std::min(std::abs(it.mode - 10), std::abs(it.mode - 26));
  const int thr = it.log2n == 2 ? 10 : it.log2n == 3 ? 7 : it.log2n == 4 ? 1 : 0;
  if (it.comp == 0 && it.mode != 1 && dist > thr) prep |= IP_FILTER;
  if (it.mode >= 2) {
    // intraPredAngle (TLibCommon/TComPrediction.cpp:211-230): horizontal modes mirror the vertical ones
    static const int kAngle[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
    const bool hor = it.mode < 18;
    const int a = hor ? -(it.mode - 10) : it.mode - 26, aa = std::abs(a);
    prep |= (uint32_t)((a < 0 ? -kAngle[aa] : kAngle[aa]) + 32) << IP_ANGLE_SHIFT | (uint32_t)aa << IP_ABS_SHIFT;
  }
  return prep;
}

bool PackedPicture::finish(const hevcrecon_pic_params *p) {
  if (!error.empty()) return false;
  if (next_ctu != seq.ctus_w * seq.ctus_h) {
    error = "picture closed before all CTUs were added";
    return false;
  }
  pp = *p;
  for (int l = 0; l < 2; l++)
    if (pp.num_ref[l] < 0 || pp.num_ref[l] > HR_MAX_REFS) {
      error = "bad reference list size";
      return false;
    }
  // Constrained intra prediction (TComPattern::is*Available, TLibCommon/TComPattern.cpp:653-830): a neighbour counts
  // only when it is intra itself.  Levels and waves were derived from the larger neighbour sets, which stays a valid
  // (merely longer) order.
  if (pp.constrained_intra_pred)
    for (size_t i = 0; i < intra.size(); i++) {
      intra[i].avail_left &= intra_nb[i].left;
      intra[i].avail_above &= intra_nb[i].above;
      if (!intra_nb[i].corner) intra[i].flags &= (uint8_t)~IN_CORNER;
    }
  for (size_t i = 0; i < intra.size(); i++) intra[i].prep = intra_prep(intra[i]) | (intra[i].prep & (3u << IP_PART_SHIFT));
  // order the CTUs that hold intra blocks by wave
  int nw = 0;
  for (size_t i = 0; i < ictu_wave.size(); i++) nw = std::max<int>(nw, ictu_wave[i]);
  std::vector<int32_t> count(nw + 1, 0);  // waves start at 1
  for (size_t i = 0; i < ictu_wave.size(); i++) count[ictu_wave[i]]++;
  wave_off.assign(nw + 1, 0);  // wave_off[w-1] = first CTU of wave w; wave_off[nw] = total
  int run = 0;
  for (int w = 1; w <= nw; w++) {
    wave_off[w - 1] = run;
    run += count[w];
  }
  wave_off[nw] = run;
  std::vector<IntraCtu> sorted(ictu.size());
  std::vector<int32_t> cursor(wave_off.begin(), wave_off.end());
  for (size_t i = 0; i < ictu.size(); i++) sorted[cursor[ictu_wave[i] - 1]++] = ictu[i];
  ictu.swap(sorted);
  coeff.resize(coeff_len);
  if (!build_ctu_info()) return false;
  build_runs();
  build_mc_lists();
  // The per-unit working records (2 MB at 1080p) are not needed any more: they go back to the pool, so that a host
  // that keeps thousands of packed pictures around does not pay for them.
  g_rec_pool.give(rec4);
  finished = true;
  return true;
}

// Sorts the picture's inter blocks into the motion-compensation lists.  A square region is uniform when every
// 4x4 unit of it inside the picture is inter and resolves to the same prediction: same references, same
// vectors after TComDataCU::clipMv (TLibCommon/TComDataCU.cpp:3207-3219, the range depends on the unit's
// coding-unit origin), same outcome of TComPrediction::xCheckIdenticalMotion
// (TLibCommon/TComPrediction.cpp:396-411; never taken with weighted bi-prediction).  Explicit weights depend on
// the reference indices only, so they are uniform too.  Every CTU is tried whole, then by 32x32 quadrants, then by 16x16
// blocks; what is left goes by 8x8 quadrants to mc_blk8 (one motion, 8-bit pictures: predicted as a block) and mc_blk
// (predicted unit by unit).
namespace {
enum { MC_NONE, MC_UNIFORM, MC_MIXED };
}

// What one motion record predicts with: the lists in use after the identical-motion shortcut, and the vectors as
// they are (the caller clips them, or knows that clipping changes nothing).  False for an inter record without any
// reference.
bool PackedPicture::mc_of_record(const Part4 &p, McTile *t) const {
  bool use[2] = {p.ref_idx[0] >= 0, p.ref_idx[1] >= 0};
  if (use[0] && use[1] && pp.slice_type == 0 && !use_wp && p.ref_idx[0] < HR_MAX_REFS && p.ref_idx[1] < HR_MAX_REFS &&
      pp.ref_poc[0][p.ref_idx[0]] == pp.ref_poc[1][p.ref_idx[1]] && p.mv[0][0] == p.mv[1][0] && p.mv[0][1] == p.mv[1][1])
    use[1] = false;
  if (!use[0] && !use[1]) return false;
  std::memset(t, 0, sizeof(*t));
  for (int l = 0; l < 2; l++) {
    t->ref[l] = use[l] ? p.ref_idx[l] : (int8_t)-1;
    if (use[l]) {
      t->mv[l][0] = p.mv[l][0];
      t->mv[l][1] = p.mv[l][1];
    }
  }
  return true;
}

// Classification of the square region of `size` luma samples at (x0, y0): units z0 .. z0 + n - 1 of CTU `addr`.
// The runs of the CTU stand for its units (a run is a stretch of identical records); when a vector could be changed
// by TComDataCU::clipMv somewhere in the CTU the units are looked at one by one instead, each with the clipping range
// of its own coding unit.
int PackedPicture::classify_mc(int addr, int x0, int y0, int size, McTile *out) const {
  const int n = (size >> 2) * (size >> 2);
  const int z0 = kZ.z[(y0 >> 2) & 15][(x0 >> 2) & 15];
  const int ctux = x0 & ~63, ctuy = y0 & ~63;
  // a vector inside this range is left alone by clipMv for every coding-unit origin of the CTU
  const int hlo = (-64 - 8 - ctux + 1) * 4, hhi = (seq.w + 8 - (ctux + 60) - 1) * 4;
  const int vlo = (-64 - 8 - ctuy + 1) * 4, vhi = (seq.h + 8 - (ctuy + 60) - 1) * 4;
  bool any_inter = false, uniform = true, have = false, by_unit = false;
  McTile first;
  std::memset(&first, 0, sizeof(first));
  for (uint32_t r = ctu_run_off[addr]; r < ctu_run_off[addr + 1] && !by_unit; r++) {
    const Part4 &p = runs[r];
    if (p.pad + p.pad2[0] < z0) continue;
    if (p.pad >= z0 + n) break;
    if (p.flags & P4_INTRA) {
      uniform = false;
      continue;
    }
    any_inter = true;
    if (!uniform) continue;
    McTile t;
    if (!mc_of_record(p, &t)) {
      uniform = false;
      continue;
    }
    for (int l = 0; l < 2; l++)
      if (t.ref[l] >= 0 && (t.mv[l][0] < hlo || t.mv[l][0] > hhi || t.mv[l][1] < vlo || t.mv[l][1] > vhi)) by_unit = true;
    t.x = (uint16_t)x0;
    t.y = (uint16_t)y0;
    if (!have) {
      first = t;
      have = true;
    } else if (std::memcmp(&first, &t, sizeof(t)) != 0) {
      uniform = false;
    }
  }
  if (by_unit) {
    any_inter = false, uniform = true, have = false;
    const Part4 *rec = &rec4[(size_t)addr * 256];
    for (int z = z0; z < z0 + n; z++) {
      const Part4 &p = rec[z];
      if (!(p.tu & P4_VALID)) continue;
      if (p.flags & P4_INTRA) {
        uniform = false;
        continue;
      }
      any_inter = true;
      if (!uniform) continue;
      McTile t;
      if (!mc_of_record(p, &t)) {
        uniform = false;
        continue;
      }
      const int cusize = 64 >> (p.flags & P4_DEPTH_MASK);
      const int cux = (ctux + 4 * kZ.x[z]) & ~(cusize - 1), cuy = (ctuy + 4 * kZ.y[z]) & ~(cusize - 1);
      const int hmax = (seq.w + 8 - cux - 1) * 4, hmin = (-64 - 8 - cux + 1) * 4;
      const int vmax = (seq.h + 8 - cuy - 1) * 4, vmin = (-64 - 8 - cuy + 1) * 4;
      for (int l = 0; l < 2; l++)
        if (t.ref[l] >= 0) {
          t.mv[l][0] = (int16_t)std::min(hmax, std::max(hmin, (int)t.mv[l][0]));
          t.mv[l][1] = (int16_t)std::min(vmax, std::max(vmin, (int)t.mv[l][1]));
        }
      t.x = (uint16_t)x0;
      t.y = (uint16_t)y0;
      if (!have) {
        first = t;
        have = true;
      } else if (std::memcmp(&first, &t, sizeof(t)) != 0) {
        uniform = false;
      }
    }
  }
  if (!any_inter) return MC_NONE;
  if (uniform && have) {
    *out = first;
    return MC_UNIFORM;
  }
  return MC_MIXED;
}

void PackedPicture::split_mc(int addr, int x0, int y0, int level) {
  if (x0 >= seq.w || y0 >= seq.h) return;
  const int size = 64 >> level;
  McTile t;
  const int kind = classify_mc(addr, x0, y0, size, &t);
  if (kind == MC_NONE) return;
  if (kind == MC_UNIFORM) {
    mc_tile[level].push_back(t);
    return;
  }
  if (level == 2) {
    // by 8x8 quadrants: one of a single motion is predicted as a block where the kernels have a path for that (8-bit
    // samples), the others unit by unit
    const bool blocks8 = seq.bd_y <= 8 && seq.bd_c <= 8;
    for (int q = 0; q < 4; q++) {
      McBlk b;
      b.x = (uint16_t)(x0 + (q & 1) * 8);
      b.y = (uint16_t)(y0 + (q >> 1) * 8);
      if (b.x >= seq.w || b.y >= seq.h) continue;
      const int k8 = classify_mc(addr, b.x, b.y, 8, &t);
      if (k8 == MC_UNIFORM && blocks8)
        mc_blk8.push_back(b);
      else if (k8 != MC_NONE)
        mc_blk.push_back(b);
    }
    return;
  }
  const int h = size >> 1;
  for (int q = 0; q < 4; q++) split_mc(addr, x0 + (q & 1) * h, y0 + (q >> 1) * h, level + 1);
}

void PackedPicture::build_mc_lists() {
  for (int i = 0; i < 3; i++) mc_tile[i].clear();
  mc_blk.clear();
  mc_blk8.clear();
  if (pp.slice_type == 2) return;
  for (int cy = 0; cy < seq.ctus_h; cy++)
    for (int cx = 0; cx < seq.ctus_w; cx++) split_mc(cy * seq.ctus_w + cx, cx * 64, cy * 64, 0);
}

// rec4 as it crosses the host link: per CTU (raster order), runs of identical records along its z-order.
void PackedPicture::build_runs() {
  runs.clear();
  ctu_run_off.assign((size_t)seq.ctus_w * seq.ctus_h + 1, 0);
  for (int cy = 0; cy < seq.ctus_h; cy++)
    for (int cx = 0; cx < seq.ctus_w; cx++) {
      const int addr = cy * seq.ctus_w + cx;
      ctu_run_off[addr] = (uint32_t)runs.size();
      const Part4 *rec = &rec4[(size_t)addr * 256];
      int z = 0;
      while (z < 256) {
        if (!(rec[z].tu & P4_VALID)) {  // outside the picture
          z++;
          continue;
        }
        int e = z + 1;
        while (e < 256 && same_record(rec[e], rec[z])) e++;
        Part4 r = rec[z];
        r.pad = (uint8_t)z;
        r.pad2[0] = (uint8_t)(e - z - 1);
        runs.push_back(r);
        z = e;
      }
    }
  ctu_run_off.back() = (uint32_t)runs.size();
}

int64_t PackedPicture::algorithmic_bytes() const {
  const int64_t by = seq.bd_y > 8 ? 2 : 1, bc = seq.bd_c > 8 ? 2 : 1;
  const int64_t sy = (int64_t)seq.w * seq.h, sc = (int64_t)seq.cw * seq.ch * 2;
  const int64_t sample_bytes = sy * by + sc * bc;
  const bool sao = pp.sao_luma || pp.sao_chroma;
  // one store of the reconstruction, read+write by deblocking, read+write by SAO when it is on
  int64_t total = sample_bytes * (1 + 2 + (sao ? 2 : 0));
  total += mc_ref_samples * by;  // luma and chroma have the same sample size in every supported profile
  total += 2 * coded_coeffs;
  total += 16 * (int64_t)seq.uw * seq.uh;
  return total;
}
