// packer.h — host side of the boundary: turns one picture's parsed syntax (the reference's
// per-CTU TComDataCU arrays, handed over as plain pointers) into the flat work lists the
// kernels consume.  Pure C++, no HIP: usable (and tested) on a machine without a GPU.
#ifndef HEVCRECON_PACKER_H
#define HEVCRECON_PACKER_H

#include <vector>
#include <string>
#include "../../include/hevcrecon.h"
#include "common.h"

struct PackedPicture {
  SeqInfo seq;
  hevcrecon_pic_params pp;
  // Working records while the picture is packed: per CTU (raster address), z-order inside, 256 each; taken from a
  // pool by begin() and given back by finish() (the runs carry the records on)
  std::vector<Part4> rec4;
  int64_t n_rec4, n_rec4_intra;     // 4x4 units of the picture (the device's raster array), and how many of them are intra
  std::vector<TuItem> tu[4];
  // Coded coefficients, sparse: per transform block one header word (number of non-zero coefficients) followed by
  // that many words (position inside the block, raster) | (value << 16)
  std::vector<uint32_t> coeff;
  size_t coeff_len;  // words written so far while the picture is packed (the vector may be larger until finish())
  int64_t coded_coeffs;  // coefficients of all coded blocks, zeros included (the figure the algorithmic bytes use)
  template <typename T> uint32_t append_block(const T *src, int nn);
  std::vector<IntraItem> intra;     // grouped by CTU, by local level inside a CTU
  std::vector<IntraCtu> ictu;       // CTUs holding intra blocks, sorted by wave after finish()
  std::vector<uint16_t> lvl_counts; // per CTU, per local level: number of blocks (host only: statistics)
  std::vector<uint16_t> rounds;     // per CTU: the rounds the intra kernel takes its blocks in (IR_*)
  std::vector<int32_t> wave_off;    // n_waves + 1 offsets into ictu
  std::vector<SaoCtu> sao;
  // Slices that differ in their parameters or are not filtered across (set any time before finish(); empty: the picture
  // parameters hold for every slice), the slice every CTU belongs to (its start address), and what finish() derives
  // from both and from the tiles for the kernels (empty when every boundary is crossed with one set of parameters)
  std::vector<hevcrecon_slice_params> slices;
  std::vector<uint32_t> ctu_slice;
  bool ctu_in_two_slices;  // some CTU holds partitions of more than one slice (refused when a slice table arrives)
  std::vector<CtuInfo> ctu_info;
  bool set_slices(int n, const hevcrecon_slice_params *sl);
  std::vector<McTile> mc_tile[3];   // filled by finish(): 64x64 / 32x32 / 16x16 regions predicted as one block
  std::vector<McBlk> mc_blk;        // filled by finish(): 8x8 blocks predicted unit by unit
  std::vector<McBlk> mc_blk8;       // filled by finish(): 8x8 blocks of one motion inside 16x16 regions of mixed motion (8-bit pictures)
  bool has_bypass;                  // some coding unit is exempt from in-loop filtering (lossless, or IPCM with the filters off)
  bool has_pcm;                     // some coding unit is IPCM
  bool use_scaling;                 // the picture de-quantises with `scaling`
  ScalingTables scaling;
  bool use_wp;                      // the picture predicts with the explicit weights `wp`
  WpTables wp;
  int max_ref_idx[2];               // largest reference index used by an inter block, per list (-1: none)
  bool inter_without_ref;           // an inter block with no reference at all (corrupt input)
  int64_t mc_ref_samples;           // luma+chroma samples predicted, counted once per reference used
  int next_ctu;
  std::vector<int32_t> ctu_order;  // raster address -> position in coding order (-1: not added yet)
  std::vector<int32_t> ctu_tile;   // raster address -> tile
  bool finished;
  std::string error;

  // scratch
  // rec4 as it crosses the host link: runs of identical records along the z-order of each CTU (in-picture units
  // only), and the index of every CTU's first run (+ the total)
  std::vector<Part4> runs;
  std::vector<uint32_t> ctu_run_off;
  void build_runs();
  std::vector<IntraItem> ctu_items;   // intra blocks of the CTU being added, in coding order
  std::vector<uint16_t> ctu_item_lvl; // their local levels
  // Per intra block, which of its available neighbour units are themselves intra (same bit layout as
  // IntraItem::avail_left / avail_above, bit 0 of `corner`): constrained intra prediction, known only when the
  // picture parameters arrive, keeps just those.
  struct IntraNeighbours {
    uint16_t left, above;
    uint8_t corner;
  };
  std::vector<IntraNeighbours> ctu_item_nb, intra_nb;  // of ctu_items / of intra (same indices)
  uint16_t ctu_lvl[2][256];           // per 4x4 unit of the CTU being added (16 x 16, luma / chroma): local level of the intra block covering it (0 = none)
  bool ctu_has_intra;                 // ctu_lvl has been cleared for this CTU
  Part4 *cur_rec;                     // records of the CTU being added
  std::vector<uint16_t> ctu_wave;     // per CTU: wave (0 = no intra blocks)
  std::vector<uint16_t> ictu_wave;    // parallel to ictu before sorting
  int ctu_dep_wave;                   // highest wave among the CTUs whose intra blocks the CTU being added reads
  int ctu_dep_mask;                   // which of them (IC_DEP_*)

  void begin(const SeqInfo &s);
  void release_work();  // gives the working records back without finishing (the picture arrives packed from elsewhere)
  bool add_ctu(int addr, const hevcrecon_ctu_syntax *ctu);
  bool finish(const hevcrecon_pic_params *pp);
  int64_t algorithmic_bytes() const;

 private:
  void walk_cu(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int depth);
  void inter_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int comp, int z, int size, int trdepth);
  void intra_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int z, int size, int trdepth, int chroma);
  void emit_tu(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int flags, uint32_t *off_out);
  void emit_intra(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int mode);
  static uint32_t intra_prep(const IntraItem &it);
  int coding_order(int ux, int uy) const;
  const hevcrecon_slice_params *slice_at(uint32_t start) const;
  bool build_ctu_info();
  void build_mc_lists();
  void split_mc(int addr, int x0, int y0, int level);
  int classify_mc(int addr, int x0, int y0, int size, McTile *out) const;
  bool mc_of_record(const Part4 &p, McTile *t) const;
  const Part4 &unit(int ux, int uy) const;
};

#endif
