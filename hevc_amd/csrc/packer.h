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
  std::vector<Part4> rec4;
  std::vector<TuItem> tu[4];
  std::vector<int16_t> coeff;
  std::vector<IntraItem> intra;     // sorted by level after finish()
  std::vector<int32_t> level_off;   // n_levels + 1
  std::vector<SaoCtu> sao;
  int64_t mc_ref_samples;           // luma+chroma samples predicted, counted once per reference used
  int next_ctu;
  bool finished;
  std::string error;

  // scratch
  std::vector<uint16_t> intra_level;  // parallel to intra before sorting
  std::vector<uint16_t> lvl_y, lvl_c; // per 4x4 unit: dependency level of the intra block covering it (0 = none)

  void begin(const SeqInfo &s);
  bool add_ctu(int addr, const hevcrecon_ctu_syntax *ctu);
  bool finish(const hevcrecon_pic_params *pp);
  int64_t algorithmic_bytes() const;

 private:
  void walk_cu(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int z, int depth);
  void inter_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int comp, int z, int size, int trdepth);
  void intra_tus(const hevcrecon_ctu_syntax *ctu, int ctux, int ctuy, int zcu, int z, int size, int trdepth, int chroma);
  void emit_tu(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int flags, uint32_t *off_out);
  void emit_intra(const hevcrecon_ctu_syntax *ctu, int comp, int z, int zcu, int x, int y, int log2n, int mode);
  int coding_order(int ux, int uy) const;
};

#endif
