// hm_bridge.cpp — glue between the reference decoder (HM 9.1) and the reconstruction engine's C ABI.
//
// TEST INFRASTRUCTURE.  This file is compiled together with the reference's own
// sources (from where they lie, never copied) into oracle/_ref/TAppDecoderBridge.
// It takes over reference call sites WITHOUT editing reference source: the build
// renames the reference's definitions with -D macros (see oracle/Makefile) and this
// file supplies the symbols the rest of the decoder links against:
//
//   TDecCu::decompressCU            TLibDecoder/TDecSlice.cpp:367   (per-CTU reconstruction -> the CTU is packed while the picture is parsed)
//   TDecGop::filterPicture          TLibDecoder/TDecTop.cpp:203     (picture finish: hand-over, reconstruction, hash check)
//   TComLoopFilter::loopFilterPic   TLibDecoder/TDecGop.cpp:181     (deblocking)
//   TComSampleAdaptiveOffset::SAOProcess  TLibDecoder/TDecGop.cpp:199 (SAO)
//   TComSampleAdaptiveOffset::PCMLFDisableProcess  TLibDecoder/TDecGop.cpp:200 (restoring lossless units after SAO)
//   TComPic::compressMotion         TLibDecoder/TDecGop.cpp:210     (runs beside the GPU; the picture is fetched after it)
//   calcMD5                         TLibDecoder/TDecGop.cpp:269     (timed; not called at all when the device hashes)
//   TComPicYuv::copyToPic +         TLibDecoder/TDecTop.cpp:244,353 (concealment of a lost reference, TDecTop::xCreateLostPicture: the copy
//   TComSlice::checkThatAllRefPicsAreAvailable   the reference makes is repeated device to device in the engine's picture buffer)
//   TComPicYuv::extendPicBorder     TLibCommon/TComSlice.cpp:374-400 (host border padding: nothing reads it in engine mode)
//   TDecTop::deletePicBuffer        App/TAppDecoder/TAppDecTop.cpp:203 (end of decoding: outstanding hash checks are collected)
//
// Modes (environment variable HEVCRECON_MODE):
//   unset / "ref"  : behave exactly like the reference.
//   "dump"         : run the reference, and write every picture's parsed syntax plus MD5s of
//                    the picture before deblocking / after deblocking / final to
//                    $HEVCRECON_DUMP (golden-vector generator).  HEVCRECON_DUMP_PLANES=1 adds
//                    the three stage pictures themselves.
//   "engine"       : the reference parses only; reconstruction, deblocking and SAO run in the
//                    engine loaded from $HEVCRECON_LIB (libhevcrecon.so).  Every CTU is handed to the
//                    engine's packer as soon as it is parsed (hevcrecon_syntax_add_ctu where decompressCU was),
//                    so packing rides inside the parse and the picture is ready for the GPU when its last CTU is.
//                    HEVCRECON_HASH unset / "host": the final picture is copied back into TComPicYuv, so the
//                    reference's own MD5 check, output writer and DPB logic keep working unchanged.
//                    HEVCRECON_HASH=device (decoding without -o): no picture crosses the host link at all; the
//                    engine hashes the pictures on the GPU beside the reconstruction of later ones
//                    (hevcrecon_picture_md5_submit / _poll) and the status lines, printed in the reference's format
//                    with its (OK) / (***ERROR***) verdicts and exit code, follow a few pictures later.
//   "hub"          : as "engine" with HEVCRECON_HASH=device, but this process never opens the GPU: the pictures are packed
//                    into a host-only syntax object and handed to the hub process named by $HEVCRECON_HUB
//                    (hevcrecon_hub_submit; hevc_amd/hevcrecon_hub), which reconstructs the pictures of all connected
//                    decoders in common batches, hashes them and returns the digests.
//   "pack"         : the reference reconstructs as usual; the CTUs are additionally packed into a host-only
//                    syntax object exactly as in engine mode (no GPU needed): timing and tests of the host side.
//   HEVCRECON_TIMING=1 prints one line of JSON to stderr when decoding ends: where the host thread's time went.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <string>
#include <vector>
#include <dlfcn.h>

// Declare both the renamed reference definition (xxx_ref) and the symbol this file supplies (xxx)
// inside the reference's own class declarations.
#define copyToPic copyToPic_ref(TComPicYuv *pcPicYuvDst); Void copyToPic
#define extendPicBorder extendPicBorder_ref(); Void extendPicBorder
#include "TLibCommon/TComPicYuv.h"
#undef copyToPic
#undef extendPicBorder
void calcMD5_ref(TComPicYuv &pic, UChar digest[3][16]);
#define checkThatAllRefPicsAreAvailable \
  checkThatAllRefPicsAreAvailable_ref(TComList<TComPic *> &rcListPic, TComReferencePictureSet *pReferencePictureSet, Bool printErrors, Int pocRandomAccess); \
  Int checkThatAllRefPicsAreAvailable
#include "TLibCommon/TComSlice.h"
#undef checkThatAllRefPicsAreAvailable
#define decompressCU decompressCU_ref(TComDataCU *pcCU); Void decompressCU
#define loopFilterPic loopFilterPic_ref(TComPic *pcPic); Void loopFilterPic
#define SAOProcess SAOProcess_ref(SAOParam *pcSaoParam); Void SAOProcess
#define PCMLFDisableProcess PCMLFDisableProcess_ref(TComPic *pcPic); Void PCMLFDisableProcess
#define compressMotion compressMotion_ref(); Void compressMotion
#define filterPicture filterPicture_ref(TComPic *&rpcPic); Void filterPicture
#define deletePicBuffer deletePicBuffer_ref(); Void deletePicBuffer
#include "TLibDecoder/TDecCu.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibCommon/TComSampleAdaptiveOffset.h"
#include "TLibCommon/TComPic.h"
#include "TLibDecoder/TDecGop.h"
#include "TLibDecoder/TDecTop.h"
#undef decompressCU
#undef loopFilterPic
#undef SAOProcess
#undef PCMLFDisableProcess
#undef compressMotion
#undef filterPicture
#undef deletePicBuffer
#include "libmd5/MD5.h"

#include "../include/hevcrecon.h"

extern bool g_md5_mismatch;  // App/TAppDecoder/decmain.cpp:46: turns into the decoder's exit code

namespace {

enum Mode { MODE_REF, MODE_DUMP, MODE_ENGINE, MODE_PACK, MODE_HUB };

double now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double wall_clock() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// Where the host thread's time goes, in seconds (HEVCRECON_TIMING=1): everything between the first and the last
// picture that is not listed here is the reference's own work (parsing, its bookkeeping, output).
struct Timing {
  bool on = false;
  long pictures = 0, ctus = 0, repacked = 0, hash_batches = 0;
  double t_process = 0, t_first = 0, t_last = 0, wall_first = 0, wall_last = 0;
  double engine_init = 0;  // loading the library + hevcrecon_create (GPU initialisation)
  double pack = 0;         // fill_ctu + hevcrecon_syntax_add_ctu, inside the parse
  double finish = 0;       // picture parameters + hevcrecon_syntax_finish
  double submit = 0;       // hevcrecon_reconstruct (queues copies and kernels)
  double motion = 0;       // TComPic::compressMotion of the reference (runs beside the GPU)
  double wait = 0;         // waiting for the GPU before the picture is fetched
  double download = 0;     // device-to-host copy + conversion into TComPicYuv
  double md5_host = 0;     // the reference's calcMD5 on the host thread
  double hash_device = 0;  // submitting and polling the device hash
};

struct EngineApi {
  void *lib;
  int (*create)(hevcrecon_ctx **, int);
  const char *(*last_error)(const hevcrecon_ctx *);
  int (*sequence_init)(hevcrecon_ctx *, const hevcrecon_seq_params *);
  int (*syntax_create)(hevcrecon_ctx *, hevcrecon_syntax **);
  int (*syntax_begin)(hevcrecon_syntax *);
  int (*syntax_set_scaling_lists)(hevcrecon_syntax *, const hevcrecon_scaling_lists *);
  int (*syntax_set_pred_weights)(hevcrecon_syntax *, const hevcrecon_pred_weights *);
  int (*syntax_set_slices)(hevcrecon_syntax *, int, const hevcrecon_slice_params *);
  int (*syntax_add_ctu)(hevcrecon_syntax *, int, const hevcrecon_ctu_syntax *);
  int (*syntax_finish)(hevcrecon_syntax *, const hevcrecon_pic_params *);
  int (*reconstruct)(hevcrecon_ctx *, hevcrecon_syntax *, int);
  int (*picture_download)(hevcrecon_ctx *, int, int16_t *, int16_t *, int16_t *, int, int);
  int (*picture_copy)(hevcrecon_ctx *, int, int, int);
  int (*synchronize)(hevcrecon_ctx *);
  int (*md5_submit)(hevcrecon_ctx *, int, const int *, int64_t *);
  int (*md5_poll)(hevcrecon_ctx *, int64_t, int *, uint8_t *);
  int (*md5_wait)(hevcrecon_ctx *, int64_t, uint8_t *);
  int (*syntax_create_detached)(const hevcrecon_seq_params *, hevcrecon_syntax **);
  int (*hub_connect)(const char *, const hevcrecon_seq_params *, hevcrecon_hub **);
  int (*hub_pic_id_base)(const hevcrecon_hub *);
  int (*hub_submit)(hevcrecon_hub *, const hevcrecon_syntax *, int);
  int (*hub_copy)(hevcrecon_hub *, int, int, int);
  int (*hub_poll)(hevcrecon_hub *, int *, uint8_t *);
  int (*hub_wait)(hevcrecon_hub *, uint8_t *);
  const char *(*hub_error)(const hevcrecon_hub *);
  void (*hub_disconnect)(hevcrecon_hub *);
  const char *(*syntax_last_error)(const hevcrecon_syntax *);
  void (*syntax_destroy)(hevcrecon_syntax *);
};

// A status line whose hash verdict is still to come (HEVCRECON_HASH=device)
struct PendingLine {
  std::string text;          // "POC ... [L1 ...] " as the reference prints it
  bool hash_field;           // the decoder prints a hash field at all (--SEIpictureDigest)
  bool check;                // an MD5 SEI message came with the picture
  unsigned char want[3][16]; // its digests
  int pic_id;
  int64_t ticket;            // -1: not submitted yet
  int index;                 // position inside its ticket
};

struct Bridge {
  Mode mode;
  FILE *dump;
  bool dump_planes;
  bool header_written;
  std::map<TComPic *, int> pic_ids;
  EngineApi api;
  hevcrecon_ctx *ctx;
  hevcrecon_hub *hub = NULL;  // MODE_HUB: the connection to the process that owns the GPU
  // Packed pictures in flight: a syntax object is rewritten only when the GPU is done with the picture it held, so
  // the decoder rotates over a few of them and never waits for that.
  enum { kMaxSyntaxObjects = 32 };
  int kSyntaxObjects;  // HEVCRECON_DEPTH (default 4): pictures the host may run ahead of the GPU
  hevcrecon_syntax *syns[kMaxSyntaxObjects];
  int syn_pos;
  hevcrecon_syntax *syn;  // the one the picture being parsed is packed into
  bool device_hash;
  Timing tm;
  unsigned char stage_md5[3][3][16];
  std::vector<uint16_t> stage_planes[3];
  Bridge() : mode(MODE_REF), dump(NULL), dump_planes(false), header_written(false), ctx(NULL), syn_pos(0), syn(NULL), device_hash(false) {
    memset(syns, 0, sizeof(syns));
    memset(&api, 0, sizeof(api));
    kSyntaxObjects = getenv("HEVCRECON_DEPTH") ? atoi(getenv("HEVCRECON_DEPTH")) : 4;
    if (kSyntaxObjects < 1 || kSyntaxObjects > kMaxSyntaxObjects) kSyntaxObjects = 4;
    tm.on = getenv("HEVCRECON_TIMING") && atoi(getenv("HEVCRECON_TIMING"));
    tm.t_process = now();
    const char *m = getenv("HEVCRECON_MODE");
    if (m && !strcmp(m, "dump")) {
      mode = MODE_DUMP;
      const char *path = getenv("HEVCRECON_DUMP");
      dump = fopen(path ? path : "hevcrecon.syn", "wb");
      if (!dump) { perror("HEVCRECON_DUMP"); exit(EXIT_FAILURE); }
      dump_planes = getenv("HEVCRECON_DUMP_PLANES") && atoi(getenv("HEVCRECON_DUMP_PLANES"));
    } else if (m && (!strcmp(m, "engine") || !strcmp(m, "pack") || !strcmp(m, "hub"))) {
      mode = !strcmp(m, "engine") ? MODE_ENGINE : !strcmp(m, "hub") ? MODE_HUB : MODE_PACK;
      const double t0 = now();
      load_engine();
      tm.engine_init = now() - t0;
      const char *h = getenv("HEVCRECON_HASH");
      device_hash = (mode == MODE_ENGINE && h && !strcmp(h, "device")) || mode == MODE_HUB;
      if (mode == MODE_HUB && writes_output_file()) {
        fprintf(stderr, "hm_bridge: with a hub the pictures stay on the GPU: no output file can be written (use HEVCRECON_MODE=engine)\n");
        exit(EXIT_FAILURE);
      }
      if (device_hash && writes_output_file()) {
        fprintf(stderr, "hm_bridge: HEVCRECON_HASH=device keeps the pictures on the GPU; an output file was asked for, so they are fetched and hashed on the host\n");
        device_hash = false;
      }
    }
  }
  ~Bridge() { if (dump) fclose(dump); }

  // Was the decoder started with -o / --ReconFile?  (App/TAppDecoder/TAppDecCfg.cpp:68-82; the option parser is the
  // application's and out of this file's reach, so the command line is read where the kernel keeps it.)
  static bool writes_output_file() {
    FILE *f = fopen("/proc/self/cmdline", "rb");
    if (!f) return true;
    std::string all;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) all.append(buf, n);
    fclose(f);
    bool first = true, found = false;
    for (size_t i = 0; i < all.size(); i += strlen(all.c_str() + i) + 1) {
      const char *a = all.c_str() + i;
      if (!first && (!strcmp(a, "-o") || !strncmp(a, "--ReconFile", 11))) found = true;
      first = false;
    }
    return found;
  }

  void fail(const char *what) {
    const char *why = (ctx && api.last_error) ? api.last_error(ctx) : (syn && api.syntax_last_error && *api.syntax_last_error(syn)) ? api.syntax_last_error(syn)
                      : (hub && api.hub_error) ? api.hub_error(hub) : "";
    fprintf(stderr, "hm_bridge: %s: %s\n", what, why);
    exit(EXIT_FAILURE);
  }
  template <class F> void sym(F &f, const char *name) {
    f = (F)dlsym(api.lib, name);
    if (!f) { fprintf(stderr, "hm_bridge: missing symbol %s\n", name); exit(EXIT_FAILURE); }
  }
  void load_engine() {
    const char *path = getenv("HEVCRECON_LIB");
    api.lib = dlopen(path ? path : "libhevcrecon.so", RTLD_NOW);
    if (!api.lib) { fprintf(stderr, "hm_bridge: %s\n", dlerror()); exit(EXIT_FAILURE); }
    sym(api.create, "hevcrecon_create");
    sym(api.last_error, "hevcrecon_last_error");
    sym(api.sequence_init, "hevcrecon_sequence_init");
    sym(api.syntax_create, "hevcrecon_syntax_create");
    sym(api.syntax_begin, "hevcrecon_syntax_begin");
    sym(api.syntax_set_scaling_lists, "hevcrecon_syntax_set_scaling_lists");
    sym(api.syntax_set_pred_weights, "hevcrecon_syntax_set_pred_weights");
    sym(api.syntax_set_slices, "hevcrecon_syntax_set_slices");
    sym(api.syntax_add_ctu, "hevcrecon_syntax_add_ctu");
    sym(api.syntax_finish, "hevcrecon_syntax_finish");
    sym(api.reconstruct, "hevcrecon_reconstruct");
    sym(api.picture_download, "hevcrecon_picture_download");
    sym(api.picture_copy, "hevcrecon_picture_copy");
    sym(api.synchronize, "hevcrecon_synchronize");
    sym(api.md5_submit, "hevcrecon_picture_md5_submit");
    sym(api.md5_poll, "hevcrecon_picture_md5_poll");
    sym(api.md5_wait, "hevcrecon_picture_md5_wait");
    sym(api.syntax_create_detached, "hevcrecon_syntax_create_detached");
    sym(api.hub_connect, "hevcrecon_hub_connect");
    sym(api.hub_pic_id_base, "hevcrecon_hub_pic_id_base");
    sym(api.hub_submit, "hevcrecon_hub_submit");
    sym(api.hub_copy, "hevcrecon_hub_copy");
    sym(api.hub_poll, "hevcrecon_hub_poll");
    sym(api.hub_wait, "hevcrecon_hub_wait");
    sym(api.hub_error, "hevcrecon_hub_error");
    sym(api.hub_disconnect, "hevcrecon_hub_disconnect");
    sym(api.syntax_last_error, "hevcrecon_syntax_last_error");
    sym(api.syntax_destroy, "hevcrecon_syntax_destroy");
    if (mode == MODE_ENGINE) {
      const char *dev = getenv("HEVCRECON_DEVICE");
      if (api.create(&ctx, dev ? atoi(dev) : 0)) fail("hevcrecon_create");
    }
  }

  bool on_gpu() const { return mode == MODE_ENGINE || mode == MODE_HUB; }  // the reference's own reconstruction does not run
  int id_base = 0;  // MODE_HUB: this connection's range of picture ids
  int pic_id(TComPic *p) {
    std::map<TComPic *, int>::iterator it = pic_ids.find(p);
    if (it != pic_ids.end()) return it->second;
    int id = (int)pic_ids.size() + id_base;
    pic_ids[p] = id;
    return id;
  }

  void fill_seq(TComPic *pic, hevcrecon_seq_params &sp) {
    TComSlice *sl = pic->getSlice(pic->getCurrSliceIdx());
    memset(&sp, 0, sizeof(sp));
    sp.width = sl->getSPS()->getPicWidthInLumaSamples();
    sp.height = sl->getSPS()->getPicHeightInLumaSamples();
    sp.bit_depth_luma = g_bitDepthY;
    sp.bit_depth_chroma = g_bitDepthC;
    sp.ctu_size = g_uiMaxCUWidth;
    sp.max_cu_depth = g_uiMaxCUDepth;
    sp.strong_intra_smoothing = sl->getSPS()->getUseStrongIntraSmoothing();
    if (sl->getSPS()->getUsePCM()) {
      sp.pcm_bit_depth_luma = sl->getSPS()->getPCMBitDepthLuma();
      sp.pcm_bit_depth_chroma = sl->getSPS()->getPCMBitDepthChroma();
      sp.pcm_loop_filter_disable = sl->getSPS()->getPCMFilterDisableFlag();
    }
  }

  void fill_pic(TComPic *pic, hevcrecon_pic_params &pp) {
    TComSlice *sl = pic->getSlice(pic->getCurrSliceIdx());
    memset(&pp, 0, sizeof(pp));
    pp.poc = sl->getPOC();
    pp.slice_type = (int)sl->getSliceType();
    pp.deblocking_disable = sl->getDeblockingFilterDisable();
    pp.beta_offset_div2 = sl->getDeblockingFilterBetaOffsetDiv2();
    pp.tc_offset_div2 = sl->getDeblockingFilterTcOffsetDiv2();
    pp.cb_qp_offset_pps = sl->getPPS()->getChromaCbQpOffset();
    pp.cr_qp_offset_pps = sl->getPPS()->getChromaCrQpOffset();
    pp.cb_qp_delta_slice = sl->getSliceQpDeltaCb();
    pp.cr_qp_delta_slice = sl->getSliceQpDeltaCr();
    // the reference applies the SAO enables of the picture's first slice to the whole picture (TLibDecoder/TDecGop.cpp:183,194-196)
    bool sao = sl->getSPS()->getUseSAO();
    pp.sao_luma = sao && pic->getSlice(0)->getSaoEnabledFlag();
    pp.sao_chroma = sao && pic->getSlice(0)->getSaoEnabledFlagChroma();
    pp.constrained_intra_pred = sl->getPPS()->getConstrainedIntraPred();
    pp.no_loop_filter_across_tiles = !sl->getPPS()->getLoopFilterAcrossTilesEnabledFlag();
    for (int l = 0; l < 2; l++) {
      pp.num_ref[l] = sl->isIntra() ? 0 : sl->getNumRefIdx(RefPicList(l));
      if (l == 1 && !sl->isInterB()) pp.num_ref[l] = 0;
      for (int i = 0; i < pp.num_ref[l] && i < HEVCRECON_MAX_REFS; i++) {
        pp.ref_pic_id[l][i] = pic_id(sl->getRefPic(RefPicList(l), i));
        pp.ref_poc[l][i] = sl->getRefPic(RefPicList(l), i)->getPOC();
      }
    }
  }

  // The scaling lists the decoder activated for this picture's slice (TLibDecoder/TDecTop.cpp:548-561).
  bool fill_scaling(TComPic *pic, hevcrecon_scaling_lists &sl) {
    TComSlice *s = pic->getSlice(pic->getCurrSliceIdx());
    if (!s->getSPS()->getScalingListFlag()) return false;
    TComScalingList *l = s->getScalingList();
    memset(&sl, 0, sizeof(sl));
    for (int id = 0; id < 6; id++) {
      for (int i = 0; i < 16; i++) sl.m4[id][i] = (uint8_t)l->getScalingListAddress(SCALING_LIST_4x4, id)[i];
      for (int i = 0; i < 64; i++) sl.m8[id][i] = (uint8_t)l->getScalingListAddress(SCALING_LIST_8x8, id)[i];
      for (int i = 0; i < 64; i++) sl.m16[id][i] = (uint8_t)l->getScalingListAddress(SCALING_LIST_16x16, id)[i];
      sl.dc16[id] = (uint8_t)l->getScalingListDC(SCALING_LIST_16x16, id);
    }
    for (int id = 0; id < 2; id++) {
      for (int i = 0; i < 64; i++) sl.m32[id][i] = (uint8_t)l->getScalingListAddress(SCALING_LIST_32x32, id)[i];
      sl.dc32[id] = (uint8_t)l->getScalingListDC(SCALING_LIST_32x32, id);
    }
    return true;
  }

  // The slice's explicit prediction weights, when the reference would use them (TLibCommon/TComPrediction.cpp:519-538).
  bool fill_weights(TComPic *pic, hevcrecon_pred_weights &pw) {
    TComSlice *s = pic->getSlice(pic->getCurrSliceIdx());
    const bool weighted = (s->getSliceType() == P_SLICE && s->getPPS()->getUseWP()) ||
                          (s->getSliceType() == B_SLICE && s->getPPS()->getWPBiPred());
    if (!weighted) return false;
    memset(&pw, 0, sizeof(pw));
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < s->getNumRefIdx(RefPicList(l)) && r < HEVCRECON_MAX_REFS; r++) {
        wpScalingParam *wp;
        s->getWpScaling(RefPicList(l), r, wp);
        for (int c = 0; c < 3; c++) {
          pw.log2_denom[l][r][c] = (int16_t)wp[c].uiLog2WeightDenom;
          pw.weight[l][r][c] = (int16_t)wp[c].iWeight;
          pw.offset[l][r][c] = (int16_t)wp[c].iOffset;
        }
      }
    return true;
  }

  // What the slices of a picture must share, because it is handed over once per picture: slice type and reference lists.
  void check_slices(TComPic *pic) {
    TComSlice *a = pic->getSlice(pic->getCurrSliceIdx());
    for (UInt i = 0; i < pic->getNumAllocatedSlice() && i <= pic->getCurrSliceIdx(); i++) {
      TComSlice *b = pic->getSlice(i);
      bool same = a->getSliceType() == b->getSliceType();
      for (int l = 0; same && l < 2; l++) {
        same = a->getNumRefIdx(RefPicList(l)) == b->getNumRefIdx(RefPicList(l));
        for (int r = 0; same && !a->isIntra() && r < a->getNumRefIdx(RefPicList(l)); r++)
          same = a->getRefPic(RefPicList(l), r) == b->getRefPic(RefPicList(l), r);
      }
      if (!same) {
        fprintf(stderr, "hm_bridge: the slices of POC %d differ in their type or reference lists\n", a->getPOC());
        exit(EXIT_FAILURE);
      }
    }
  }

  // The per-slice parameters of the in-loop filters and of the chroma de-quantisation; returns false when the picture
  // parameters already say everything (one slice, or all slices agree with them and filter across their boundaries).
  bool fill_slices(TComPic *pic, const hevcrecon_pic_params &pp, std::vector<hevcrecon_slice_params> &out) {
    out.clear();
    bool needed = false;
    for (UInt i = 0; i < pic->getNumAllocatedSlice() && i <= pic->getCurrSliceIdx(); i++) {
      TComSlice *b = pic->getSlice(i);
      hevcrecon_slice_params sp;
      memset(&sp, 0, sizeof(sp));
      sp.start_addr = b->getSliceCurStartCUAddr();
      bool seen = false;  // dependent slices repeat their slice's address
      for (size_t k = 0; k < out.size(); k++) seen = seen || out[k].start_addr == sp.start_addr;
      if (seen) continue;
      sp.deblocking_disable = b->getDeblockingFilterDisable();
      sp.beta_offset_div2 = b->getDeblockingFilterBetaOffsetDiv2();
      sp.tc_offset_div2 = b->getDeblockingFilterTcOffsetDiv2();
      sp.cb_qp_delta_slice = b->getSliceQpDeltaCb();
      sp.cr_qp_delta_slice = b->getSliceQpDeltaCr();
      sp.loop_filter_across_slices = b->getLFCrossSliceBoundaryFlag();
      needed = needed || sp.deblocking_disable != pp.deblocking_disable || sp.beta_offset_div2 != pp.beta_offset_div2 ||
               sp.tc_offset_div2 != pp.tc_offset_div2 || sp.cb_qp_delta_slice != pp.cb_qp_delta_slice ||
               sp.cr_qp_delta_slice != pp.cr_qp_delta_slice || !sp.loop_filter_across_slices;
      out.push_back(sp);
    }
    return needed && out.size() > 1;
  }

  void fill_ctu(TComPic *pic, int addr, const hevcrecon_pic_params &pp, hevcrecon_ctu_syntax &cs, int8_t *refbuf, uint32_t *slicebuf) {
    TComDataCU *cu = pic->getCU(addr);
    memset(&cs, 0, sizeof(cs));
    for (int i = 0; i < HEVCRECON_PARTS_PER_CTU; i++) slicebuf[i] = cu->getSliceStartCU(i);
    cs.slice_start = slicebuf;
    cs.coding_addr_plus1 = (int32_t)pic->getPicSym()->getInverseCUOrderMap(addr) + 1;
    cs.tile_id = (int32_t)pic->getPicSym()->getTileIdxMap(addr);
    cs.depth = cu->getDepth();
    cs.pred_mode = (const int8_t *)cu->getPredictionMode();
    cs.part_size = (const int8_t *)cu->getPartitionSize();
    cs.tr_idx = cu->getTransformIdx();
    cs.qp = (const int8_t *)cu->getQP();
    cs.luma_dir = cu->getLumaIntraDir();
    cs.chroma_dir = cu->getChromaIntraDir();
    cs.transquant_bypass = (const uint8_t *)cu->getCUTransquantBypass();
    cs.ipcm = (const uint8_t *)cu->getIPCMFlag();
    const TextType tt[3] = {TEXT_LUMA, TEXT_CHROMA_U, TEXT_CHROMA_V};
    for (int c = 0; c < 3; c++) {
      cs.cbf[c] = cu->getCbf(tt[c]);
      cs.transform_skip[c] = cu->getTransformSkip(tt[c]);
    }
    cs.coeff[0] = cu->getCoeffY();
    cs.coeff[1] = cu->getCoeffCb();
    cs.coeff[2] = cu->getCoeffCr();
    cs.pcm[0] = cu->getPCMSampleY();
    cs.pcm[1] = cu->getPCMSampleCb();
    cs.pcm[2] = cu->getPCMSampleCr();
    for (int l = 0; l < 2; l++) {
      TComCUMvField *f = cu->getCUMvField(RefPicList(l));
      cs.mv[l] = (const int16_t *)&f->getMv(0);
      for (int i = 0; i < HEVCRECON_PARTS_PER_CTU; i++) refbuf[l * HEVCRECON_PARTS_PER_CTU + i] = (int8_t)f->getRefIdx(i);
      cs.ref_idx[l] = refbuf + l * HEVCRECON_PARTS_PER_CTU;
    }
    SAOParam *sao = pic->getPicSym()->getSaoParam();
    for (int c = 0; c < 3; c++) {
      bool on = c == 0 ? pp.sao_luma : pp.sao_chroma;
      cs.sao_type[c] = -1;
      if (on && sao && sao->saoLcuParam[c]) {
        const SaoLcuParam &p = sao->saoLcuParam[c][addr];
        cs.sao_type[c] = p.typeIdx;
        cs.sao_subtype[c] = p.subTypeIdx;
        for (int i = 0; i < 4; i++) cs.sao_offset[c][i] = p.typeIdx >= 0 ? p.offset[i] : 0;
      }
    }
  }

  // MD5 with the byte convention of calcMD5 (TLibCommon/TComPicYuvMD5.cpp:188-213).
  void capture_stage(TComPic *pic, int stage) {
    TComPicYuv *yuv = pic->getPicYuvRec();
    int w = yuv->getWidth(), h = yuv->getHeight();
    if (dump_planes) stage_planes[stage].clear();
    for (int c = 0; c < 3; c++) {
      Pel *p = c == 0 ? yuv->getLumaAddr() : c == 1 ? yuv->getCbAddr() : yuv->getCrAddr();
      int stride = c == 0 ? yuv->getStride() : yuv->getCStride();
      int pw = c == 0 ? w : w / 2, ph = c == 0 ? h : h / 2;
      int bd = c == 0 ? g_bitDepthY : g_bitDepthC;
      MD5 md5;
      std::vector<unsigned char> row(pw * 2);
      for (int y = 0; y < ph; y++) {
        int n = 0;
        for (int x = 0; x < pw; x++) {
          Pel v = p[y * stride + x];
          row[n++] = (unsigned char)(v & 0xff);
          if (bd > 8) row[n++] = (unsigned char)(v >> 8);
          if (dump_planes) stage_planes[stage].push_back((uint16_t)v);
        }
        md5.update(&row[0], n);
      }
      md5.finalize(stage_md5[stage][c]);
    }
  }

  template <class T> void put(const T *p, size_t n) { fwrite(p, sizeof(T), n, dump); }

  void write_picture(TComPic *pic) {
    hevcrecon_seq_params sp;
    fill_seq(pic, sp);
    if (!header_written) {
      fwrite("HVSYN2\0\0", 1, 8, dump);
      put(&sp, 1);
      header_written = true;
    }
    hevcrecon_pic_params pp;
    fill_pic(pic, pp);
    check_slices(pic);
    int nctu = pic->getNumCUsInFrame();
    hevcrecon_scaling_lists sl;
    if (fill_scaling(pic, sl)) {  // optional record in front of the picture
      int32_t shdr[4] = {0x4C414353 /* "SCAL" */, 0, 0, 0};
      put(shdr, 4);
      put(&sl, 1);
    }
    std::vector<hevcrecon_ctu_syntax> cs(nctu);
    std::vector<int8_t> refbuf(nctu * 2 * HEVCRECON_PARTS_PER_CTU);
    std::vector<uint32_t> slicebuf(nctu * HEVCRECON_PARTS_PER_CTU);
    for (int a = 0; a < nctu; a++)
      fill_ctu(pic, a, pp, cs[a], &refbuf[a * 2 * HEVCRECON_PARTS_PER_CTU], &slicebuf[a * HEVCRECON_PARTS_PER_CTU]);
    const int P = HEVCRECON_PARTS_PER_CTU;
    {  // optional record in front of the picture: coding order and tile of every CTU, when there are tiles
      bool any = false;
      std::vector<int32_t> tl(2 * nctu);
      for (int a = 0; a < nctu; a++) {
        tl[2 * a] = cs[a].coding_addr_plus1;
        tl[2 * a + 1] = cs[a].tile_id;
        any = any || cs[a].tile_id != 0;
      }
      if (any) {
        int32_t thdr[4] = {0x454C4954 /* "TILE" */, nctu, 0, 0};
        put(thdr, 4);
        put(&tl[0], tl.size());
      }
    }
    {  // optional record in front of the picture: slice start addresses, when there is more than one slice
      bool any = false;
      for (size_t i = 0; i < slicebuf.size() && !any; i++) any = slicebuf[i] != 0;
      if (any) {
        int32_t shdr[4] = {0x54534C53 /* "SLST" */, nctu, 0, 0};
        put(shdr, 4);
        put(&slicebuf[0], slicebuf.size());
      }
    }
    {  // optional record in front of the picture: the PCM sample arrays of the CTUs that hold IPCM units
      std::vector<int32_t> with_pcm;
      for (int a = 0; a < nctu; a++) {
        bool any = false;
        for (int z = 0; z < P && !any; z++) any = cs[a].ipcm[z] != 0;
        if (any) with_pcm.push_back(a);
      }
      if (!with_pcm.empty()) {
        int32_t phdr[4] = {0x534D4350 /* "PCMS" */, (int32_t)with_pcm.size(), 0, 0};
        put(phdr, 4);
        for (size_t i = 0; i < with_pcm.size(); i++) {
          const int a = with_pcm[i];
          put(&with_pcm[i], 1);
          put(cs[a].pcm[0], 16 * P);
          put(cs[a].pcm[1], 4 * P);
          put(cs[a].pcm[2], 4 * P);
        }
      }
    }
    std::vector<hevcrecon_slice_params> slices;
    if (fill_slices(pic, pp, slices)) {  // optional record in front of the picture
      int32_t lhdr[4] = {0x52504C53 /* "SLPR" */, (int32_t)slices.size(), 0, 0};
      put(lhdr, 4);
      put(&slices[0], slices.size());
    }
    hevcrecon_pred_weights pw;
    if (fill_weights(pic, pw)) {  // optional record in front of the picture
      int32_t whdr[4] = {0x44525057 /* "WPRD" */, 0, 0, 0};
      put(whdr, 4);
      put(&pw, 1);
    }
    int32_t hdr[4] = {0x54434950 /* "PICT" */, nctu, pic_id(pic), dump_planes ? 1 : 0};
    put(hdr, 4);
    put(&pp, 1);
    for (int a = 0; a < nctu; a++) put(cs[a].depth, P);
    for (int a = 0; a < nctu; a++) put(cs[a].pred_mode, P);
    for (int a = 0; a < nctu; a++) put(cs[a].part_size, P);
    for (int a = 0; a < nctu; a++) put(cs[a].tr_idx, P);
    for (int c = 0; c < 3; c++) for (int a = 0; a < nctu; a++) put(cs[a].cbf[c], P);
    for (int a = 0; a < nctu; a++) put(cs[a].qp, P);
    for (int a = 0; a < nctu; a++) put(cs[a].luma_dir, P);
    for (int a = 0; a < nctu; a++) put(cs[a].chroma_dir, P);
    for (int c = 0; c < 3; c++) for (int a = 0; a < nctu; a++) put(cs[a].transform_skip[c], P);
    for (int l = 0; l < 2; l++) for (int a = 0; a < nctu; a++) put(cs[a].mv[l], 2 * P);
    for (int l = 0; l < 2; l++) for (int a = 0; a < nctu; a++) put(cs[a].ref_idx[l], P);
    for (int a = 0; a < nctu; a++) put(cs[a].transquant_bypass, P);
    for (int a = 0; a < nctu; a++) put(cs[a].ipcm, P);
    std::vector<int16_t> c16;
    for (int c = 0; c < 3; c++) {
      int n = c == 0 ? 16 * P : 4 * P;
      c16.resize(n);
      for (int a = 0; a < nctu; a++) {
        for (int i = 0; i < n; i++) {
          int32_t v = cs[a].coeff[c][i];
          c16[i] = (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
        }
        put(&c16[0], n);
      }
    }
    for (int a = 0; a < nctu; a++) {
      int32_t s[18];
      for (int c = 0; c < 3; c++) {
        s[c * 6 + 0] = cs[a].sao_type[c];
        s[c * 6 + 1] = cs[a].sao_subtype[c];
        for (int i = 0; i < 4; i++) s[c * 6 + 2 + i] = cs[a].sao_offset[c][i];
      }
      put(s, 18);
    }
    fwrite(stage_md5, 1, sizeof(stage_md5), dump);
    if (dump_planes)
      for (int s = 0; s < 3; s++) put(&stage_planes[s][0], stage_planes[s].size());
    fflush(dump);
  }

  // ---- engine / pack mode: the picture is packed while it is parsed ---------------------------------------
  TComPic *cur = NULL;   // picture whose CTUs are being added
  int cur_added = 0;     // CTUs added so far (they arrive in coding order)
  bool cur_ok = false;   // false: something arrived out of order, the picture is packed again from its arrays at the end
  hevcrecon_pic_params cur_pp;

  void begin_picture(TComPic *pic) {
    hevcrecon_seq_params sp;
    fill_seq(pic, sp);
    if (mode == MODE_ENGINE) {
      if (api.sequence_init(ctx, &sp)) fail("hevcrecon_sequence_init");
      syn_pos = (syn_pos + 1) % kSyntaxObjects;
      if (!syns[syn_pos] && api.syntax_create(ctx, &syns[syn_pos])) fail("hevcrecon_syntax_create");
    } else {
      if (syns[0] && memcmp(&sp, &pack_sp, sizeof(sp))) {
        api.syntax_destroy(syns[0]);
        syns[0] = NULL;
        if (hub) {  // new sequence parameters: a connection of its own (everything of the old one is collected first)
          pump_hashes(true);
          api.hub_disconnect(hub);
          hub = NULL;
          pic_ids.clear();
        }
      }
      if (mode == MODE_HUB && !hub) {
        const double t0 = now();
        const int rc = api.hub_connect(getenv("HEVCRECON_HUB"), &sp, &hub);
        if (rc) {
          fprintf(stderr, "hm_bridge: no hub to connect to (hevcrecon_hub_connect: %d)\n", rc);
          exit(EXIT_FAILURE);
        }
        id_base = api.hub_pic_id_base(hub);
        tm.engine_init += now() - t0;
      }
      pack_sp = sp;
      if (!syns[0] && api.syntax_create_detached(&sp, &syns[0])) fail("hevcrecon_syntax_create_detached");
      syn_pos = 0;
    }
    syn = syns[syn_pos];
    if (api.syntax_begin(syn)) fail("hevcrecon_syntax_begin");
    // what fill_ctu needs of the picture parameters while the picture is still being parsed: the SAO enables (those of
    // the first slice, which has arrived by now); everything else is filled in when the picture is complete
    memset(&cur_pp, 0, sizeof(cur_pp));
    const bool sao = pic->getSlice(0)->getSPS()->getUseSAO();
    cur_pp.sao_luma = sao && pic->getSlice(0)->getSaoEnabledFlag();
    cur_pp.sao_chroma = sao && pic->getSlice(0)->getSaoEnabledFlagChroma();
    cur = pic;
    cur_added = 0;
    cur_ok = true;
    if (tm.pictures == 0) {
      tm.t_first = now();
      tm.wall_first = wall_clock();
    }
  }
  hevcrecon_seq_params pack_sp;

  void add_one_ctu(TComPic *pic, int addr) {
    hevcrecon_ctu_syntax cs;
    int8_t refbuf[2 * HEVCRECON_PARTS_PER_CTU];
    uint32_t slicebuf[HEVCRECON_PARTS_PER_CTU];
    fill_ctu(pic, addr, cur_pp, cs, refbuf, slicebuf);
    if (api.syntax_add_ctu(syn, addr, &cs)) fail("hevcrecon_syntax_add_ctu");
  }

  // Where the reference reconstructed the CTU it had just parsed (TLibDecoder/TDecSlice.cpp:366-367)
  void ctu_parsed(TComDataCU *cu) {
    const double t0 = now();
    TComPic *pic = cu->getPic();
    if (pic != cur) begin_picture(pic);
    if (cur_ok) {
      const int addr = (int)cu->getAddr();
      if ((int)pic->getPicSym()->getInverseCUOrderMap(addr) != cur_added) {
        cur_ok = false;  // a CTU twice (a slice that ends inside it) or one missing: pack from the arrays at the end
      } else {
        add_one_ctu(pic, addr);
        cur_added++;
        tm.ctus++;
      }
    }
    tm.pack += now() - t0;
  }

  // The picture is complete (TDecGop::filterPicture): the rest of the hand-over, then the GPU takes over.
  void finish_picture(TComPic *pic) {
    double t0 = now();
    const int nctu = pic->getNumCUsInFrame();
    if (cur != pic || !cur_ok || cur_added != nctu) {
      begin_picture(pic);
      for (int i = 0; i < nctu; i++) add_one_ctu(pic, (int)pic->getPicSym()->getCUOrderMap(i));  // coding order (tile scan)
      tm.repacked++;
      tm.pack += now() - t0;
      t0 = now();
    }
    hevcrecon_pic_params pp;
    fill_pic(pic, pp);
    check_slices(pic);
    hevcrecon_scaling_lists sl;
    if (fill_scaling(pic, sl) && api.syntax_set_scaling_lists(syn, &sl)) fail("hevcrecon_syntax_set_scaling_lists");
    hevcrecon_pred_weights pw;
    if (fill_weights(pic, pw) && api.syntax_set_pred_weights(syn, &pw)) fail("hevcrecon_syntax_set_pred_weights");
    std::vector<hevcrecon_slice_params> slices;
    if (fill_slices(pic, pp, slices) && api.syntax_set_slices(syn, (int)slices.size(), &slices[0])) fail("hevcrecon_syntax_set_slices");
    if (api.syntax_finish(syn, &pp)) fail("hevcrecon_syntax_finish");
    cur = NULL;
    tm.finish += now() - t0;
    if (mode == MODE_ENGINE) {
      t0 = now();
      const int id = pic_id(pic);
      submit_hashes_reading(id);  // a picture still waiting for its hash must not be overwritten before the hash is queued
      if (api.reconstruct(ctx, syn, id)) fail("hevcrecon_reconstruct");  // queued: the GPU works from here on
      pending = pic;
      tm.submit += now() - t0;
    } else if (mode == MODE_HUB) {
      t0 = now();
      if (api.hub_submit(hub, syn, pic_id(pic))) fail("hevcrecon_hub_submit");
      tm.submit += now() - t0;
    }
    tm.pictures++;
    tm.t_last = now();
    tm.wall_last = wall_clock();
  }

  // The picture is fetched only when the decoder is about to need it: after TComPic::compressMotion
  // (TLibDecoder/TDecGop.cpp:210; TLibCommon/TComMotionInfo.cpp:330-351), so that the host's sub-sampling of the
  // motion field runs beside the GPU's reconstruction instead of after it.  The packer copied the motion vectors
  // when the CTUs were added, before compressMotion changes them.
  TComPic *pending = NULL;
  TComPicYuv *lost_src_yuv = NULL, *lost_dst_yuv = NULL;  // the copy TDecTop::xCreateLostPicture has just made
  void engine_fetch(TComPic *pic) {
    if (pending != pic) return;
    pending = NULL;
    if (device_hash) return;  // the picture stays on the GPU
    double t0 = now();
    if (api.synchronize(ctx)) fail("hevcrecon_synchronize");
    tm.wait += now() - t0;
    t0 = now();
    TComPicYuv *yuv = pic->getPicYuvRec();
    if (api.picture_download(ctx, pic_id(pic), yuv->getLumaAddr(), yuv->getCbAddr(), yuv->getCrAddr(), yuv->getStride(),
                             yuv->getCStride()))
      fail("hevcrecon_picture_download");
    tm.download += now() - t0;
    tm.t_last = now();
    tm.wall_last = wall_clock();
  }

  // ---- HEVCRECON_HASH=device: status lines wait for their digests -----------------------------------------------
  std::deque<PendingLine> lines;     // in decoding order
  std::deque<int64_t> tickets;       // hash requests in flight, oldest first
  std::deque<int> ticket_sizes;

  // The oldest hash request: its digests, if they are back (or once they are, with `block`), printed behind their lines.
  bool collect_front(bool block) {
    const int n = ticket_sizes.front();
    std::vector<uint8_t> dig((size_t)48 * n);
    int done = 0;
    if (block) {
      if (api.md5_wait(ctx, tickets.front(), &dig[0])) fail("hevcrecon_picture_md5_wait");
      done = 1;
    } else if (api.md5_poll(ctx, tickets.front(), &done, &dig[0])) {
      fail("hevcrecon_picture_md5_poll");
    }
    if (!done) return false;
    while (!lines.empty() && (!lines.front().check || lines.front().ticket == tickets.front())) {
      unsigned char got[3][16];
      if (lines.front().check) memcpy(got, &dig[(size_t)48 * lines.front().index], 48);
      print_line(lines.front(), got);
      lines.pop_front();
    }
    tickets.pop_front();
    ticket_sizes.pop_front();
    return true;
  }

  void submit_hashes() {
    while (tickets.size() >= 32) collect_front(true);  // the engine keeps track of 64 requests
    std::vector<int> ids;
    for (size_t i = 0; i < lines.size(); i++)
      if (lines[i].check && lines[i].ticket < 0 && lines[i].index < 0) ids.push_back(lines[i].pic_id);
    if (ids.empty()) return;
    int64_t t = -1;
    if (api.md5_submit(ctx, (int)ids.size(), &ids[0], &t)) fail("hevcrecon_picture_md5_submit");
    int k = 0;
    for (size_t i = 0; i < lines.size(); i++)
      if (lines[i].check && lines[i].ticket < 0 && lines[i].index < 0) {
        lines[i].ticket = t;
        lines[i].index = k++;
      }
    tickets.push_back(t);
    ticket_sizes.push_back(k);
    tm.hash_batches++;
  }
  void submit_hashes_reading(int id) {
    if (mode == MODE_HUB) return;  // the hub hashes every picture right behind its reconstruction
    for (size_t i = 0; i < lines.size(); i++)
      if (lines[i].check && lines[i].ticket < 0 && lines[i].index < 0 && lines[i].pic_id == id) {
        submit_hashes();
        return;
      }
  }
  // The verdict of TDecGop.cpp:294-320, printed behind the line it belongs to
  void print_line(const PendingLine &l, unsigned char got[3][16]) {
    fputs(l.text.c_str(), stdout);
    tm.t_last = now();  // a picture is done when its verdict is
    tm.wall_last = wall_clock();
    if (!l.hash_field) return;
    if (!l.check) {
      printf("[:,(unk)] ");
      return;
    }
    const bool ok = !memcmp(got, l.want, 48);
    printf("[MD5:%s,%s] ", digestToString(got, 16), ok ? "(OK)" : "(***ERROR***)");
    if (!ok) {
      g_md5_mismatch = true;
      unsigned char rx[3][16];
      memcpy(rx, l.want, 48);
      printf("[rxMD5:%s] ", digestToString(rx, 16));
    }
  }
  // Collects what has come back (all of it with `block`), prints the lines that are complete, in order, and keeps one
  // hash request in flight: the batch size follows the decoder's speed by itself.
  void pump_hashes(bool block) {
    const double t0 = now();
    if (mode == MODE_HUB) {  // one digest per picture comes back from the hub, in order
      while (!lines.empty()) {
        unsigned char got[3][16];
        int have = 0;
        if (block) {
          if (api.hub_wait(hub, &got[0][0])) fail("hevcrecon_hub_wait");
          have = 1;
        } else if (api.hub_poll(hub, &have, &got[0][0])) {
          fail("hevcrecon_hub_poll");
        }
        if (!have) break;
        print_line(lines.front(), got);
        lines.pop_front();
      }
      tm.hash_device += now() - t0;
      return;
    }
    while (!tickets.empty() && collect_front(block)) {
    }
    if (tickets.empty()) {
      while (!lines.empty() && !lines.front().check) {  // lines without a hash to wait for
        unsigned char none[3][16];
        print_line(lines.front(), none);
        lines.pop_front();
      }
      submit_hashes();
      if (block && !tickets.empty()) {
        tm.hash_device += now() - t0;
        pump_hashes(true);
        return;
      }
    }
    tm.hash_device += now() - t0;
  }

  void report() {
    if (!tm.on || !tm.pictures) return;
    const double span = tm.t_last - tm.t_first;
    const double ours = tm.pack + tm.finish + tm.submit + tm.motion + tm.wait + tm.download + tm.md5_host + tm.hash_device;
    fprintf(stderr,
            "hm_bridge timing: {\"mode\": \"%s\", \"hash\": \"%s\", \"pictures\": %ld, \"ctus\": %ld, \"repacked\": %ld, \"hash_batches\": %ld, "
            "\"startup_s\": %.6f, \"engine_init_s\": %.6f, \"first_to_last_s\": %.6f, \"wall_first\": %.6f, \"wall_last\": %.6f, "
            "\"pack_s\": %.6f, \"finish_s\": %.6f, \"submit_s\": %.6f, \"motion_s\": %.6f, \"wait_s\": %.6f, \"download_s\": %.6f, "
            "\"md5_host_s\": %.6f, \"hash_device_s\": %.6f, \"parse_and_rest_s\": %.6f}\n",
            mode == MODE_ENGINE ? "engine" : mode == MODE_HUB ? "hub" : "pack", device_hash ? "device" : "host", tm.pictures, tm.ctus, tm.repacked, tm.hash_batches,
            tm.t_first - tm.t_process, tm.engine_init, span, tm.wall_first, tm.wall_last, tm.pack, tm.finish, tm.submit, tm.motion, tm.wait,
            tm.download, tm.md5_host, tm.hash_device, span - ours);
  }
};

Bridge &bridge() {
  static Bridge b;
  return b;
}

}  // namespace

Void TDecCu::decompressCU(TComDataCU *pcCU) {
  Bridge &b = bridge();
  if (b.on_gpu()) {  // reconstruction is deferred to the engine; the CTU's syntax is handed over now
    b.ctu_parsed(pcCU);
    return;
  }
  decompressCU_ref(pcCU);
  if (b.mode == MODE_PACK) b.ctu_parsed(pcCU);
}

Void TComLoopFilter::loopFilterPic(TComPic *pcPic) {
  Bridge &b = bridge();
  if (b.on_gpu()) {
    b.finish_picture(pcPic);
    return;
  }
  if (b.mode == MODE_PACK) b.finish_picture(pcPic);
  if (b.mode == MODE_DUMP) b.capture_stage(pcPic, 0);
  loopFilterPic_ref(pcPic);
  if (b.mode == MODE_DUMP) b.capture_stage(pcPic, 1);
}

Void TComSampleAdaptiveOffset::SAOProcess(SAOParam *pcSaoParam) {
  if (bridge().on_gpu()) return;  // already applied by the engine
  SAOProcess_ref(pcSaoParam);
}

Void TComSampleAdaptiveOffset::PCMLFDisableProcess(TComPic *pcPic) {
  // the engine has already put lossless coding units back; the reference would copy from per-CU buffers that
  // its skipped reconstruction never filled
  if (bridge().on_gpu()) return;
  PCMLFDisableProcess_ref(pcPic);
}

Void TComPic::compressMotion() {
  Bridge &b = bridge();
  if (b.mode == MODE_DUMP) {
    b.capture_stage(this, 2);
    b.write_picture(this);  // before the motion field is sub-sampled
  }
  const double t0 = now();
  compressMotion_ref();
  b.tm.motion += now() - t0;
  if (b.mode == MODE_ENGINE) b.engine_fetch(this);
}

void calcMD5(TComPicYuv &pic, UChar digest[3][16]) {
  const double t0 = now();
  calcMD5_ref(pic, digest);
  bridge().tm.md5_host += now() - t0;
}

// Host-side border padding of reference pictures (TLibCommon/TComSlice.cpp:374-400 calls it for every reference of every
// slice): only the reference's own motion compensation reads the border, and in engine mode that never runs.
Void TComPicYuv::extendPicBorder() {
  if (bridge().on_gpu()) return;
  extendPicBorder_ref();
}

// TDecTop::xCreateLostPicture (TLibDecoder/TDecTop.cpp:214-261) conceals a lost reference with a copy of the nearest
// picture (:244).  The copy is noted here (the reference chooses source and destination) and repeated in the engine's
// picture buffer, device to device (hevcrecon_picture_copy), as soon as the decoder's picture list comes by (below).
Void TComPicYuv::copyToPic(TComPicYuv *pcPicYuvDst) {
  Bridge &b = bridge();
  if (b.on_gpu()) {
    b.lost_src_yuv = this;
    b.lost_dst_yuv = pcPicYuvDst;
    if (b.device_hash) return;  // no host copy of any picture exists
  }
  copyToPic_ref(pcPicYuvDst);
}

// Called in a loop by TDecTop::xDecodeSlice until no reference is missing any more (TLibDecoder/TDecTop.cpp:351-356),
// with the decoder's picture list: the call after a TDecTop::xCreateLostPicture finds the two pictures of the copy
// noted above in that list and has the engine make the same copy.
Int TComSlice::checkThatAllRefPicsAreAvailable(TComList<TComPic *> &rcListPic, TComReferencePictureSet *pReferencePictureSet, Bool printErrors,
                                               Int pocRandomAccess) {
  Bridge &b = bridge();
  if (b.on_gpu() && b.lost_dst_yuv) {
    TComPic *src = NULL, *dst = NULL;
    for (TComList<TComPic *>::iterator it = rcListPic.begin(); it != rcListPic.end(); ++it) {
      if ((*it)->getPicYuvRec() == b.lost_src_yuv) src = *it;
      if ((*it)->getPicYuvRec() == b.lost_dst_yuv) dst = *it;
    }
    b.lost_src_yuv = b.lost_dst_yuv = NULL;
    if (!src || !dst || b.pic_ids.find(src) == b.pic_ids.end()) {
      fprintf(stderr, "hm_bridge: concealment from a picture the engine never reconstructed\n");
      exit(EXIT_FAILURE);
    }
    b.submit_hashes_reading(b.pic_id(dst));
    if (b.mode == MODE_HUB) {
      if (b.api.hub_copy(b.hub, b.pic_id(dst), b.pic_id(src), dst->getSlice(0)->getPOC())) b.fail("hevcrecon_hub_copy");
    } else if (b.api.picture_copy(b.ctx, b.pic_id(dst), b.pic_id(src), dst->getSlice(0)->getPOC())) {
      b.fail("hevcrecon_picture_copy");
    }
  }
  return checkThatAllRefPicsAreAvailable_ref(rcListPic, pReferencePictureSet, printErrors, pocRandomAccess);
}

Void TDecGop::filterPicture(TComPic *&rpcPic) {
  Bridge &b = bridge();
  const SEImessages *seis = rpcPic->getSEIs();
  const bool md5_sei = seis && seis->picture_digest && seis->picture_digest->method == SEIDecodedPictureHash::MD5;
  if (!(b.on_gpu() && b.device_hash && (b.mode == MODE_HUB || md5_sei || !(seis && seis->picture_digest) || !m_decodedPictureHashSEIEnabled))) {
    if (b.on_gpu() && b.device_hash) b.pump_hashes(true);  // keep the lines in order around this picture
    filterPicture_ref(rpcPic);
    return;
  }
  // The reference's picture finish (TLibDecoder/TDecGop.cpp:171-242) with the in-loop filters on the GPU and the hash
  // check deferred: hand-over and reconstruction are queued, the motion field is sub-sampled beside them, the status
  // line is kept until the digests are back.
  long before = clock();
  b.finish_picture(rpcPic);
  rpcPic->compressMotion();
  TComSlice *sl = rpcPic->getSlice(0);
  char c = sl->isIntra() ? 'I' : sl->isInterP() ? 'P' : 'B';
  if (!sl->isReferenced()) c += 32;
  char buf[256];
  PendingLine line;
  m_dDecTime += (double)(clock() - before) / CLOCKS_PER_SEC;
  snprintf(buf, sizeof(buf), "\nPOC %4d TId: %1d ( %c-SLICE, QP%3d ) [DT %6.3f] ", sl->getPOC(), sl->getTLayer(), c, sl->getSliceQp(), m_dDecTime);
  line.text = buf;
  m_dDecTime = 0;
  for (int l = 0; l < 2; l++) {
    snprintf(buf, sizeof(buf), "[L%d ", l);
    line.text += buf;
    for (int r = 0; r < sl->getNumRefIdx(RefPicList(l)); r++) {
      snprintf(buf, sizeof(buf), "%d ", sl->getRefPOC(RefPicList(l), r));
      line.text += buf;
    }
    line.text += "] ";
  }
  line.hash_field = m_decodedPictureHashSEIEnabled != 0;
  line.check = m_decodedPictureHashSEIEnabled && md5_sei;
  line.pic_id = b.pic_id(rpcPic);
  line.ticket = -1;
  line.index = -1;
  memset(line.want, 0, sizeof(line.want));
  if (line.check) memcpy(line.want, seis->picture_digest->digest, 48);
  if (!m_decodedPictureHashSEIEnabled && b.mode != MODE_HUB) {  // the reference prints no hash field at all then
    fputs(line.text.c_str(), stdout);
  } else {
    b.lines.push_back(line);
    b.pump_hashes(false);
  }
  rpcPic->setOutputMark(true);
  rpcPic->setReconMark(true);
  m_sliceStartCUAddress.clear();
  m_LFCrossSliceBoundaryFlag.clear();
}

Void TDecTop::deletePicBuffer() {
  Bridge &b = bridge();
  if (b.on_gpu() && b.device_hash) b.pump_hashes(true);
  if (b.hub) {
    b.api.hub_disconnect(b.hub);
    b.hub = NULL;
  }
  b.report();
  deletePicBuffer_ref();
}
