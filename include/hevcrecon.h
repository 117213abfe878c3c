/*
 * hevcrecon.h — C ABI of the MI355X picture-reconstruction engine.
 *
 * The engine replaces, inside an otherwise unchanged HM 9.1 decoder, the
 * sample-domain half of decoding: de-quantisation + inverse transform, inter
 * motion compensation, intra prediction + residual add, deblocking and SAO.
 * Bitstream parsing (CABAC, MV derivation, DPB management) stays in the host
 * decoder; it hands the engine the per-CTU syntax arrays it has just parsed.
 *
 * The reference has no plugin interface: the boundary is four C++ call sites.
 * Each entry point below names the reference call it stands in for
 * (paths relative to source/Lib of the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     hevcrecon_last_error() returns a message for the last failure.
 *   - the caller owns all host buffers; the engine copies what it needs during
 *     the call and owns all device memory.
 *   - a context is used from one thread at a time.  Work is queued on the
 *     context's HIP stream; hevcrecon_picture_download() and
 *     hevcrecon_synchronize() are the synchronisation points.
 *   - pictures are named by small non-negative integers chosen by the caller
 *     (the decoder uses one id per TComPic buffer it recycles).
 */
#ifndef HEVCRECON_H
#define HEVCRECON_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEVCRECON_MAX_REFS 16
#define HEVCRECON_PARTS_PER_CTU 256 /* 64x64 CTU in 4x4 partitions, z-order */

typedef struct hevcrecon_ctx hevcrecon_ctx;       /* one engine instance on one GPU */
typedef struct hevcrecon_syntax hevcrecon_syntax; /* packed syntax of one picture (host staging + HBM copy) */

/* Sequence-level parameters: what TDecTop::xActivateParameterSets
 * (TLibDecoder/TDecTop.cpp:264-293) hands to m_cSAO.create / m_cLoopFilter.create,
 * plus the globals g_bitDepthY/C, g_uiMaxCUWidth, g_uiMaxCUDepth (TLibCommon/TComRom.cpp:93-96,362-363)
 * that the reference reads everywhere on the path. */
typedef struct hevcrecon_seq_params {
  int32_t width;                  /* luma samples, multiple of 8 */
  int32_t height;
  int32_t bit_depth_luma;         /* 8..12 */
  int32_t bit_depth_chroma;
  int32_t ctu_size;               /* 64 */
  int32_t max_cu_depth;           /* g_uiMaxCUDepth incl. g_uiAddCUDepth: ctu_size >> max_cu_depth == 4 */
  int32_t strong_intra_smoothing; /* sps->getUseStrongIntraSmoothing() */
  /* IPCM coding units (TDecCu::xReconPCM, TLibDecoder/TDecCu.cpp:845-871); all zero when the SPS has PCM off */
  int32_t pcm_bit_depth_luma;      /* sps->getPCMBitDepthLuma() */
  int32_t pcm_bit_depth_chroma;    /* sps->getPCMBitDepthChroma() */
  int32_t pcm_loop_filter_disable; /* sps->getUsePCM() && sps->getPCMFilterDisableFlag() */
  int32_t reserved[6];
} hevcrecon_seq_params;

/* Picture-level parameters, and the slice-level ones of a picture whose slices all agree in them (pictures whose
 * slices differ add a table with hevcrecon_syntax_set_slices).
 * Sources: TComSlice getters used by TDecCu.cpp:616-625,772-781 (chroma QP offsets),
 * TComLoopFilter.cpp:371,569-570,769 (deblocking), TDecGop.cpp:192-196 (SAO enables),
 * TComSlice::getRefPic / getRefPOC (reference lists). */
typedef struct hevcrecon_pic_params {
  int32_t poc;
  int32_t slice_type;             /* 0 = B, 1 = P, 2 = I  (TLibCommon/TypeDef.h:379-384) */
  int32_t deblocking_disable;
  int32_t beta_offset_div2;
  int32_t tc_offset_div2;
  int32_t cb_qp_offset_pps;
  int32_t cr_qp_offset_pps;
  int32_t cb_qp_delta_slice;
  int32_t cr_qp_delta_slice;
  int32_t sao_luma;               /* slice SAO enables; 0 when the SPS has SAO off */
  int32_t sao_chroma;
  int32_t num_ref[2];
  int32_t ref_pic_id[2][HEVCRECON_MAX_REFS]; /* engine picture ids of the reference pictures */
  int32_t ref_poc[2][HEVCRECON_MAX_REFS];
  /* pps->getConstrainedIntraPred(): intra blocks use only neighbours that are themselves intra
   * (TComPattern::is*Available, TLibCommon/TComPattern.cpp:653-830) */
  int32_t constrained_intra_pred;
  /* !pps->getLoopFilterAcrossTilesEnabledFlag(): deblocking leaves the edges between tiles alone and SAO treats
   * them like picture borders (m_bLFCrossTileBoundary, TLibCommon/TComLoopFilter.cpp:383,405,435;
   * m_bIndependentTileBoundaryForNDBFilter, TLibCommon/TComPic.cpp:170,304-327) */
  int32_t no_loop_filter_across_tiles;
  int32_t reserved[6];
} hevcrecon_pic_params;

/* What the in-loop filters and the chroma de-quantisation read from the slice a CTU belongs to: one entry per slice of
 * the picture.  Sources: TComLoopFilter.cpp:371-418,569-570,769 (pcCU->getSlice()->getDeblockingFilterDisable /
 * ...BetaOffsetDiv2 / ...TcOffsetDiv2 / getLFCrossSliceBoundaryFlag), TDecCu.cpp:616-625,772-781 (getSliceQpDeltaCb/Cr),
 * TDecGop.cpp:151-154 with TComDataCU::setNDBFilterBlockBorderAvailability (TLibCommon/TComDataCU.cpp:3813-4166) for SAO.
 * Slices end at CTU boundaries (the reference's encoder cannot end them elsewhere).  Slice type, reference lists and
 * prediction weights are not in the table: the slices of a picture must share them, and the SAO enables are those of
 * the picture's first slice, which is what the reference applies to the whole picture (TDecGop.cpp:183,194-196). */
typedef struct hevcrecon_slice_params {
  uint32_t start_addr;               /* slice->getSliceCurStartCUAddr(): what hevcrecon_ctu_syntax.slice_start holds */
  int32_t deblocking_disable;
  int32_t beta_offset_div2;
  int32_t tc_offset_div2;
  int32_t cb_qp_delta_slice;
  int32_t cr_qp_delta_slice;
  int32_t loop_filter_across_slices; /* slice->getLFCrossSliceBoundaryFlag() */
  int32_t reserved;
} hevcrecon_slice_params;

/* One CTU's parsed syntax: plain pointers to the TComDataCU member arrays
 * (TLibCommon/TComDataCU.h:122-204), HEVCRECON_PARTS_PER_CTU entries each, z-order.
 * mv[l] points at TComMv[256] viewed as int16 pairs (hor, ver) (TLibCommon/TComMv.h:51-55).
 * coeff[c] are the TCoeff (int32) arrays, 4096 luma / 1024 per chroma component,
 * z-order block offset 16*absPartIdx (4*absPartIdx chroma), raster inside a TU.
 * sao_* are this CTU's entries of SAOParam::saoLcuParam[c][addr] (TLibCommon/TypeDef.h:342-352),
 * merge flags already resolved by the parser; sao_type -1 = off. */
typedef struct hevcrecon_ctu_syntax {
  const uint8_t *depth;
  const int8_t  *pred_mode;       /* 0 inter, 1 intra, 15 none (outside picture) */
  const int8_t  *part_size;
  const uint8_t *tr_idx;
  const uint8_t *cbf[3];
  const int8_t  *qp;
  const uint8_t *luma_dir;
  const uint8_t *chroma_dir;
  const uint8_t *transform_skip[3];
  const int16_t *mv[2];
  const int8_t  *ref_idx[2];
  const uint8_t *transquant_bypass; /* may be NULL */
  const uint8_t *ipcm;              /* may be NULL */
  const int32_t *coeff[3];
  int32_t sao_type[3];
  int32_t sao_subtype[3];
  int32_t sao_offset[3][4];
  /* TComDataCU::getPCMSampleY/Cb/Cr (Pel arrays laid out like coeff[]: offset 16*absPartIdx luma, 4*absPartIdx
   * chroma, raster inside the coding unit); read only for partitions whose ipcm flag is set; may be NULL */
  const int16_t *pcm[3];
  /* TComDataCU::getSliceStartCU per partition: address (CTU address * 256 + z index) of the first partition of the
   * slice the partition belongs to.  Intra prediction does not use neighbours of other CTUs that lie before it
   * (TComDataCU::getPULeft / getPUAbove / ... with bEnforceSliceRestriction, TLibCommon/TComDataCU.cpp:1141-1589).
   * NULL: one slice.  Without a slice table (hevcrecon_syntax_set_slices) the slices of a picture agree in everything
   * hevcrecon_pic_params carries and filter across their boundaries. */
  const uint32_t *slice_start;
  /* Tiles (TComPicSym::getInverseCUOrderMap / getTileIdxMap, TLibCommon/TComPicSym.h): 1 + the CTU's address in
   * coding order (tile scan), 0 = the same as its raster address; and the index of its tile.  CTUs are always added
   * in coding order.  Intra prediction does not use neighbours of other CTUs that belong to another tile
   * (bEnforceTileRestriction, TLibCommon/TComDataCU.cpp:1141-1589).  Whether the in-loop filters cross tile
   * boundaries: hevcrecon_pic_params.no_loop_filter_across_tiles. */
  int32_t coding_addr_plus1;
  int32_t tile_id;
} hevcrecon_ctu_syntax;

/* Scaling lists in effect for a picture: the TComScalingList that TDecTop::xDecodeSlice has resolved from
 * SPS / PPS / default tables and hands to m_cTrQuant.setScalingListDec (TLibDecoder/TDecTop.cpp:548-561).
 * Matrices are raster (TComScalingList::getScalingListAddress); listId = (intra ? 0 : 3) + component;
 * a 16x16 (32x32) entry covers 2x2 (4x4) coefficients and the DC entry is replaced by dc16 / dc32
 * (TComTrQuant::processScalingListDec, TLibCommon/TComTrQuant.cpp:2577-2590); 32x32 has the two luma lists only. */
typedef struct hevcrecon_scaling_lists {
  uint8_t m4[6][16];
  uint8_t m8[6][64];
  uint8_t m16[6][64];
  uint8_t m32[2][64]; /* 0 intra luma, 1 inter luma */
  uint8_t dc16[6];
  uint8_t dc32[2];
} hevcrecon_scaling_lists;

/* Explicit weighted prediction of a picture's slice: the entries of TComSlice::m_weightPredTable
 * (wpScalingParam: uiLog2WeightDenom, iWeight, iOffset; TLibCommon/TComSlice.h:1104-1118, read through
 * TComSlice::getWpScaling by TComWeightPrediction::getWpScaling, TLibCommon/TComWeightPrediction.cpp:271-334),
 * indexed [list][reference index][component Y, Cb, Cr].  Entries without a transmitted weight hold the parser's
 * defaults (weight 1 << denominator, offset 0).  Offsets are the transmitted 8-bit-scale values. */
typedef struct hevcrecon_pred_weights {
  int16_t log2_denom[2][HEVCRECON_MAX_REFS][3];
  int16_t weight[2][HEVCRECON_MAX_REFS][3];
  int16_t offset[2][HEVCRECON_MAX_REFS][3];
} hevcrecon_pred_weights;

/* ---- context ---------------------------------------------------------- */

/* Creates an engine on HIP device `device`.  Fails (non-zero, *out = NULL) when no GPU is present. */
int hevcrecon_create(hevcrecon_ctx **out, int device);
void hevcrecon_destroy(hevcrecon_ctx *ctx);
/* No counterpart in the reference.  Packed pictures wait for their trip to the device in one pinned ring per context
 * (and its mirror in device memory), so that pictures finished one after the other and reconstructed in one batch
 * cross the host link in a few large copies.  Default 64 MB, allocated when the first picture is finished; call this
 * before that to choose another size (0: no ring, every syntax object stages in buffers of its own, which is also
 * what happens to a picture the ring has no room for).  A host that keeps many packed pictures alive sizes the
 * ring for all of them. */
int hevcrecon_set_upload_ring(hevcrecon_ctx *ctx, int64_t bytes);
const char *hevcrecon_last_error(const hevcrecon_ctx *ctx);
/* The same for a syntax object (the only place to ask when it was created detached): why hevcrecon_syntax_add_ctu or
 * hevcrecon_syntax_finish refused the picture -- the counterpart of the reference's asserts on parsed values
 * (e.g. TLibCommon/TComPrediction.cpp:336-338 on block sizes, :488 and :510 on reference indices). */
const char *hevcrecon_syntax_last_error(const hevcrecon_syntax *syn);

/* Replaces m_cSAO.create / m_cLoopFilter.create in TDecTop::xActivateParameterSets
 * (TLibDecoder/TDecTop.cpp:290-292).  Idempotent when the parameters are unchanged
 * (the reference calls its counterpart once per slice). */
int hevcrecon_sequence_init(hevcrecon_ctx *ctx, const hevcrecon_seq_params *sp);

/* ---- per-picture syntax ------------------------------------------------ */

/* A syntax object belongs to the context it was created on and must be destroyed before that context
 * (hevcrecon_syntax_destroy waits for the context's streams and gives its space in the upload ring back). */
int hevcrecon_syntax_create(hevcrecon_ctx *ctx, hevcrecon_syntax **out);
void hevcrecon_syntax_destroy(hevcrecon_syntax *syn);

/* Host-only syntax object (no GPU, no context): packs and can be inspected with hevcrecon_syntax_stats,
 * cannot be reconstructed.  No counterpart in the reference; it lets the host-side packing logic be
 * exercised on machines without a GPU. */
int hevcrecon_syntax_create_detached(const hevcrecon_seq_params *sp, hevcrecon_syntax **out);

/* Counters of the packed picture: out[0..3] coded transform blocks of size 4/8/16/32, [4] coded coefficients,
 * [5] intra prediction blocks, [6] intra CTU waves, [7] reference samples read by inter prediction,
 * [8] 4x4 records, [9] 4x4 records that are intra, [10] finished flag, [11] CTUs holding intra blocks,
 * [12] sum over those CTUs of their local dependency levels, [13] intra rounds on the critical path (slowest CTU
 * of every wave), [14] intra rounds in total, [15] dependency levels on the critical path,
 * [16..18] motion-compensation regions predicted as one block (64x64, 32x32, 16x16), [19] 8x8 blocks predicted
 * unit by unit (blocks of one motion outside those regions are predicted whole and not counted here), [20..22] reference blocks read by the regions of [16..18] (one per list in use),
 * [23..25] luma samples those regions cover inside the picture, [26..28] the same weighted by lists in use,
 * [29] bytes copied to the device for this picture, [30] runs the 4x4 records were coded into,
 * [31] bytes of the sparse coefficient stream. */
int hevcrecon_syntax_stats(const hevcrecon_syntax *syn, int64_t out[32]);

/* Start a new picture in `syn` (drops whatever it held).  Stands where TDecTop::xDecodeSlice
 * obtains the picture buffer for a first slice (TLibDecoder/TDecTop.cpp:358-376). */
int hevcrecon_syntax_begin(hevcrecon_syntax *syn);

/* Optional, between hevcrecon_syntax_begin and hevcrecon_syntax_finish: the picture de-quantises with these
 * scaling lists (TComTrQuant::xDeQuant, scaling-list branch, TLibCommon/TComTrQuant.cpp:1142-1173) instead of
 * the flat path.  Replaces m_cTrQuant.setScalingListDec / setUseScalingList(true) (TLibDecoder/TDecTop.cpp:560-561).
 * NULL returns to flat de-quantisation.  hevcrecon_syntax_begin resets to flat. */
int hevcrecon_syntax_set_scaling_lists(hevcrecon_syntax *syn, const hevcrecon_scaling_lists *lists);

/* Optional, between hevcrecon_syntax_begin and hevcrecon_syntax_finish: the picture's inter prediction uses explicit
 * weights (TComWeightPrediction::xWeightedPredictionUni / xWeightedPredictionBi instead of the plain rounding /
 * TComYuv::addAvg; TLibCommon/TComPrediction.cpp:519-542).  The caller passes them exactly when the reference would
 * weight: P slice with pps->getUseWP(), or B slice with pps->getWPBiPred(); the identical-motion shortcut
 * (TComPrediction::xCheckIdenticalMotion, :396-411) is then off for B slices, as in the reference.
 * NULL returns to default prediction.  hevcrecon_syntax_begin resets to default. */
int hevcrecon_syntax_set_pred_weights(hevcrecon_syntax *syn, const hevcrecon_pred_weights *weights);

/* Optional, any time between hevcrecon_syntax_begin and hevcrecon_syntax_finish (a decoder that adds CTUs while it
 * parses learns of a picture's slices as they arrive, TLibDecoder/TDecTop.cpp:295-600, and hands the table over when
 * the picture is complete): the picture's slices differ in the parameters of hevcrecon_slice_params, or do not all
 * filter across their boundaries.  `slices` holds one entry per slice (any order; copied; a later call replaces the
 * table).  Checked by hevcrecon_syntax_finish: every hevcrecon_ctu_syntax.slice_start value of the picture must be the
 * start_addr of one of them, and all partitions of a CTU must belong to one slice.  With a table, the deblocking
 * parameters and slice chroma QP deltas of hevcrecon_pic_params are not used.  n = 0 returns to one set of parameters
 * per picture; hevcrecon_syntax_begin does the same. */
int hevcrecon_syntax_set_slices(hevcrecon_syntax *syn, int n, const hevcrecon_slice_params *slices);

/* Replaces the call m_pcCuDecoder->decompressCU(pcCU) in TDecSlice::decompressSlice
 * (TLibDecoder/TDecSlice.cpp:367): instead of reconstructing the CTU at once, its syntax
 * is packed into the picture's work lists.  CTUs must arrive in coding (raster) order.
 * Must be called before TComPic::compressMotion (TLibDecoder/TDecGop.cpp:210) sub-samples the MVs. */
int hevcrecon_syntax_add_ctu(hevcrecon_syntax *syn, int ctu_addr, const hevcrecon_ctu_syntax *ctu);

/* Closes the picture and queues the host-to-device copy of the packed syntax. */
int hevcrecon_syntax_finish(hevcrecon_syntax *syn, const hevcrecon_pic_params *pp);

/* No counterpart in the reference: a packed picture as one position-independent block of bytes, so that parsing and
 * reconstruction may live in different processes.  The reference decoder is one thread with global state
 * (TLibCommon/TComRom.cpp:93-96), so a host that feeds one GPU from many streams runs one decoder process per stream;
 * a GPU serves only a few processes at a time, and pictures of different streams batch into one set of launches only
 * inside one context.  A decoder process that never opens the GPU packs into a host-only object
 * (hevcrecon_syntax_create_detached) and exports; the process that owns the GPU imports and reconstructs
 * (hevc_amd/csrc/hub.cpp is such a process).
 * hevcrecon_syntax_export: `syn` is finished; *bytes receives the size; with dst = NULL nothing else happens, else
 * cap must be at least that.  hevcrecon_syntax_import: stands for begin + add_ctu x N + finish on an object of a
 * context (or a host-only one) with the same sequence parameters; the block is checked (every index a kernel will
 * use lies inside its array) and refused with a message otherwise.  Reference picture ids are taken as they are:
 * exporter and importer agree on them. */
int hevcrecon_syntax_export(const hevcrecon_syntax *syn, void *dst, int64_t cap, int64_t *bytes);
int hevcrecon_syntax_import(hevcrecon_syntax *syn, const void *src, int64_t bytes);

/* ---- many decoder processes, one GPU process ------------------------------------------------ */

/* No counterpart in the reference.  The decoder-process side of the hub (hevc_amd/csrc/hub.cpp, started as
 * `hevcrecon_hub --name NAME --device N`): the decoder packs into a host-only syntax object and, where it would call
 * hevcrecon_reconstruct and hevcrecon_picture_md5_submit, hands the finished picture to the hub, which reconstructs
 * the pictures of all connected decoders in common batches and hashes them on the device.  The decoder process itself
 * never opens the GPU.  A connection has 256 picture ids of its own, hevcrecon_hub_pic_id_base() .. + 255: the
 * decoder adds the base to every id it names (destinations, and the reference ids in hevcrecon_pic_params).
 *   hevcrecon_hub_connect      waits for the hub to come up, takes a free slot; all pictures of a connection share `sp`
 *   hevcrecon_hub_submit       queues the finished picture `syn` for reconstruction into `dst_pic_id` (waits while
 *                              the connection's request ring is full); requests run in the order they are made
 *   hevcrecon_hub_copy         queues hevcrecon_picture_copy (concealment of a lost reference)
 *   hevcrecon_hub_poll / _wait the digests (3 x 16 bytes: Y, Cb, Cr) of the next picture in submission order
 *   hevcrecon_hub_disconnect   the hub finishes what is queued, then frees the connection's pictures
 * Failures of the hub on behalf of this connection (a picture it refused) surface from every later call: non-zero,
 * message in hevcrecon_hub_error(). */
typedef struct hevcrecon_hub hevcrecon_hub;
int hevcrecon_hub_connect(const char *name, const hevcrecon_seq_params *sp, hevcrecon_hub **out);
int hevcrecon_hub_pic_id_base(const hevcrecon_hub *hub);
int hevcrecon_hub_submit(hevcrecon_hub *hub, const hevcrecon_syntax *syn, int dst_pic_id);
int hevcrecon_hub_copy(hevcrecon_hub *hub, int dst_pic_id, int src_pic_id, int dst_poc);
int hevcrecon_hub_poll(hevcrecon_hub *hub, int *got, uint8_t digest[48]);
int hevcrecon_hub_wait(hevcrecon_hub *hub, uint8_t digest[48]);
const char *hevcrecon_hub_error(const hevcrecon_hub *hub);
void hevcrecon_hub_disconnect(hevcrecon_hub *hub);

/* No counterpart in the reference.  Marks an already finished picture for another trip to the device
 * (hevcrecon_syntax_finish marks it once): for hosts that keep packed pictures around, and for measuring the
 * hand-over.  The pictures of a batch that are so marked cross the host link when the batch is submitted: copy
 * commands from their pinned staging (pictures that lie back to back in the context's upload ring share one command,
 * cut into pieces of a few megabytes) on a copy stream that the contexts of a device share, beside the kernels of
 * the batches queued before; the batch's kernels wait for them once. */
int hevcrecon_syntax_upload(hevcrecon_syntax *syn);
int hevcrecon_syntax_upload_batch(int n, hevcrecon_syntax *const *syn);

/* ---- reconstruction ---------------------------------------------------- */

/* Replaces, for one picture, every decompressCU of its CTUs (TLibDecoder/TDecCu.cpp:142,404-466)
 * followed by TComLoopFilter::loopFilterPic (TLibDecoder/TDecGop.cpp:181) and
 * TComSampleAdaptiveOffset::SAOProcess (TLibDecoder/TDecGop.cpp:199).  The result is picture
 * `dst_pic_id` (created or overwritten), resident on the device and usable as a reference.
 * Reads only device memory; asynchronous. */
int hevcrecon_reconstruct(hevcrecon_ctx *ctx, hevcrecon_syntax *syn, int dst_pic_id);

/* Same for `n` mutually independent pictures (different streams, or pictures of one
 * stream none of which references another) with one set of kernel launches. */
int hevcrecon_reconstruct_batch(hevcrecon_ctx *ctx, int n, hevcrecon_syntax *const *syn, const int *dst_pic_ids);

/* No counterpart in the reference.  hevcrecon_reconstruct for callers that cannot allocate: tight serving loops,
 * harnesses that forbid hipMalloc, set-up code around a graph capture (the call itself still copies the job descriptor
 * from pinned memory and may wait for an earlier call's copy, so it is not itself meant to be captured).  Same
 * validation, same kernels, same result, but the
 * call allocates nothing on the device or the host: the job descriptor, the working picture, the residual planes and the
 * intra kernel's scheduling words lie in a fixed 16 MiB array in device memory that belongs to the library (one copy
 * per GPU, held until the process ends), and per-kernel timers and stage copies are off.
 *   - the destination picture must exist already (an earlier hevcrecon_reconstruct, hevcrecon_picture_upload or
 *     hevcrecon_picture_copy with that id created it); `syn` was finished before, which is where its own device
 *     buffers were allocated;
 *   - one picture per call, at most 1920x1088 luma samples (any bit depth): larger pictures are refused with a
 *     message, use hevcrecon_reconstruct for them;
 *   - every call on a GPU uses that GPU's one array: calls of one context are ordered by its stream, and a call
 *     from another context of the same process queues a wait for the previous user's last kernel first, so
 *     contexts that take turns are serialised on the device instead of racing (two processes on one GPU each have
 *     their own copy of the array).  The array keeps the last call's contents, and the call clears what has to
 *     start from zero itself. */
int hevcrecon_reconstruct_noalloc(hevcrecon_ctx *ctx, hevcrecon_syntax *syn, int dst_pic_id);

/* Copies the final picture into TComPicYuv-style planes (Pel = int16; strides in samples)
 * so that calcAndPrintHashStatus (TLibDecoder/TDecGop.cpp:235) and TVideoIOYuv::write keep working.
 * Blocks until the picture is complete. */
int hevcrecon_picture_download(hevcrecon_ctx *ctx, int pic_id, int16_t *y, int16_t *cb, int16_t *cr,
                               int stride_y, int stride_c);

/* Replaces TVideoIOYuv::write (TLibVideoIO/TVideoIOYuv.cpp:422-489) for a picture that is resident on the device:
 * `out` receives exactly the bytes the reference writes to its output file — Y, Cb, Cr planes of width x height
 * luma samples (the picture size minus the cropping the caller passes to write(); like the reference, the window
 * starts at the top left sample), scaled from the internal to the file bit depth (scalePlane / invScalePlane,
 * :62-126), one byte per sample, or two (low byte first) when either file depth exceeds 8 (writePlane, :305-340).
 * Only those bytes cross the host link.  Blocks until done. */
int hevcrecon_picture_download_file(hevcrecon_ctx *ctx, int pic_id, int file_bit_depth_luma, int file_bit_depth_chroma,
                                    int width, int height, void *out, int64_t out_bytes);

/* Replaces calcMD5 (TLibCommon/TComPicYuvMD5.cpp:188-213), the picture hash that TDecGop::calcAndPrintHashStatus
 * (TLibDecoder/TDecGop.cpp:235) compares with the SEI message, for n resident pictures at once: digests receives
 * n x 3 x 16 bytes (Y, Cb, Cr of each picture), and only those cross the host link.  MD5 is one serial chain per
 * plane, so a call takes tens of milliseconds whatever n is (one lane per plane): worth it for batches
 * (n up to 4096; a picture may be named more than once).  Blocks until done. */
int hevcrecon_picture_md5_batch(hevcrecon_ctx *ctx, int n, const int *pic_ids, uint8_t *digests);

/* The same hash without blocking the caller, for a decoder that checks every picture against its SEI message
 * (TDecGop::calcAndPrintHashStatus, TLibDecoder/TDecGop.cpp:233-236,255-321) and must not stall on it: the hash kernel
 * is queued on a stream of the context's own, behind everything queued so far and beside the reconstruction of later
 * pictures; *ticket names the request.  hevcrecon_picture_md5_poll sets *done and fills digests (n x 3 x 16 bytes, as
 * above) once the digests are back, hevcrecon_picture_md5_wait blocks until then; either collects a ticket once.  Up
 * to 64 tickets may be outstanding.  The pictures are hashed as they are when everything queued before the call has
 * run: the engine hashes copies of its own (made on the device, in order with the reconstruction kernels), so the
 * caller may reuse the picture ids at once, as a decoder does with its picture buffers. */
int hevcrecon_picture_md5_submit(hevcrecon_ctx *ctx, int n, const int *pic_ids, int64_t *ticket);
/* The same without the copies: the caller keeps the pictures as they are until it has collected the ticket (a host
 * that sets pictures aside itself, with hevcrecon_picture_copy, to hash many of them in one request). */
int hevcrecon_picture_md5_submit_in_place(hevcrecon_ctx *ctx, int n, const int *pic_ids, int64_t *ticket);
int hevcrecon_picture_md5_poll(hevcrecon_ctx *ctx, int64_t ticket, int *done, uint8_t *digests);
int hevcrecon_picture_md5_wait(hevcrecon_ctx *ctx, int64_t ticket, uint8_t *digests);

/* Stage-level parity checks: after hevcrecon_debug_keep_stages(ctx, 1) every reconstruct call also keeps a
 * copy of its first picture before deblocking (stage 0) and after deblocking, before SAO (stage 1). */
int hevcrecon_debug_keep_stages(hevcrecon_ctx *ctx, int on);
int hevcrecon_debug_download_stage(hevcrecon_ctx *ctx, int stage, int16_t *y, int16_t *cb, int16_t *cr,
                                   int stride_y, int stride_c);

/* Uploads host planes as picture `pic_id` (tests, and concealment of a lost reference:
 * TDecTop::xCreateLostPicture, TLibDecoder/TDecTop.cpp:214-261). */
int hevcrecon_picture_upload(hevcrecon_ctx *ctx, int pic_id, int poc, const int16_t *y, const int16_t *cb,
                             const int16_t *cr, int stride_y, int stride_c);

/* Replaces the picture copy in TDecTop::xCreateLostPicture (TLibDecoder/TDecTop.cpp:214-261; TComPicYuv::copyToPic,
 * :244), which conceals a lost reference with the nearest picture: picture `dst_pic_id` (created or overwritten)
 * becomes a copy of the resident picture `src_pic_id` and takes the POC `dst_poc`.  Device to device, asynchronous. */
int hevcrecon_picture_copy(hevcrecon_ctx *ctx, int dst_pic_id, int src_pic_id, int dst_poc);

/* Frees the device picture (TComPic buffers being destroyed: TLibDecoder/TDecTop.cpp:98-118). */
int hevcrecon_picture_release(hevcrecon_ctx *ctx, int pic_id);

/* Waits for everything queued on the context.  Also the place (with every call that brings a picture or its digests
 * back: hevcrecon_picture_download, _download_file, _md5_batch, _md5_poll / _wait) where a work list that a kernel
 * refused is reported: non-zero, message in hevcrecon_last_error. */
int hevcrecon_synchronize(hevcrecon_ctx *ctx);

/* ---- measurement ------------------------------------------------------- */

/* Per-kernel GPU time, measured with HIP events on the context's stream while profiling is on.
 * Kernel indices: see hevcrecon_kernel_name(). */
#define HEVCRECON_NUM_KERNELS 11
int hevcrecon_profile_enable(hevcrecon_ctx *ctx, int on);
int hevcrecon_profile_read(hevcrecon_ctx *ctx, double ms_out[HEVCRECON_NUM_KERNELS],
                           int64_t launches_out[HEVCRECON_NUM_KERNELS]);
const char *hevcrecon_kernel_name(int idx);

/* Algorithmic bytes of one picture as packed in `syn`:
 * S*b*(1 + 2 + 2*[SAO on]) + MC reads (w*h*3/2*b per reference used) + 2*coded coefficients + 16*(W/4)*(H/4). */
int64_t hevcrecon_syntax_algorithmic_bytes(const hevcrecon_syntax *syn);

#ifdef __cplusplus
}
#endif
#endif /* HEVCRECON_H */
