// kernels.h — launch interface between the engine (engine.cpp) and the kernels (kernels.hip).
#ifndef HEVCRECON_KERNELS_H
#define HEVCRECON_KERNELS_H
#include "common.h"

// Launches kernel class `kind` (K_* of common.h) for a batch of `n` pictures described by the device
// array `jobs`.  arg/arg2: K_TRANSFORM: log2 block size, max blocks of that size over the batch;
// K_OTHER: expansion of the run-length coded motion
// records of the pictures whose need_expand is set.
void hr_launch(int kind, const PicJob *jobs, int n, const SeqInfo &s, hipStream_t stream, int arg, int arg2);

// Intra prediction + residual of a batch, one launch: `list` holds n_items entries (picture of the batch << 16 |
// index into that picture's ictu[]) in an order in which every CTU follows the CTUs it reads from, or is NULL for
// a single picture (its ictu[] is in such an order already).  `ticket` and every picture's ctu_done words must be
// zero when the kernel starts.
// max_parallel: the most CTUs that can be in work at a time (the batch's largest wave); dense: the batch's pictures are
// coded mostly in intra blocks (the launch then gets more workgroups); inside: some picture of the batch is not a plain
// intra picture (its CTUs hold inter-predicted or IPCM samples that have to be staged).
void hr_launch_intra(const PicJob *jobs, const SeqInfo &s, const unsigned *list, int n_items, int max_parallel, bool dense, bool inside,
                     unsigned *ticket, hipStream_t stream);

// Converts picture planes to the bytes TVideoIOYuv::write puts into a file: width x height samples of each plane
// (chroma: half of each), `file_bd` bits deep (scaled from the internal depth with rounding and clipping, or shifted
// up), one byte per sample or two (little endian) when wide is set.  planes / dst: device memory.
void hr_launch_output(const void *const planes[3], void *dst, const SeqInfo &s, int width, int height, int file_bd_y, int file_bd_c,
                      bool wide, hipStream_t stream);

// MD5 of n planes (device array `planes`), 16 bytes each into `digests` (device memory).  One lane per plane: the
// hash is a serial chain, so this only pays with many planes in flight.
void hr_launch_md5(const Md5Plane *planes, int n, uint8_t *digests, hipStream_t stream);

#ifdef HEVCRECON_INTRA_STAMPS
void hr_intra_stamps_report();
#endif

#endif
