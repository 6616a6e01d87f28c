/*
 * pt_dist.h — C ABI of the N-GPU exchange step of the render() hot path (libpt_dist.so).
 *
 * The reference renders one frame on one device (render.hpp:141-160).  Here the frame's 8x8 tiles are dealt
 * round-robin to N GPUs (PtRenderParams.shard_index / shard_count, include/pt_render.h), every pixel keeps its GLOBAL
 * seed (render.hpp:130-131), so no rank needs anything from another while it renders; the ONLY communication is this:
 * one gather of the float tiles to the root over xGMI (RCCL ncclGather: each peer -> root transfer rides its own
 * point-to-point link) and a device-side un-interleave into the reference's frame layout on the root
 * (pt_unshard_tiles).  One process (or thread) per GPU, one ncclComm_t per rank, created by the caller:
 * ncclCommInitRank across processes, ncclCommInitAll inside one process.
 *
 * Kept out of libpt_render.so so that the single-GPU library does not depend on RCCL.  `nccl_comm` is an ncclComm_t
 * passed as void* (no RCCL type in the signature); `stream` a hipStream_t.  Errors: the PT_* codes of pt_render.h,
 * text from pt_dist_last_error().
 */
#ifndef PT_DIST_H
#define PT_DIST_H

#include "pt_render.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Floats of the root's receive buffer: shard_count x pt_framebuffer_floats(p).  -1 for invalid params. */
int64_t pt_dist_gather_floats(const PtRenderParams* p);

/* Exchange step.  local_tiles_device: this rank's [tiles_per_shard][64][3] tiles (what pt_render wrote for
 * shard_index == this rank).  On the root: gather_ws_device (pt_dist_gather_floats floats) receives all shards and
 * fb_device ([height][width][3]) the assembled frame; other ranks pass NULL for both.  Asynchronous on `stream`.
 * shard_count == 1: local tiles ARE the frame; it is copied to fb_device (if different).                       */
int pt_dist_gather_frame(const float* local_tiles_device, const PtRenderParams* p, void* nccl_comm, int root,
                         float* gather_ws_device, float* fb_device, void* stream);

/* pt_render of this rank's shard (p->shard_index must be the rank of `nccl_comm`, p->shard_count its size) into
 * local_tiles_device (pt_framebuffer_floats(p) floats), then pt_dist_gather_frame.                              */
int pt_dist_render(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p, void* nccl_comm, int root,
                   float* local_tiles_device, float* gather_ws_device, float* fb_device, void* stream);

const char* pt_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
