// pt_dist.cpp — libpt_dist.so: the RCCL exchange step of the N-GPU path behind a C ABI (include/pt_dist.h).
// Host code only (the un-interleave kernel lives in libpt_render.so: pt_unshard_tiles).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

#include "../../include/pt_dist.h"

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define PTD_HIP(expr)                                                                              \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define PTD_NCCL(expr)                                                                             \
  do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); } while (0)
} // namespace

extern "C" {

const char* pt_dist_last_error(void) { return g_err.c_str(); }

int64_t pt_dist_gather_floats(const PtRenderParams* p) {
  const int64_t per = pt_framebuffer_floats(p);
  return per < 0 ? -1 : per * p->shard_count;
}

int pt_dist_gather_frame(const float* local, const PtRenderParams* p, void* nccl_comm, int root, float* gather_ws,
                         float* fb, void* stream) {
  const int64_t per = pt_framebuffer_floats(p);
  if (per < 0 || !local) return fail(PT_ERR_INVALID_ARG, "pt_dist_gather_frame: bad params or NULL tiles");
  hipStream_t st = (hipStream_t)stream;
  if (p->shard_count == 1) { // one GPU: the tiles are the frame
    if (fb && fb != local) PTD_HIP(hipMemcpyAsync(fb, local, (size_t)per * sizeof(float), hipMemcpyDeviceToDevice, st));
    return PT_OK;
  }
  if (!nccl_comm) return fail(PT_ERR_INVALID_ARG, "pt_dist_gather_frame: NULL communicator");
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  int rank = -1, size = 0;
  PTD_NCCL(ncclCommUserRank(comm, &rank));
  PTD_NCCL(ncclCommCount(comm, &size));
  if (size != p->shard_count || rank != p->shard_index)
    return fail(PT_ERR_INVALID_ARG, "pt_dist_gather_frame: shard_index / shard_count must be the communicator's rank / size");
  if (root < 0 || root >= size) return fail(PT_ERR_INVALID_ARG, "pt_dist_gather_frame: bad root");
  if (rank == root && (!gather_ws || !fb)) return fail(PT_ERR_INVALID_ARG, "pt_dist_gather_frame: the root needs gather_ws and fb");
  // equal counts per rank (the last shard's missing tiles are zero padding): one gather, rank r lands at gather_ws[r * per]
  PTD_NCCL(ncclGather(local, gather_ws, (size_t)per, ncclFloat, root, comm, st));
  if (rank == root) {
    int rc = pt_unshard_tiles(gather_ws, p, fb, stream);
    if (rc) return fail(rc, std::string("pt_unshard_tiles: ") + pt_last_error());
  }
  return PT_OK;
}

int pt_dist_render(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p, void* nccl_comm, int root,
                   float* local, float* gather_ws, float* fb, void* stream) {
  if (!local) return fail(PT_ERR_INVALID_ARG, "pt_dist_render: NULL local tile buffer");
  int rc = pt_render(scene, cam, p, local, stream);
  if (rc) return fail(rc, std::string("pt_render: ") + pt_last_error());
  return pt_dist_gather_frame(local, p, nccl_comm, root, gather_ws, fb, stream);
}

} // extern "C"
