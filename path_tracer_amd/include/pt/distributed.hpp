// distributed.hpp — the N-GPU path of the C++20 host, without Python: tiles round-robin over the GPUs of one node,
// one RCCL gather of the float tiles to the root over xGMI, device-side un-interleave (include/pt_dist.h).
//
//   render_sharded(comm, rank, nranks, ...)    one rank of a multi-process job (communicator from ncclCommInitRank)
//   render_multi_gpu(devices, ...)             one process driving several GPUs (ncclCommInitAll + a thread per GPU)
//
// Every pixel keeps its global seed (render.hpp:130-131), so the frame equals the single-GPU frame bit for bit.
// Link with -lpt_dist -lpt_render -lrccl -lamdhip64.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <mutex>
#include <thread>

#include "../../../include/pt_dist.h"
#include "path_tracer.hpp"

namespace pt {

namespace detail {
struct dev_buf {
  float* p = nullptr;
  explicit dev_buf(std::size_t floats) {
    if (floats && hipMalloc((void**)&p, floats * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
  }
  ~dev_buf() { if (p) (void)hipFree(p); }
  dev_buf(const dev_buf&) = delete;
  dev_buf& operator=(const dev_buf&) = delete;
};
inline void check_dist(int rc, const char* where) {
  if (rc != PT_OK) throw std::runtime_error(std::string(where) + ": " + pt_error_string(rc) + " (" + pt_dist_last_error() + ")");
}
} // namespace detail

// One rank's part: render shard `rank` of `nranks` on the CURRENT device, gather to `root`, and (on the root) copy the
// assembled frame to the host.  frame_buf is resized on the root only.
inline void render_sharded(ncclComm_t comm, int rank, int nranks, int width, int height, int samples, frame_buffer& frame_buf,
                           const std::vector<hittable_t>& hittables, const camera& cam, int depth = 50, int root = 0,
                           hipStream_t stream = nullptr) {
  device_scene scene(hittables);
  PtRenderParams p{width, height, samples, depth, rank, nranks, 0, 0};
  const int64_t per = pt_framebuffer_floats(&p);
  if (per < 0) throw pt_error(PT_ERR_INVALID_ARG, "pt_framebuffer_floats");
  const bool is_root = rank == root;
  detail::dev_buf local((std::size_t)per);
  detail::dev_buf ws(is_root && nranks > 1 ? (std::size_t)pt_dist_gather_floats(&p) : 0);
  detail::dev_buf fb(is_root ? (std::size_t)width * height * 3 : 0);
  detail::check_dist(pt_dist_render(scene.s, &cam.c, &p, (void*)comm, root, local.p, ws.p, fb.p, (void*)stream), "pt_dist_render");
  if (is_root) {
    frame_buf.resize((std::size_t)width * height);
    if (hipMemcpyAsync(frame_buf.data(), fb.p, frame_buf.size() * sizeof(color), hipMemcpyDeviceToHost, stream) != hipSuccess)
      throw std::runtime_error("hipMemcpy D2H failed");
  }
  if (hipStreamSynchronize(stream) != hipSuccess) throw std::runtime_error("hipStreamSynchronize failed");
}

// One process, several GPUs of the node: a communicator per device (ncclCommInitAll), a host thread per device.
inline void render_multi_gpu(const std::vector<int>& devices, int width, int height, int samples, frame_buffer& frame_buf,
                             const std::vector<hittable_t>& hittables, const camera& cam, int depth = 50) {
  const int n = (int)devices.size();
  if (n < 1) throw std::invalid_argument("render_multi_gpu: no device");
  std::vector<ncclComm_t> comms((std::size_t)n);
  if (ncclCommInitAll(comms.data(), n, devices.data()) != ncclSuccess) throw std::runtime_error("ncclCommInitAll failed");
  std::vector<std::string> errors((std::size_t)n);
  std::vector<std::thread> threads;
  // A rank that fails before it reaches the gather (hipSetDevice, scene upload, an allocation, a rejected parameter) would
  // leave its peers blocked inside the collective for ever: the failing thread aborts EVERY communicator, which ends the
  // peers' pending collectives, so that all threads can be joined and the first error is rethrown.
  std::atomic<bool> aborted{false};
  std::mutex abort_mutex;
  for (int r = 0; r < n; r++)
    threads.emplace_back([&, r] {
      try {
        if (hipSetDevice(devices[(std::size_t)r]) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        render_sharded(comms[(std::size_t)r], r, n, width, height, samples, frame_buf, hittables, cam, depth, 0);
      } catch (const std::exception& e) {
        errors[(std::size_t)r] = e.what();
        std::lock_guard<std::mutex> g(abort_mutex);
        if (!aborted.exchange(true))
          for (auto c : comms) (void)ncclCommAbort(c);
      }
    });
  for (auto& t : threads) t.join();
  if (!aborted.load())
    for (auto c : comms) (void)ncclCommDestroy(c);
  for (std::size_t r = 0; r < errors.size(); r++) // the rank that failed first names the cause; peers only report the abort
    if (!errors[r].empty() && errors[r].find("pt_dist_render") == std::string::npos) throw std::runtime_error("render_multi_gpu: rank " + std::to_string(r) + ": " + errors[r]);
  for (auto& e : errors) if (!e.empty()) throw std::runtime_error("render_multi_gpu: " + e);
}

} // namespace pt
