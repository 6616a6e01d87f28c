// Exercises the C++ N-GPU host path (path_tracer_amd/include/pt/distributed.hpp + include/pt_dist.h) without Python:
//
//   dist_main multi  <w> <h> <spp> <out.f32> <ndev>   pt::render_multi_gpu over devices 0..ndev-1: ncclCommInitAll, one
//                                                      thread per GPU, RCCL gather to device 0, un-interleave, copy back
//   dist_main shards <w> <h> <spp> <out.f32> <n>      what n ranks do, replayed on ONE GPU without a communicator: every
//                                                      shard rendered by pt_render into its slot of the root's gather
//                                                      buffer (where ncclGather would put it), then pt_unshard_tiles
// The scene is the Cornell-style box of the facade test (camera aspect = w / h).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <numeric>
#include <string>

#include "pt/distributed.hpp"

using namespace pt;

static std::vector<hittable_t> cornell() {
  material_t white = lambertian_material(color{0.73f, 0.73f, 0.73f});
  material_t red = lambertian_material(color{0.65f, 0.05f, 0.05f});
  material_t green = lambertian_material(color{0.12f, 0.45f, 0.15f});
  material_t light = lightsource_material(color{15.0f, 15.0f, 15.0f});
  std::vector<hittable_t> h;
  h.emplace_back(box(point{555, 0, 0}, point{556, 555, 555}, green));
  h.emplace_back(box(point{-1, 0, 0}, point{0, 555, 555}, red));
  h.emplace_back(box(point{213, 554, 227}, point{343, 554.5f, 332}, light));
  h.emplace_back(box(point{0, -1, 0}, point{555, 0, 555}, white));
  h.emplace_back(box(point{0, 555, 0}, point{555, 556, 555}, white));
  h.emplace_back(xy_rect(0, 555, 0, 555, 555, white));
  h.emplace_back(box(point{130, 0, 65}, point{295, 165, 230}, white));
  h.emplace_back(box(point{265, 0, 295}, point{430, 330, 460}, white));
  return h;
}

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: see the file header\n"); return 2; }
  const std::string mode = argv[1];
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), spp = std::atoi(argv[4]), n = std::atoi(argv[6]);
  std::vector<hittable_t> hittables = cornell();
  camera cam(point{278, 278, -800}, point{278, 278, 0}, vec{0, 1, 0}, 40, float(w) / h, 0, 800, 0, 1);
  frame_buffer fb;
  try {
    if (mode == "multi") {
      std::vector<int> devices((std::size_t)n);
      std::iota(devices.begin(), devices.end(), 0);
      render_multi_gpu(devices, w, h, spp, fb, hittables, cam);
    } else if (mode == "shards") {
      device_scene scene(hittables);
      PtRenderParams p{w, h, spp, 50, 0, n, 0, 0};
      const int64_t per = pt_framebuffer_floats(&p);
      detail::dev_buf ws((std::size_t)pt_dist_gather_floats(&p)), frame((std::size_t)w * h * 3);
      for (int r = 0; r < n; r++) {
        p.shard_index = r;
        check(pt_render(scene.s, &cam.c, &p, ws.p + (std::size_t)r * per, nullptr), "pt_render");
      }
      if (n > 1) check(pt_unshard_tiles(ws.p, &p, frame.p, nullptr), "pt_unshard_tiles");
      fb.resize((std::size_t)w * h);
      if (hipMemcpy(fb.data(), n > 1 ? frame.p : ws.p, fb.size() * sizeof(color), hipMemcpyDeviceToHost) != hipSuccess) return 4;
    } else return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
  std::ofstream f(argv[5], std::ios::binary);
  f.write((const char*)fb.data(), fb.size() * sizeof(color));
  return 0;
}
