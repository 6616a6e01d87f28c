// Exercises the C++20 host facade (path_tracer_amd/include/pt/path_tracer.hpp): builds scenes with the
// reference-shaped constructors (the way src/main.cpp:67-161 does), then either dumps the flattened C-ABI
// tables (CPU-only check against the Python packer) or renders through pt_render_host (GPU check).
//
//   facade_main dump   <scene> <out.bin>
//   facade_main render <scene> <w> <h> <spp> <out.f32>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "pt/path_tracer.hpp"

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

static std::vector<hittable_t> zoo() { // every alternative of hittable_t / material_t / texture_t
  std::vector<hittable_t> h;
  texture_t checker = checker_texture(color{0.2f, 0.3f, 0.1f}, color{0.9f, 0.9f, 0.9f});
  h.emplace_back(sphere(point{0, -100.5f, -1}, 100, lambertian_material(checker)));
  h.emplace_back(sphere(point{0, 0, -1}, 0.5f, lambertian_material(color{0.7f, 0.3f, 0.3f})));
  h.emplace_back(sphere(point{1, 0, -1}, 0.5f, metal_material(color{0.8f, 0.6f, 0.2f}, 0.3f)));
  h.emplace_back(sphere(point{-1, 0, -1}, 0.5f, dielectric_material(1.5f, color{1, 1, 1})));
  h.emplace_back(sphere(point{0.3f, 0.1f, -0.2f}, point{0.3f, 0.3f, -0.2f}, 0.0f, 1.0f, 0.12f, lambertian_material(color{0.7f, 0.3f, 0.3f})));
  h.emplace_back(triangle(point{-0.5f, 0.6f, -1.2f}, point{0.5f, 0.6f, -1.2f}, point{0, 1.3f, -0.9f}, lambertian_material(color{0.1f, 0.2f, 0.9f})));
  h.emplace_back(box(point{1.2f, -0.5f, -2.5f}, point{1.8f, 0.7f, -1.9f}, metal_material(color{0.7f, 0.6f, 0.5f}, 7.0f)));
  h.emplace_back(constant_medium(sphere(point{0.8f, 0.9f, -1.2f}, 0.4f, lambertian_material(color{1, 1, 1})), 3.0f, color{0.9f, 0.9f, 1.0f}));
  h.emplace_back(xz_rect(-1, 1, -2, 0, 2.5f, lightsource_material(color{4, 4, 4})));
  h.emplace_back(yz_rect(-0.5f, 1.5f, -2.5f, -0.5f, -2.2f, lambertian_material(color{0.2f, 0.8f, 0.2f})));
  h.emplace_back(xy_rect(-2, -1, -0.5f, 1, -1.5f, lambertian_material(color{0.7f, 0.3f, 0.3f})));
  h.emplace_back(constant_medium(box(point{-1.9f, -0.5f, -0.9f}, point{-1.3f, 0.2f, -0.3f}, lambertian_material(color{1, 1, 1})), 5.0f, checker));
  return h;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: see the file header\n"); return 2; }
  std::string mode = argv[1], name = argv[2];
  std::vector<hittable_t> hittables = name == "cornell" ? cornell() : zoo();
  try {
    if (mode == "dump") {
      scene_tables t = flatten(hittables);
      camera cam = name == "cornell" ? camera(point{278, 278, -800}, point{278, 278, 0}, vec{0, 1, 0}, 40, 64.0f / 36.0f, 0, 800, 0, 1)
                                     : camera(point{0.3f, 0.6f, 2.5f}, point{0, 0.2f, -1}, vec{0, 1, 0}, 50, 1.5f, 0.1f, 3.4f, 0, 1);
      std::ofstream f(argv[3], std::ios::binary);
      int32_t n[3] = {(int32_t)t.hittables.size(), (int32_t)t.materials.size(), (int32_t)t.textures.size()};
      f.write((const char*)n, sizeof n);
      f.write((const char*)t.hittables.data(), t.hittables.size() * sizeof(PtHittable));
      f.write((const char*)t.materials.data(), t.materials.size() * sizeof(PtMaterial));
      f.write((const char*)t.textures.data(), t.textures.size() * sizeof(PtTexture));
      f.write((const char*)&cam.c, sizeof cam.c);
      return 0;
    }
    if (mode == "render" && argc >= 7) {
      int w = std::atoi(argv[3]), h = std::atoi(argv[4]), spp = std::atoi(argv[5]);
      camera cam = name == "cornell" ? camera(point{278, 278, -800}, point{278, 278, 0}, vec{0, 1, 0}, 40, float(w) / h, 0, 800, 0, 1)
                                     : camera(point{0.3f, 0.6f, 2.5f}, point{0, 0.2f, -1}, vec{0, 1, 0}, 50, float(w) / h, 0.1f, 3.4f, 0, 1);
      frame_buffer fb;
      render(w, h, spp, fb, hittables, cam);
      std::ofstream f(argv[6], std::ios::binary);
      f.write((const char*)fb.data(), fb.size() * sizeof(color));
      return 0;
    }
  } catch (const pt_error& e) {
    std::fprintf(stderr, "pt_error %d: %s\n", e.code, e.what());
    return 3;
  }
  return 2;
}
