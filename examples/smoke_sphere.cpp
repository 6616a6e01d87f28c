// examples/smoke_sphere.cpp — the reference's src/main.cpp:61-197 rewritten against the C++20 facade: same scene literal
// (main.cpp:83 leaves the order of the two RNG calls among its constructor arguments to the compiler; here it is DEFINED
// as last-to-first, g++'s order, which reproduces the counters the SURVEY recorded from the reference's own build), same camera, same output stage (gamma 2, clamp, x256, rows flipped) into out.png (main.cpp:33-59)
// or, for any other file name, a binary PPM (main.cpp:17-31).  stb is not a dependency: pt/image_io.hpp decodes and writes.  The two
// image textures are loaded like main.cpp:133,145 does, through image_texture::image_texture_factory, from <images_dir>/Xilinx.jpg and
// <images_dir>/SYCL.png (default "../images" as in the reference; a directory holding the decoded-pixel exports Xilinx.ppm / SYCL.ppm
// of `python -m path_tracer_amd --export-textures DIR` works too); a file that cannot be loaded gets the reference's treatment — a
// message on stderr and the fallback texel.  images_dir "procedural" selects small generated stand-ins instead.
//
//   g++ -std=c++20 -O2 -ffp-contract=off -Ipath_tracer_amd/include examples/smoke_sphere.cpp -Lpath_tracer_amd -lpt_render \
//       -Wl,-rpath,$PWD/path_tracer_amd -Wl,-rpath,/opt/rocm/lib -o sycl-rt-mi355x
//   ./sycl-rt-mi355x [width height samples out.png|out.ppm [tables.bin|- [images_dir]]]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "pt/path_tracer.hpp"

using namespace pt;

// LocalPseudoRNG (rtweekend.hpp:33-57) for host-side scene construction, main.cpp:76
struct HostRNG {
  uint32_t s = 2463534242u; // xorshift.hpp:18
  float float_t() {
    s ^= s >> 7; s ^= s << 1; s ^= s >> 9;
    return (float)s * (1.0f / 4294967296.0f);
  }
  float float_t(float mn, float mx) { return mn + (mx - mn) * float_t(); }
  vec vec_t() { float a = float_t(), b = float_t(), c = float_t(); return {a, b, c}; }
  vec vec_t(float mn, float mx) { vec v = vec_t(); float sc = mx - mn; return {v.x() * sc + mn, v.y() * sc + mn, v.z() * sc + mn}; }
};

static std::vector<uint8_t> procedural_image(int w, int h, int seed) {
  std::vector<uint8_t> px((size_t)w * h * 3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint8_t* p = &px[((size_t)y * w + x) * 3];
      p[0] = (uint8_t)(x * 255 / std::max(1, w - 1));
      p[1] = (uint8_t)(y * 255 / std::max(1, h - 1));
      p[2] = (uint8_t)((((x / 8) + (y / 8) + seed) % 2) * 200 + 30);
    }
  return px;
}

int main(int argc, char** argv) {
  const int width = argc > 1 ? std::atoi(argv[1]) : 800, height = argc > 2 ? std::atoi(argv[2]) : 480; // CMakeLists.txt:44-54
  const int samples = argc > 3 ? std::atoi(argv[3]) : 100;                                            // main.cpp:186
  const char* out = argc > 4 ? argv[4] : "out.png"; // main.cpp:57
  const std::string images_dir = argc > 6 ? argv[6] : "../images";
  const bool procedural = images_dir == "procedural";

  // the reference's own files (main.cpp:133,145: "../images/Xilinx.jpg", "../images/SYCL.png"), decoded by pt/image_io.hpp; a directory
  // that only holds the decoded-pixel exports (`python -m path_tracer_amd --export-textures DIR`: Xilinx.ppm, SYCL.ppm) works too
  auto image_file = [&](const char* name, const char* exported) {
    const std::string a = images_dir + "/" + name, b = images_dir + "/" + exported;
    if (std::FILE* f = std::fopen(a.c_str(), "rb")) { std::fclose(f); return a; }
    if (std::FILE* f = std::fopen(b.c_str(), "rb")) { std::fclose(f); return b; }
    return a; // neither: the factory reports the reference's file name and falls back to texel 0
  };
  std::vector<hittable_t> hittables;
  texture_t t = checker_texture(color{0.2f, 0.3f, 0.1f}, color{0.9f, 0.9f, 0.9f});
  hittables.emplace_back(sphere(point{0, -1000, 0}, 1000, lambertian_material(t)));
  HostRNG rng;
  for (int a = -11; a < 11; a++) {
    for (int b = -11; b < 11; b++) {
      float choose_mat = rng.float_t();
      // main.cpp:83 leaves the order of its two draws unspecified; g++ evaluates the constructor's arguments last to
      // first, and that order reproduces the SURVEY's recorded counters of the reference: z first (scenes.py "rtl")
      float cz = b + 0.9f * rng.float_t();
      float cx = a + 0.9f * rng.float_t();
      point center(cx, 0.2f, cz);
      float dx = cx - 4.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;
      if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9f) {
        if (choose_mat < 0.4f) {
          vec p = rng.vec_t(), q = rng.vec_t();
          hittables.emplace_back(sphere(center, 0.2f, lambertian_material(color{p.x() * q.x(), p.y() * q.y(), p.z() * q.z()})));
        } else if (choose_mat < 0.8f) {
          vec p = rng.vec_t(), q = rng.vec_t();
          point center2(cx, 0.2f + rng.float_t(0, 0.25f), cz);
          hittables.emplace_back(sphere(center, center2, 0.0f, 1.0f, 0.2f, lambertian_material(color{p.x() * q.x(), p.y() * q.y(), p.z() * q.z()})));
        } else if (choose_mat < 0.95f) {
          vec albedo = rng.vec_t(0.5f, 1);
          float fuzz = rng.float_t(0, 0.5f);
          hittables.emplace_back(sphere(center, 0.2f, metal_material(albedo, fuzz)));
        } else {
          hittables.emplace_back(sphere(center, 0.2f, dielectric_material(1.5f, color{1.0f, 1.0f, 1.0f})));
        }
      }
    }
  }
  // pyramid main.cpp:113-126
  hittables.emplace_back(triangle(point{6.5f, 0.0f, 1.30f}, point{6.25f, 0.50f, 1.05f}, point{6.5f, 0.0f, 0.80f}, lambertian_material(color(0.68f, 0.50f, 0.1f))));
  hittables.emplace_back(triangle(point{6.0f, 0.0f, 1.30f}, point{6.25f, 0.50f, 1.05f}, point{6.5f, 0.0f, 1.30f}, lambertian_material(color(0.89f, 0.73f, 0.29f))));
  hittables.emplace_back(triangle(point{6.5f, 0.0f, 0.80f}, point{6.25f, 0.50f, 1.05f}, point{6.0f, 0.0f, 0.80f}, lambertian_material(color(0.0f, 0.0f, 1))));
  hittables.emplace_back(triangle(point{6.0f, 0.0f, 0.80f}, point{6.25f, 0.50f, 1.05f}, point{6.0f, 0.0f, 1.30f}, lambertian_material(color(0.0f, 0.0f, 1))));
  hittables.emplace_back(sphere(point{4, 1, 0}, 0.2f, lightsource_material(color(10, 0, 10))));
  if (procedural) { auto xil = procedural_image(256, 128, 0); t = image_texture::from_rgb8(xil.data(), 256, 128); }
  else t = image_texture::image_texture_factory(image_file("Xilinx.jpg", "Xilinx.ppm").c_str()); // main.cpp:133
  hittables.emplace_back(xy_rect(2, 4, 0, 1, -1, lambertian_material(t)));
  hittables.emplace_back(sphere(point{4, 1, 2.25f}, 1, lambertian_material(t)));
  hittables.emplace_back(sphere(point{0, 1, 0}, 1, dielectric_material(1.5f, color{1.0f, 0.5f, 0.5f})));
  hittables.emplace_back(sphere(point{-4, 1, 0}, 1, lambertian_material(color(0.4f, 0.2f, 0.1f))));
  hittables.emplace_back(sphere(point{0, 1, -2.25f}, 1, metal_material(color(0.7f, 0.6f, 0.5f), 0.0f)));
  if (procedural) { auto syc = procedural_image(320, 140, 1); t = image_texture::from_rgb8(syc.data(), 320, 140, 5); }
  else t = image_texture::image_texture_factory(image_file("SYCL.png", "SYCL.ppm").c_str(), 5); // main.cpp:145
  hittables.emplace_back(sphere{point{-60, 3, 5}, 4, lambertian_material{t}});
  hittables.emplace_back(box{point{6.5f, 0, -1.5f}, point{7.0f, 3.0f, -1.0f}, metal_material{color{0.7f, 0.6f, 0.5f}, 0.25f}});
  sphere smoke_sphere = sphere{point{5, 1, 3.5f}, 1, lambertian_material{color{0.75f, 0.75f, 0.75f}}};
  hittables.emplace_back(constant_medium{smoke_sphere, 1, color{1, 1, 1}});

  point look_from{13, 3, 3}, look_at{0, -1, 0};
  vec vup{0, 1, 0};
  float fx = look_at.x() - look_from.x(), fy = look_at.y() - look_from.y(), fz = look_at.z() - look_from.z();
  real_t focus_dist = std::sqrt(fx * fx + fy * fy + fz * fz); // main.cpp:179
  camera cam{look_from, look_at, vup, 40, static_cast<real_t>(width) / height, 0.04f, focus_dist, 0.0f, 1.0f};

  if (argc > 5 && std::string(argv[5]) != "-") { // testing aid: dump the flattened C-ABI tables instead of rendering (no GPU needed)
    scene_tables tb = flatten(hittables);
    std::ofstream d(argv[5], std::ios::binary);
    int32_t n[3] = {(int32_t)tb.hittables.size(), (int32_t)tb.materials.size(), (int32_t)tb.textures.size()};
    d.write((const char*)n, sizeof n);
    d.write((const char*)tb.hittables.data(), tb.hittables.size() * sizeof(PtHittable));
    d.write((const char*)tb.materials.data(), tb.materials.size() * sizeof(PtMaterial));
    d.write((const char*)tb.textures.data(), tb.textures.size() * sizeof(PtTexture));
    d.write((const char*)&cam.c, sizeof cam.c);
    return 0;
  }

  frame_buffer fb;
  auto t0 = std::chrono::steady_clock::now();
  try {
    render(width, height, samples, fb, hittables, cam);
  } catch (const pt_error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // the output stage of main.cpp:33-59 (sqrt gamma, clamp to [0, 0.999], x 256, rows flipped), then save_image_png (a file name ending in
  // .png: pt/image_io.hpp's writer in place of stbi_write_png) or dump_image_ppm (main.cpp:17-31; binary P6 instead of text P3)
  std::vector<uint8_t> pixels;
  pixels.reserve((size_t)width * height * 3);
  for (int y = height - 1; y >= 0; y--)
    for (int x = 0; x < width; x++) {
      const color& c = fb[(size_t)y * width + x];
      for (int k = 0; k < 3; k++) {
        float s = std::sqrt(c.v[k]);
        float cl = std::clamp(s, 0.0f, 0.999f);
        float v = 256 * cl;
        pixels.push_back((uint8_t)(v == v ? (int)v : 0));
      }
    }
  const std::string out_name = out;
  if (out_name.size() >= 4 && out_name.compare(out_name.size() - 4, 4, ".png") == 0) {
    if (!image_io::write_png(out, pixels.data(), (size_t)width, (size_t)height)) { std::fprintf(stderr, "cannot write %s\n", out); return 1; }
  } else {
    std::ofstream f(out, std::ios::binary);
    f << "P6\n" << width << " " << height << "\n255\n";
    f.write((const char*)pixels.data(), (std::streamsize)pixels.size());
  }
  std::printf("%zu hittables, %dx%dx%d spp in %.3f s (scene upload + render + copy back) -> %s\n", hittables.size(), width,
              height, samples, sec, out);
  return 0;
}
