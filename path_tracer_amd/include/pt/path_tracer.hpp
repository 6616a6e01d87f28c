// path_tracer.hpp — C++20 host facade over the C ABI (include/pt_render.h).
//
// Keeps the reference's scene-description types as the input API surface (same names,
// constructor argument order and meaning; citations into /root/reference/include):
//
//   sphere(cen, r, mat) / sphere(cen0, cen1, t0, t1, r, mat)     sphere.hpp:30,40
//   xy_rect / xz_rect / yz_rect (a0, a1, b0, b1, k, mat)          rectangle.hpp:21,59,97
//   triangle(v0, v1, v2, mat)                                      triangle.hpp:107
//   box(p0, p1, mat)                                               box.hpp:15
//   constant_medium(boundary, density, color | texture)            constant_medium.hpp:18,23
//   lambertian_material / metal_material / dielectric_material /
//   lightsource_material / isotropic_material                      material.hpp:11-131
//   solid_texture / checker_texture / image_texture                texture.hpp:18-152
//   camera(look_from, look_at, vup, vfov, aspect, aperture, focus_dist, t0, t1)   camera.hpp:67-69
//   hittable_t = std::variant<sphere, xy_rect, triangle, box, constant_medium[, xz_rect, yz_rect]>
//                                                                  render.hpp:22-23
//   render<width, height, samples>(frame_buf, hittables, cam)      render.hpp:141-143
//
// The std::variant values never reach the device: flatten() walks them once on the host
// (std::visit) into the tagged tables of PtSceneDesc; pt_scene_create() turns those into the
// 16-byte record runs the kernels read from LDS.  Header-only; link with -lpt_render.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <variant>
#include <vector>

#include "../../../include/pt_render.h"
#include "image_io.hpp"

namespace pt {

using real_t = float; // vec.hpp:8

// Minimal float3 with the accessor names of sycl::float3 (rtweekend.hpp:25-27 aliases it three ways).
struct float3 {
  float v[3]{0, 0, 0};
  constexpr float3() = default;
  constexpr float3(float x, float y, float z) : v{x, y, z} {}
  constexpr float x() const { return v[0]; }
  constexpr float y() const { return v[1]; }
  constexpr float z() const { return v[2]; }
  friend bool operator==(const float3& a, const float3& b) { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2]; }
  friend bool operator<(const float3& a, const float3& b) { return std::tie(a.v[0], a.v[1], a.v[2]) < std::tie(b.v[0], b.v[1], b.v[2]); }
};
using point = float3;
using color = float3;
using vec = float3;

class pt_error : public std::runtime_error {
 public:
  pt_error(int code, const std::string& where)
      : std::runtime_error(where + ": " + pt_error_string(code) + " (" + pt_last_error() + ")"), code(code) {}
  int code;
};
inline void check(int rc, const char* where) {
  if (rc != PT_OK) throw pt_error(rc, where);
}

// ---- textures (texture.hpp) ------------------------------------------------------------------
struct solid_texture {
  solid_texture() = default;
  solid_texture(const color& c) : color_value{c} {}
  solid_texture(float r, float g, float b) : color_value{r, g, b} {}
  color color_value;
  auto key() const { return std::tie(color_value); }
};
struct checker_texture {
  checker_texture() = default;
  checker_texture(const solid_texture& x, const solid_texture& y) : odd{x}, even{y} {} // texture.hpp:35
  checker_texture(const color& c1, const color& c2) : odd{c1}, even{c2} {}              // texture.hpp:38
  solid_texture odd, even;
  auto key() const { return std::tie(odd.color_value, even.color_value); }
};
// The atlas behind image_texture (texture.hpp:71,157): starts with the {0,0,1} fallback texel.
struct texture_atlas {
  std::vector<uint8_t> data{0, 0, 1};
};
inline texture_atlas& default_atlas() {
  static texture_atlas a;
  return a;
}
struct image_texture {
  // texture.hpp:97-117 with the decode step left to the caller (stb is not a dependency here):
  // `rgb` is height*width*3 bytes, rows top-down.
  static image_texture from_rgb8(const uint8_t* rgb, std::size_t width, std::size_t height, float cyclic_frequency = 1.f,
                                 texture_atlas& atlas = default_atlas()) {
    image_texture t;
    if (!rgb || !width || !height) return t; // load failure: 1x1 texture at the fallback texel (texture.hpp:106-111)
    t.width = width; t.height = height; t.cyclic_frequency = cyclic_frequency;
    t.offset = atlas.data.size() / 3;
    atlas.data.insert(atlas.data.end(), rgb, rgb + width * height * 3);
    return t;
  }
  // texture.hpp:97-117: load an image file and append its texels to the atlas.  stb_image is not a dependency of this host: PNG,
  // baseline JPEG and binary PPM are decoded by pt/image_io.hpp (the reference's own images/Xilinx.jpg and images/SYCL.png load as
  // main.cpp:133,145 load them; the decoded texels are the ones the Python host's loader produces — image_io.hpp says which
  // decoder choices that pins).  Everything else keeps the reference's failure semantics (texture.hpp:106-111): a message on
  // stderr and the 1x1 texture at offset 0, i.e. the {0,0,1} fallback texel the atlas starts with; never an exception.
  static image_texture image_texture_factory(const char* file_name, float cyclic_frequency = 1.f,
                                             texture_atlas& atlas = default_atlas()) {
    image_io::Image img;
    const char* reason = image_io::load_rgb8(file_name, img);
    if (reason) {
      std::cerr << "ERROR: Could not load texture image file '" << (file_name ? file_name : "(null)") << "'.\n" << reason << std::endl;
      image_texture t;
      t.cyclic_frequency = cyclic_frequency; // texture.hpp:116: w = h = 1, offset 0, the caller's frequency
      return t;
    }
    return from_rgb8(img.rgb.data(), img.width, img.height, cyclic_frequency, atlas);
  }
  std::size_t width{1}, height{1}, offset{0};
  float cyclic_frequency{1.f};
  auto key() const { return std::tie(width, height, offset, cyclic_frequency); }

};
using texture_t = std::variant<checker_texture, solid_texture, image_texture>; // texture.hpp:154

// ---- materials (material.hpp) ------------------------------------------------------------------
struct lambertian_material {
  lambertian_material() = default;
  lambertian_material(const color& a) : albedo{solid_texture{a}} {}
  lambertian_material(const texture_t& a) : albedo{a} {}
  texture_t albedo;
};
struct metal_material {
  metal_material() = default;
  metal_material(const color& a, float f) : albedo{a}, fuzz{std::clamp(f, 0.0f, 1.0f)} {} // material.hpp:35-37
  color albedo;
  float fuzz{0};
};
struct dielectric_material {
  dielectric_material() = default;
  dielectric_material(real_t ri, const color& albedo) : ref_idx{ri}, albedo{albedo} {}
  real_t ref_idx{1};
  color albedo;
};
struct lightsource_material {
  lightsource_material() = default;
  lightsource_material(const texture_t& a) : emit{a} {}
  lightsource_material(const color& a) : emit{solid_texture{a}} {}
  texture_t emit;
};
struct isotropic_material {
  isotropic_material(const color& a) : albedo{solid_texture{a}} {}
  isotropic_material(const texture_t& a) : albedo{a} {}
  texture_t albedo;
};
using material_t = std::variant<lambertian_material, metal_material, dielectric_material, lightsource_material,
                                isotropic_material>; // material.hpp:133-135

// ---- hittables ------------------------------------------------------------------------------------
struct sphere {
  sphere() = default;
  sphere(const point& cen, real_t r, const material_t& m) : center0{cen}, center1{cen}, radius{r}, time0{0}, time1{0}, material_type{m} {}
  sphere(const point& cen0, const point& cen1, real_t t0, real_t t1, real_t r, const material_t& m)
      : center0{cen0}, center1{cen1}, radius{r}, time0{t0}, time1{t1}, material_type{m} {}
  point center0, center1;
  real_t radius{0}, time0{0}, time1{0};
  material_t material_type;
};
struct xy_rect {
  xy_rect() = default;
  xy_rect(real_t x0, real_t x1, real_t y0, real_t y1, real_t k, const material_t& m) : x0{x0}, x1{x1}, y0{y0}, y1{y1}, k{k}, material_type{m} {}
  real_t x0{}, x1{}, y0{}, y1{}, k{};
  material_t material_type;
};
struct xz_rect {
  xz_rect() = default;
  xz_rect(real_t x0, real_t x1, real_t z0, real_t z1, real_t k, const material_t& m) : x0{x0}, x1{x1}, z0{z0}, z1{z1}, k{k}, material_type{m} {}
  real_t x0{}, x1{}, z0{}, z1{}, k{};
  material_t material_type;
};
struct yz_rect {
  yz_rect() = default;
  yz_rect(real_t y0, real_t y1, real_t z0, real_t z1, real_t k, const material_t& m) : y0{y0}, y1{y1}, z0{z0}, z1{z1}, k{k}, material_type{m} {}
  real_t y0{}, y1{}, z0{}, z1{}, k{};
  material_t material_type;
};
// triangle.hpp:102-122: _triangle<IntersectionStrategy>; `triangle` = the Moller-Trumbore default (what main.cpp builds),
// `badouel_triangle` = _triangle<badouel_ray_triangle_intersec> (triangle.hpp:14-56).
template <int Strategy = PT_TRI_MOLLER_TRUMBORE>
struct _triangle {
  _triangle() = default;
  _triangle(const point& v0, const point& v1, const point& v2, const material_t& m) : v0{v0}, v1{v1}, v2{v2}, material_type{m} {}
  point v0, v1, v2;
  material_t material_type;
  static constexpr int strategy = Strategy;
};
using triangle = _triangle<>;
using badouel_triangle = _triangle<PT_TRI_BADOUEL>;
struct box {
  box() = default;
  box(const point& p0, const point& p1, const material_t& m) : box_min{p0}, box_max{p1}, material_type{m} {}
  point box_min, box_max;
  material_t material_type;
};
using hittableVolume_t = std::variant<sphere, box>; // constant_medium.hpp:10
struct constant_medium {
  constant_medium(const hittableVolume_t& b, real_t d, const texture_t& a) : boundary{b}, neg_inv_density{-1 / d}, phase_function{isotropic_material{a}} {}
  constant_medium(const hittableVolume_t& b, real_t d, const color& a) : boundary{b}, neg_inv_density{-1 / d}, phase_function{isotropic_material{a}} {}
  hittableVolume_t boundary;
  real_t neg_inv_density;
  material_t phase_function;
};
// The reference's five alternatives first (same indices); xz_rect/yz_rect and the Badouel-strategy triangle are extensions.
using hittable_t = std::variant<sphere, xy_rect, triangle, box, constant_medium, xz_rect, yz_rect, badouel_triangle>;

// ---- camera (camera.hpp:67-87) -----------------------------------------------------------------------
class camera {
 public:
  camera(const point& look_from, const point& look_at, const vec& vup, real_t degree_vfov, real_t aspect_ratio,
         real_t aperture, real_t focus_dist, real_t time0 = 0, real_t time1 = 0) {
    check(pt_camera_init(&c, look_from.v, look_at.v, vup.v, degree_vfov, aspect_ratio, aperture, focus_dist, time0, time1),
          "pt_camera_init");
  }
  PtCamera c{};
};

// ---- flattening the variants into the C-ABI tables ------------------------------------------------------
struct scene_tables {
  std::vector<PtHittable> hittables;
  std::vector<PtMaterial> materials;
  std::vector<PtTexture> textures;
  std::vector<uint8_t> atlas;
  PtSceneDesc desc() const {
    PtSceneDesc d{};
    d.hittables = hittables.data(); d.n_hittables = (int32_t)hittables.size();
    d.materials = materials.data(); d.n_materials = (int32_t)materials.size();
    d.textures = textures.data(); d.n_textures = (int32_t)textures.size();
    d.atlas = atlas.empty() ? nullptr : atlas.data(); d.atlas_bytes = atlas.size();
    return d;
  }
};

namespace detail {
inline void put3(float* d, const float3& s) { d[0] = s.v[0]; d[1] = s.v[1]; d[2] = s.v[2]; }

struct flattener {
  scene_tables& out;
  bool uses_image = false;
  // value-equal textures / materials share one table entry (same rule as the Python packer)
  std::map<std::tuple<int, std::array<float, 7>, std::array<uint64_t, 3>>, int> tex_ids;
  std::map<std::tuple<int, int, std::array<float, 4>>, int> mat_ids;

  int texture(const texture_t& t) {
    PtTexture e{};
    std::array<float, 7> f{};
    std::array<uint64_t, 3> u{};
    e.kind = (int32_t)t.index(); // variant order == ABI tag order (texture.hpp:154)
    if (auto* s = std::get_if<solid_texture>(&t)) {
      put3(e.color0, s->color_value);
    } else if (auto* c = std::get_if<checker_texture>(&t)) {
      put3(e.color0, c->odd.color_value); put3(e.color1, c->even.color_value);
    } else {
      auto& i = std::get<image_texture>(t);
      e.width = (uint32_t)i.width; e.height = (uint32_t)i.height; e.offset = (uint32_t)i.offset; e.freq = i.cyclic_frequency;
      u = {i.width, i.height, i.offset};
      uses_image = true;
    }
    std::copy(e.color0, e.color0 + 3, f.begin()); std::copy(e.color1, e.color1 + 3, f.begin() + 3); f[6] = e.freq;
    auto key = std::make_tuple(e.kind, f, u);
    auto it = tex_ids.find(key);
    if (it != tex_ids.end()) return it->second;
    out.textures.push_back(e);
    return tex_ids[key] = (int)out.textures.size() - 1;
  }

  int material(const material_t& m) {
    PtMaterial e{};
    e.kind = (int32_t)m.index(); // variant order == ABI tag order (material.hpp:133-135)
    e.texture = -1;
    std::visit([&](auto&& a) {
      using T = std::decay_t<decltype(a)>;
      if constexpr (std::is_same_v<T, lambertian_material> || std::is_same_v<T, isotropic_material>) e.texture = texture(a.albedo);
      else if constexpr (std::is_same_v<T, lightsource_material>) e.texture = texture(a.emit);
      else if constexpr (std::is_same_v<T, metal_material>) { put3(e.color, a.albedo); e.param = a.fuzz; }
      else { put3(e.color, a.albedo); e.param = a.ref_idx; }
    }, m);
    auto key = std::make_tuple(e.kind, e.texture, std::array<float, 4>{e.color[0], e.color[1], e.color[2], e.param});
    auto it = mat_ids.find(key);
    if (it != mat_ids.end()) return it->second;
    out.materials.push_back(e);
    return mat_ids[key] = (int)out.materials.size() - 1;
  }

  static void fill(float* f, const sphere& s) { put3(f, s.center0); put3(f + 3, s.center1); f[6] = s.radius; f[7] = s.time0; f[8] = s.time1; }
  static void fill(float* f, const box& b) { put3(f, b.box_min); put3(f + 3, b.box_max); }

  void hittable(const hittable_t& h) {
    PtHittable e{};
    std::visit([&](auto&& a) {
      using T = std::decay_t<decltype(a)>;
      if constexpr (std::is_same_v<T, sphere>) { e.kind = PT_HIT_SPHERE; e.material = material(a.material_type); fill(e.f, a); }
      else if constexpr (std::is_same_v<T, xy_rect>) { e.kind = PT_HIT_XY_RECT; e.material = material(a.material_type); float v[5] = {a.x0, a.x1, a.y0, a.y1, a.k}; std::copy(v, v + 5, e.f); }
      else if constexpr (std::is_same_v<T, xz_rect>) { e.kind = PT_HIT_XZ_RECT; e.material = material(a.material_type); float v[5] = {a.x0, a.x1, a.z0, a.z1, a.k}; std::copy(v, v + 5, e.f); }
      else if constexpr (std::is_same_v<T, yz_rect>) { e.kind = PT_HIT_YZ_RECT; e.material = material(a.material_type); float v[5] = {a.y0, a.y1, a.z0, a.z1, a.k}; std::copy(v, v + 5, e.f); }
      else if constexpr (std::is_same_v<T, triangle> || std::is_same_v<T, badouel_triangle>) {
        e.kind = PT_HIT_TRIANGLE; e.strategy = T::strategy; e.material = material(a.material_type);
        put3(e.f, a.v0); put3(e.f + 3, a.v1); put3(e.f + 6, a.v2);
      }
      else if constexpr (std::is_same_v<T, box>) { e.kind = PT_HIT_BOX; e.material = material(a.material_type); fill(e.f, a); }
      else {
        e.kind = PT_HIT_CONSTANT_MEDIUM; e.material = material(a.phase_function);
        if (auto* s = std::get_if<sphere>(&a.boundary)) { e.boundary_kind = PT_HIT_SPHERE; fill(e.f, *s); }
        else { e.boundary_kind = PT_HIT_BOX; fill(e.f, std::get<box>(a.boundary)); }
        e.f[9] = a.neg_inv_density;
      }
    }, h);
    out.hittables.push_back(e);
  }
};
} // namespace detail

inline scene_tables flatten(const std::vector<hittable_t>& hittables, const texture_atlas& atlas = default_atlas()) {
  scene_tables t;
  detail::flattener f{t};
  for (auto& h : hittables) f.hittable(h); // list order == traversal order (render.hpp:37)
  if (f.uses_image) t.atlas = atlas.data;
  return t;
}

// Device-resident scene; RAII over pt_scene_create / pt_scene_destroy.
class device_scene {
 public:
  explicit device_scene(const std::vector<hittable_t>& hittables, const texture_atlas& atlas = default_atlas()) {
    scene_tables t = flatten(hittables, atlas);
    PtSceneDesc d = t.desc();
    check(pt_scene_create(&d, &s), "pt_scene_create");
  }
  ~device_scene() { pt_scene_destroy(s); }
  device_scene(const device_scene&) = delete;
  device_scene& operator=(const device_scene&) = delete;
  PtScene* s = nullptr;
};

// frame buffer [height][width] of color, y = 0 is the bottom scan-line (render.hpp:105, main.cpp:41)
using frame_buffer = std::vector<color>;

// render.hpp:141-160 with run-time sizes; depth 50 as render.hpp:144.
inline void render(int width, int height, int samples, frame_buffer& frame_buf, const std::vector<hittable_t>& hittables,
                   const camera& cam, int depth = 50) {
  device_scene scene(hittables);
  PtRenderParams p{width, height, samples, depth, 0, 1, 0, 0};
  frame_buf.resize((std::size_t)width * height);
  static_assert(sizeof(color) == 12);
  check(pt_render_host(scene.s, &cam.c, &p, reinterpret_cast<float*>(frame_buf.data())), "pt_render_host");
}

// The reference's call shape: render<width, height, samples>(frame_buf, hittables, cam).
template <int width, int height, int samples>
void render(frame_buffer& frame_buf, std::vector<hittable_t>& hittables, camera& cam) {
  render(width, height, samples, frame_buf, hittables, cam);
}

} // namespace pt
