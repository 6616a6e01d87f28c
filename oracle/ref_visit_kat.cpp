// ref_visit_kat.cpp — known-answer program built from the REFERENCE's own visit.hpp (standard headers only: <cassert>,
// <type_traits>, <variant>), compiled from where it lies under /root/reference/include with plain g++ — no stand-in headers,
// nothing copied.  It pins what the ABI's integer tags (include/pt_render.h) assume about dev_visit (visit.hpp:51-67): a variant
// is dispatched by its index(), the callable receives std::get<index()>, so "tag k" == "k-th alternative of the variant as
// declared".  The alternatives here are this build's own tag types, declared in the reference's variant orders (render.hpp:22-23
// hittables, material.hpp:133-135 materials, texture.hpp:154 textures, rectangle.hpp:130 rects, constant_medium.hpp:10 volumes —
// those orders themselves are watched by tests/test_reference_drift_cpu.py, which parses the reference's declarations).
//
// Output: one line per alternative: "<family> <index()> <tag the visited type carries> <value seen through the reference>".
// Build: oracle/Makefile (target _ref/visit_kat).  TEST INFRASTRUCTURE ONLY.
#include <cstdio>
#include <variant>

#include "../include/pt_render.h"
#include "visit.hpp" // from -I/root/reference/include

template <int Tag> struct alt { static constexpr int tag = Tag; int payload; };

using hittable_v = std::variant<alt<PT_HIT_SPHERE>, alt<PT_HIT_XY_RECT>, alt<PT_HIT_TRIANGLE>, alt<PT_HIT_BOX>, alt<PT_HIT_CONSTANT_MEDIUM>>;
using material_v = std::variant<alt<PT_MAT_LAMBERTIAN>, alt<PT_MAT_METAL>, alt<PT_MAT_DIELECTRIC>, alt<PT_MAT_LIGHTSOURCE>, alt<PT_MAT_ISOTROPIC>>;
using texture_v = std::variant<alt<PT_TEX_CHECKER>, alt<PT_TEX_SOLID>, alt<PT_TEX_IMAGE>>;
using rect_v = std::variant<alt<0>, alt<1>, alt<2>>;   // xy, xz, yz: the device's rect axis (pt_flatten.hpp)
using volume_v = std::variant<alt<PT_HIT_SPHERE>, alt<PT_HIT_BOX + 100>>; // sphere 0, box 1 (constant_medium.hpp:10): printed as index only

template <typename V, std::size_t... I>
static void family(const char* name, std::index_sequence<I...>) {
  (([&] {
     V v{std::in_place_index<I>, std::variant_alternative_t<I, V>{1000 + (int)I}};
     // the reference's dispatcher: returns what the callable returns for the alternative it selected
     const int tag = dev_visit([](auto&& a) { return std::remove_reference_t<decltype(a)>::tag; }, v);
     const int payload = dev_visit([](auto&& a) { return a.payload; }, v);
     std::printf("%s %zu %d %d\n", name, v.index(), tag, payload);
   }()),
   ...);
}

int main() {
  family<hittable_v>("hittable", std::make_index_sequence<std::variant_size_v<hittable_v>>{});
  family<material_v>("material", std::make_index_sequence<std::variant_size_v<material_v>>{});
  family<texture_v>("texture", std::make_index_sequence<std::variant_size_v<texture_v>>{});
  family<rect_v>("rectangle", std::make_index_sequence<std::variant_size_v<rect_v>>{});
  family<volume_v>("volume", std::make_index_sequence<std::variant_size_v<volume_v>>{});
  return 0;
}
