// pt_flatten.hpp — host side of pt_scene_create(): validate the ABI tables and flatten
// them into the device blob the kernels read (layout documented in pt_device.hpp and
// DESIGN.md "Data layout in HBM").  Pure host C++ (no HIP), so it is unit-testable on a
// machine without a GPU through pt_debug_flatten().
//
// Replaces: the AoS sycl::buffer<hittable_t> of 624-byte variants (render.hpp:146-147).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_render.h"
#include "pt_tripool.hpp"

namespace ptf {

struct F4 {
  float x, y, z, w;
};
static_assert(sizeof(F4) == 16);

inline float as_f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

enum { DK_SPHERE = 0, DK_RECT = 1, DK_TRI = 2, DK_BOX = 3, DK_MEDIUM = 4, DK_TRI_B = 5 /* Badouel-strategy triangles */,
       DK_ABSORBED = 8 /* flag in a run header's kind: a sphere run that an earlier run's lists test (flatten: "absorbed sphere runs") */ };
// hit id (pt_device.hpp: hit_pack): [24:0] record offset in the blob (F4 units), [27:25] box side, [30:28] device kind
enum { kHitOffBits = 25, kHitSideShift = 25, kHitKindShift = 28 };

// The pool buffer is described, not materialised, on the host: a list of segments (offset in F4 units, the dwords that go there), the big
// ones MOVED out of the TriPool — a copy of the 100 k-triangle mesh's maps (0.6 GB at round 6's resolutions, 2.9 GB at round 5's) into one contiguous host vector cost seconds of
// first-touch page faults.  pt_scene_create uploads segment by segment into a zeroed device buffer; pt_debug_flatten_pool assembles them.
struct PoolSegment { uint64_t at_f4; std::vector<uint32_t> dwords; };
struct PoolLayout {
  std::vector<PoolSegment> segments;
  uint64_t size_f4 = 0;
  uint32_t put(std::vector<uint32_t>&& v, int spare_f4) { // returns the segment's offset; `spare_f4` zeroed records follow it (the scans may load a whole chunk without clamping)
    const uint64_t at = size_f4; // (< 2^31: flatten() checks a pool's whole size before it puts any of its tables)
    size_f4 += (v.size() + 3) / 4 + (uint64_t)spare_f4;
    segments.push_back(PoolSegment{at, std::move(v)});
    return (uint32_t)at;
  }
  void assemble(F4* out) const { // out: size_f4 records
    std::memset((void*)out, 0, (size_t)size_f4 * 16);
    for (const PoolSegment& sg : segments) if (!sg.dwords.empty()) std::memcpy((void*)(out + sg.at_f4), sg.dwords.data(), sg.dwords.size() * 4);
  }
};
struct Flat {
  std::vector<F4> blob; // [n_runs run headers][records; a sphere run is preceded by its offset lists + aux F4]
  std::vector<F4> mats; // 4 F4 per material, texture inlined
  PoolLayout pool;      // tables of the triangle pools (pt_tripool.hpp): a buffer of their own, addressed by 32-bit F4 offsets from the pools' headers
  int32_t n_runs = 0;
  bool has_image = false;
  bool has_medium = false;
  bool coop_ok = true; // no triangle/medium carries an image texture (stale u,v cannot matter): pt_device.hpp coop
  int32_t coop_prefix = 0; // hittables before the first constant_medium
  bool fast_ok = true; // all rect/box coordinates finite with |v| <= 2^60 (pt_device.hpp: RayCtx)
  bool has_badouel = false; // some triangle uses the Badouel strategy (its own device kind and kernel instantiations)
  int grid_spheres = 0;     // spheres that sit in a culling grid (pt_scene_create: their scan is cheap)
  int pooled = 0;           // rects and boxes that sit in a slab pool (pt_device.hpp: slab_pool)
  int tri_pooled = 0;       // triangles that sit in a triangle pool (pt_tripool.hpp; pt_device.hpp: tri_pool_scan)
  double tri_cells_per_triangle = 0; // statistics of the (last) triangle pool, for the tests
  int tri_wide = 0, tri_maps = 0;   // triangles whose band covers every direction; direction maps built
  long long tri_map_entries[3] = {0, 0, 0};
  int tri_map_res[3] = {0, 0, 0};
  int absorbed_spheres = 0;         // spheres of short later runs that an earlier run's lists test (flatten: "absorbed sphere runs")
  int tri_pool_runs = 0;            // runs that got a pool; the binned renderer (pt_render.hip: launch_binned) serves scenes with exactly one
  int tri_pool_run = -1, tri_pool_hdr = 0, tri_pool_goff = 0, tri_pool_count = 0; // that run: its index, its pool header and first record in the blob, its triangles
};

inline int device_kind(int32_t k) {
  switch (k) {
    case PT_HIT_SPHERE: return DK_SPHERE;
    case PT_HIT_XY_RECT: case PT_HIT_XZ_RECT: case PT_HIT_YZ_RECT: return DK_RECT;
    case PT_HIT_TRIANGLE: return DK_TRI;
    case PT_HIT_BOX: return DK_BOX;
    case PT_HIT_CONSTANT_MEDIUM: return DK_MEDIUM;
    default: return -1;
  }
}

inline int device_kind(const PtHittable& h) { // a Badouel-strategy triangle is a device kind of its own (own run, own loop)
  if (h.kind == PT_HIT_TRIANGLE && h.strategy == PT_TRI_BADOUEL) return DK_TRI_B;
  return device_kind(h.kind);
}

inline int record_size(int dk) {
  switch (dk) { case DK_SPHERE: return 3; case DK_RECT: return 2; case DK_TRI: case DK_TRI_B: return 3; case DK_BOX: return 2; default: return 4; }
}

inline int validate(const PtSceneDesc* sc, std::string& err) {
  if (!sc) { err = "scene description is NULL"; return PT_ERR_INVALID_ARG; }
  if (sc->n_hittables < 0 || sc->n_materials < 0 || sc->n_textures < 0) { err = "negative table size"; return PT_ERR_INVALID_ARG; }
  if ((sc->n_hittables && !sc->hittables) || (sc->n_materials && !sc->materials) || (sc->n_textures && !sc->textures)) {
    err = "table pointer is NULL"; return PT_ERR_INVALID_ARG;
  }
  for (int i = 0; i < sc->n_textures; i++) {
    const PtTexture& t = sc->textures[i];
    if (t.kind < 0 || t.kind > PT_TEX_IMAGE) { err = "texture " + std::to_string(i) + ": bad kind"; return PT_ERR_BAD_SCENE; }
    if (t.kind == PT_TEX_IMAGE) {
      if (t.width < 1 || t.height < 1) { err = "texture " + std::to_string(i) + ": empty image"; return PT_ERR_BAD_SCENE; }
      if (!sc->atlas || ((uint64_t)t.offset + (uint64_t)t.width * t.height) * 3 > sc->atlas_bytes) {
        err = "texture " + std::to_string(i) + ": image outside the atlas"; return PT_ERR_BAD_SCENE;
      }
    }
  }
  for (int i = 0; i < sc->n_materials; i++) {
    const PtMaterial& m = sc->materials[i];
    if (m.kind < 0 || m.kind > PT_MAT_ISOTROPIC) { err = "material " + std::to_string(i) + ": bad kind"; return PT_ERR_BAD_SCENE; }
    bool needs_tex = m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_LIGHTSOURCE || m.kind == PT_MAT_ISOTROPIC;
    if (needs_tex && (m.texture < 0 || m.texture >= sc->n_textures)) {
      err = "material " + std::to_string(i) + ": texture index out of range"; return PT_ERR_BAD_SCENE;
    }
  }
  for (int i = 0; i < sc->n_hittables; i++) {
    const PtHittable& h = sc->hittables[i];
    if (device_kind(h.kind) < 0) { err = "hittable " + std::to_string(i) + ": bad kind"; return PT_ERR_BAD_SCENE; }
    if (h.material < 0 || h.material >= sc->n_materials) { err = "hittable " + std::to_string(i) + ": material index out of range"; return PT_ERR_BAD_SCENE; }
    if (h.kind == PT_HIT_CONSTANT_MEDIUM && h.boundary_kind != PT_HIT_SPHERE && h.boundary_kind != PT_HIT_BOX) {
      err = "hittable " + std::to_string(i) + ": constant_medium boundary must be a sphere or a box"; return PT_ERR_BAD_SCENE;
    }
    if (h.kind == PT_HIT_TRIANGLE && h.strategy != PT_TRI_MOLLER_TRUMBORE && h.strategy != PT_TRI_BADOUEL) {
      err = "hittable " + std::to_string(i) + ": unknown triangle strategy"; return PT_ERR_BAD_SCENE;
    }
  }
  return PT_OK;
}

inline void put_sphere(std::vector<F4>& b, const float* f, int32_t mat, int32_t hidx) {
  const float r2 = f[6] * f[6];                          // radius^2 (sphere.hpp:71)
  b.push_back({f[0], f[1], f[2], f[7] != f[8] ? -r2 : r2}); // c0; sign bit of r^2 = "moving" (sphere.hpp:52)
  b.push_back({f[6], as_f(mat), f[7], f[8]});            // radius, material, time0, time1
  b.push_back({f[3] - f[0], f[4] - f[1], f[5] - f[2], as_f(hidx)}); // center1 - center0 (sphere.hpp:55), ray-independent
}
// In front of a sphere run's records (pt_device.hpp: sphere_scan / sphere_grid_scan), in address order:
//   [grid cell table][grid candidates][big static list][big moving list]     only when the run has a grid (flags bit 2)
//   [static list][moving list]                                                all spheres of the run
//   [ghdr0..ghdr3]                                                            only with a grid
//   [aux F4 = (time0, time1, number of static spheres, flags)]
// lists: record offsets (F4 units, relative to the run's first record) of the static / the moving spheres, each in list
// order, four per F4, padded to a multiple of four by repeating the last entry.  flags bit 0: every moving sphere of the run
// has the shutter interval (time0, time1); bit 1: something moves; bit 2: grid.
//
// The grid is EXACT culling for the run's small spheres (the "big" lists hold the others): a uniform grid over their
// bounds; a sphere is listed in every cell its box [centre range -+ (|r| + m)] touches.  Why that is enough: the reference's
// discriminant b*b - a*cc (sphere.hpp:69-72), evaluated in binary32, differs from the exact a*(r^2 - p^2) (p = distance of
// the centre from the ray's line) by less than 2^-18 * a * |o - c|^2 (21 roundings' worth, doubled), so a sphere can only
// pass `discriminant > 0` if p^2 < r^2 + 2^-18 |o - c|^2, and then the accepted root's point lies within that same radius
// of the centre.  A ray whose origin is within rlimit of the grid's centre has 2^-18 |o - c|^2 <= 2 r_min m + m^2 for
// every small sphere, i.e. the inflated radius is <= |r| + m: the point of any acceptable hit lies inside the sphere's
// box, hence in a cell that lists the sphere and that the ray's forward segment crosses.  Rays from farther away, rays
// outside the run's shutter interval and irregular rays take the full lists (wave-level fallback).
struct SphereGrid {
  bool ok = false;
  float origin[3], inv_cell, center[3], rlimit2;
  int n[3];
  std::vector<uint32_t> cells; // per cell: (first candidate << 8) | count
  std::vector<uint16_t> cand;  // index of the sphere in the run | moving << 15 (two per dword: LDS space sets the occupancy)
  std::vector<int32_t> big_st, big_mv;
};

// margin m in median radii; cell edge in (median radius + margin).  Swept on the 496-hittable scene at 1080p x 1024 spp
// (tools/grid_sweep.sh, tools/grid_rep.sh): (1.5, 2.8) — round 2's first choice — 3 730 Msamples/s, (0.5, 3.0) 4 240,
// (0.5, 3.33) 4 190, (0.75, 3.43) 4 070, (0.25, *) <= 3 360.  A smaller margin means fewer cells per sphere (a ray tests
// fewer candidates) but a smaller rlimit (more waves fall back to the full lists for rays that start far out).
struct GridTuning { float m = 0.5f, cell = 3.0f; };
inline SphereGrid build_sphere_grid(const PtHittable* h, int count, bool uniform, GridTuning tune = GridTuning()) {
  SphereGrid g;
  if (count < 48 || count > 32767 || !uniform) return g;
  std::vector<float> rad;
  for (int i = 0; i < count; i++) {
    for (int k = 0; k < 9; k++) if (!std::isfinite(h[i].f[k])) return g;
    rad.push_back(std::fabs(h[i].f[6]));
  }
  std::vector<float> sorted = rad;
  std::nth_element(sorted.begin(), sorted.begin() + count / 2, sorted.end());
  const float r_med = sorted[(size_t)count / 2];
  if (!(r_med > 0.0f)) return g;
  const float m = tune.m * r_med, r_small = 4.0f * r_med;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  float r_min = 3.4e38f;
  int n_small = 0;
  std::vector<char> small((size_t)count, 0);
  auto box_of = [&](int i, double blo[3], double bhi[3]) {
    const float* f = h[i].f;
    const double e = (double)rad[(size_t)i] + m + 1e-3 * (r_med + m); // + slack for the walk's own rounding
    for (int k = 0; k < 3; k++) {
      const double c0 = f[k], c1 = (f[7] != f[8]) ? f[3 + k] : f[k];
      blo[k] = std::min(c0, c1) - e; bhi[k] = std::max(c0, c1) + e;
    }
  };
  for (int i = 0; i < count; i++) {
    if (!(rad[(size_t)i] <= r_small) || !(rad[(size_t)i] > 0.0f)) continue;
    small[(size_t)i] = 1; n_small++;
    r_min = std::min(r_min, rad[(size_t)i]);
    double blo[3], bhi[3];
    box_of(i, blo, bhi);
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[k]); hi[k] = std::max(hi[k], bhi[k]); }
  }
  if (n_small < 48) return g;
  const double cell = (double)tune.cell * ((double)r_med + m);
  long long total = 1;
  for (int k = 0; k < 3; k++) {
    g.n[k] = (int)std::min<double>(64.0, std::max<double>(1.0, std::ceil((hi[k] - lo[k]) / cell)));
    total *= g.n[k];
  }
  // one cell size for all axes (the walk is in cell units); the box is grown to whole cells
  const double kappa = 1.0 / 262144.0; // 2^-18
  double half_diag2 = 0.0;
  for (int k = 0; k < 3; k++) {
    const double ext = g.n[k] * cell, mid = 0.5 * (lo[k] + hi[k]);
    if (ext < hi[k] - lo[k]) return g; // more than 64 cells of this size along an axis: no grid
    g.origin[k] = (float)(mid - 0.5 * ext);
    g.center[k] = (float)mid;
    half_diag2 += 0.25 * ext * ext;
  }
  const double allowed = std::sqrt((2.0 * r_min * m + (double)m * m) / kappa) - std::sqrt(half_diag2) - 1.0;
  if (!(allowed > 0.0) || total > 16384) return g;
  g.rlimit2 = (float)(allowed * allowed * 0.999);
  g.inv_cell = (float)(1.0 / cell);
  const double inv = (double)g.inv_cell; // assign with the float value the device uses
  std::vector<std::vector<uint16_t>> lists((size_t)total);
  for (int i = 0; i < count; i++) {
    if (!small[(size_t)i]) { (h[i].f[7] != h[i].f[8] ? g.big_mv : g.big_st).push_back(i * 3); continue; }
    double blo[3], bhi[3];
    box_of(i, blo, bhi);
    int c0[3], c1[3];
    for (int k = 0; k < 3; k++) {
      c0[k] = std::max(0, std::min(g.n[k] - 1, (int)std::floor((blo[k] - g.origin[k]) * inv)));
      c1[k] = std::max(0, std::min(g.n[k] - 1, (int)std::floor((bhi[k] - g.origin[k]) * inv)));
    }
    const uint16_t entry = (uint16_t)((uint32_t)i | ((h[i].f[7] != h[i].f[8]) ? 0x8000u : 0u));
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) lists[((size_t)z * g.n[1] + y) * g.n[0] + x].push_back(entry);
  }
  for (auto& l : lists) {
    if (l.size() > 255 || g.cand.size() + l.size() >= (1u << 24)) return g; // a cell this crowded: no grid
    g.cells.push_back((uint32_t)(g.cand.size() << 8) | (uint32_t)l.size());
    g.cand.insert(g.cand.end(), l.begin(), l.end());
  }
  g.ok = true;
  return g;
}

inline void put_dwords(std::vector<F4>& b, const uint32_t* d, size_t n) {
  for (size_t k = 0; k < n; k += 4) {
    uint32_t v[4] = {0, 0, 0, 0};
    for (size_t j = 0; j < 4 && k + j < n; j++) v[j] = d[k + j];
    b.push_back({as_f((int32_t)v[0]), as_f((int32_t)v[1]), as_f((int32_t)v[2]), as_f((int32_t)v[3])});
  }
}
inline int put_offset_list(std::vector<F4>& b, std::vector<int32_t> l) { // returns the number of F4 (entries of four)
  if (l.empty()) return 0;
  while (l.size() % 4) l.push_back(l.back());
  for (size_t k = 0; k < l.size(); k += 4) b.push_back({as_f(l[k]), as_f(l[k + 1]), as_f(l[k + 2]), as_f(l[k + 3])});
  return (int)(l.size() / 4);
}

// foreign / patch (round 6, "absorbed runs"): static spheres of LATER runs that this run's lists test as well (flatten() says which and
// why that changes no result); their offsets are known only once the runs between are laid out, so the lists carry the placeholders
// -(1 + k) of `foreign` and flatten() patches the F4 ranges returned in `patch`.
inline int put_sphere_run_aux(std::vector<F4>& b, const PtHittable* h, int count, bool allow_grid, GridTuning tune = GridTuning(),
                              const std::vector<int32_t>* foreign = nullptr, std::vector<std::pair<size_t, size_t>>* patch = nullptr) { // returns the spheres in the grid
  std::vector<int32_t> st, mv;
  bool uniform = true;
  float t0 = 0.0f, t1 = 0.0f;
  for (int i = 0; i < count; i++) {
    const float* f = h[i].f;
    if (f[7] != f[8]) { // moving (sphere.hpp:52)
      if (mv.empty()) { t0 = f[7]; t1 = f[8]; }
      else if (std::memcmp(&t0, &f[7], 4) != 0 || std::memcmp(&t1, &f[8], 4) != 0) uniform = false;
      mv.push_back(i * 3);
    } else st.push_back(i * 3);
  }
  const bool any = !mv.empty();
  SphereGrid g;
  if (allow_grid) g = build_sphere_grid(h, count, uniform && (!any || t0 < t1), tune);
  int n_cell_f4 = 0, n_cand_f4 = 0, qbs = 0, qbm = 0;
  if (g.ok) {
    size_t before = b.size();
    put_dwords(b, g.cells.data(), g.cells.size()); n_cell_f4 = (int)(b.size() - before); before = b.size();
    std::vector<uint32_t> packed((g.cand.size() + 1) / 2, 0u);
    for (size_t k = 0; k < g.cand.size(); k++) packed[k / 2] |= (uint32_t)g.cand[k] << (16 * (k & 1));
    put_dwords(b, packed.data(), packed.size());   n_cand_f4 = (int)(b.size() - before);
    if (foreign && !foreign->empty()) { g.big_st.insert(g.big_st.end(), foreign->begin(), foreign->end()); if (patch) patch->push_back({b.size(), 0}); }
    qbs = put_offset_list(b, g.big_st);
    if (foreign && !foreign->empty() && patch) patch->back().second = (size_t)qbs;
    qbm = put_offset_list(b, g.big_mv);
  }
  if (foreign && !foreign->empty()) { st.insert(st.end(), foreign->begin(), foreign->end()); if (patch) patch->push_back({b.size(), 0}); }
  const int qst = put_offset_list(b, st);
  if (foreign && !foreign->empty() && patch) patch->back().second = (size_t)qst;
  put_offset_list(b, mv);
  if (g.ok) {
    b.push_back({g.origin[0], g.origin[1], g.origin[2], g.inv_cell});
    b.push_back({as_f(g.n[0]), as_f(g.n[1]), as_f(g.n[2]), 1.0f / g.inv_cell});
    b.push_back({g.center[0], g.center[1], g.center[2], g.rlimit2});
    b.push_back({as_f(n_cell_f4), as_f(n_cand_f4), as_f(qbs), as_f(qbm)});
  }
  // aux: (time0, time1, entries of the static list — the run's own static spheres and the absorbed ones —, flags: 1 the moving spheres share
  // their shutter interval, 2 some sphere moves, 4 the run has a grid, 8 this run is absorbed by an earlier one; bits 8 ...: absorbed spheres in its lists)
  b.push_back({t0, t1, as_f((int32_t)st.size()), as_f((uniform ? 1 : 0) | (any ? 2 : 0) | (g.ok ? 4 : 0) | ((foreign ? (int32_t)foreign->size() : 0) << 8))});
  return g.ok ? count + (foreign ? (int)foreign->size() : 0) - (int)g.big_st.size() - (int)g.big_mv.size() : 0;
}
inline void put_box(std::vector<F4>& b, const float* f, int32_t mat, int32_t hidx) {
  b.push_back({f[0], f[1], f[2], as_f(mat)});
  b.push_back({f[3], f[4], f[5], as_f(hidx)});
}

// box_cull: 0 = no slab pools, 1 = where the cost model says they pay, 2 = every stretch of >= 2 rects / boxes (tests)
// In front of a triangle run's records: ONE aux F4 = (1 if the run has a triangle pool else 0, header offset, 0, 0), and for a
// pooled run, before it, the pool's header (pt_device.hpp: tri_pool_scan reads it through the scalar cache).  The pool's TABLES
// live in a second buffer (Flat::pool -> PtScene::pool): round 4 kept them in the blob, whose record offsets are hit-id bits, so a
// mesh of 830 k triangles lost its pool; the tables are addressed by 32-bit F4 offsets into their own buffer (64 GB).
//   H0 (grid origin xyz, 1 / cell)   H1 (nx, ny, nz, cell)   H2 (centre xyz, R)   H3 (rlimit^2, kappa, triangles, direction maps)
//   H4 (cell_first, cell_cand (positions in the Morton-ordered copy), that copy of the run's records, the band records in the
//       same order: pool offsets)
//   H5 (-, KQ, P / L, KT)          H6 (ball_abs, kr_a, kr_b, ea)
//   H7 (centroid quantisation origin xyz, eps_c)                   H8 (centroid quantisation step xyz, eps_n)
//   per direction map k, one F4: (R, rho_max, first: pool offset, candidates (positions in the Morton copy): pool offset)
enum { kTriPoolHeaderF4 = 9, kTriPoolMaxMaps = 3 };
inline int32_t put_tri_pool(std::vector<F4>& b, PoolLayout& pool, TriPool& tp, const PtHittable* tri, int count) {
  // the run's records in MORTON order of the centroids (R2.w = the triangle's index in the run, for the tie rule): pt_tripool.hpp
  const size_t ntri = (size_t)count;
  const std::vector<uint32_t>& order = tp.order;
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  std::vector<uint32_t> recs;
  recs.reserve(12 * ntri);
  for (uint32_t i : order) {
    const float* f = tri[i].f;
    const float e1[3] = {f[3] - f[0], f[4] - f[1], f[5] - f[2]}, e2[3] = {f[6] - f[0], f[7] - f[1], f[8] - f[2]};
    const uint32_t r[12] = {bits(f[0]), bits(f[1]), bits(f[2]), (uint32_t)tri[i].material, bits(e1[0]), bits(e1[1]), bits(e1[2]), 0u, bits(e2[0]), bits(e2[1]), bits(e2[2]), i};
    recs.insert(recs.end(), r, r + 12);
  }
  const uint32_t tri_sorted = pool.put(std::move(recs), 12);
  std::vector<uint32_t> bandv;
  bandv.reserve(4 * ntri);
  for (uint32_t i : order) bandv.insert(bandv.end(), &tp.band_q[4 * (size_t)i], &tp.band_q[4 * (size_t)i] + 4);
  const uint32_t band = pool.put(std::move(bandv), 260); // (the full stream reads up to 128 records past the last one: dead records)
  std::vector<uint32_t> readyv; // the ready band records, in the same (Morton) order: three F4 per triangle
  readyv.reserve(12 * ntri);
  for (uint32_t i : order) for (int k = 0; k < 12; k++) readyv.push_back(bits(tp.band_ready[12 * (size_t)i + (size_t)k]));
  const uint32_t ready = pool.put(std::move(readyv), 192); // (a trip gathers 64 records; indices past a list's end are clamped to the table)
  const uint32_t cell_first = pool.put(std::move(tp.cell_first), 2);
  const uint32_t cell_cand = pool.put(std::move(tp.cell_cand), 20);
  uint32_t mfirst[kTriPoolMaxMaps] = {0, 0, 0}, mcand[kTriPoolMaxMaps] = {0, 0, 0};
  const int n_maps = std::min((int)tp.maps.size(), (int)kTriPoolMaxMaps);
  for (int k = 0; k < n_maps; k++) {
    TriDirMap& dm = tp.maps[(size_t)k];
    mfirst[k] = pool.put(std::move(dm.first), 2);
    mcand[k] = pool.put(std::move(dm.cand), 200); // (the pipelined scan loads indices up to three trips of 64 PT_BAND_PER (<= 4) past a list's end)
  }
  const int32_t hdr = (int32_t)b.size();
  b.push_back({tp.origin[0], tp.origin[1], tp.origin[2], tp.inv_cell});
  b.push_back({as_f(tp.n[0]), as_f(tp.n[1]), as_f(tp.n[2]), tp.cell});
  b.push_back({tp.centre[0], tp.centre[1], tp.centre[2], tp.R});
  b.push_back({tp.rlimit2, tp.kappa, as_f((int32_t)ntri), as_f(n_maps)});
  b.push_back({as_f((int32_t)cell_first), as_f((int32_t)cell_cand), as_f((int32_t)tri_sorted), as_f((int32_t)band)});
  b.push_back({0.0f, tp.kq, tp.p_per_L, tp.kt});
  b.push_back({tp.ball_abs, tp.kr_a, tp.kr_b, tp.ea});
  b.push_back({tp.cq_lo[0], tp.cq_lo[1], tp.cq_lo[2], tp.eps_c});
  b.push_back({tp.cq_step[0], tp.cq_step[1], tp.cq_step[2], tp.eps_n});
  for (int k = 0; k < kTriPoolMaxMaps; k++) {
    if (k < n_maps) b.push_back({as_f(tp.maps[(size_t)k].R), tp.maps[(size_t)k].rho_max, as_f((int32_t)mfirst[k]), as_f((int32_t)mcand[k])});
    else b.push_back({as_f(0), -1.0f, as_f(0), as_f(0)});
  }
  b.push_back({as_f((int32_t)ready), 0.0f, 0.0f, 0.0f}); // hdr + 12: the ready band records (pt_render.hip: band_kernel)
  return hdr;
}

inline int flatten(const PtSceneDesc* sc, Flat& out, std::string& err, bool allow_grid = true, int box_cull = 1, GridTuning tune = GridTuning(),
                   bool allow_tri_pool = true, TriPoolTuning tri_tune = TriPoolTuning(), bool sphere_merge = true) {
  int rc = validate(sc, err);
  if (rc) return rc;
  out = Flat();
  // materials with their texture inlined
  out.mats.reserve((size_t)sc->n_materials * 4);
  for (int i = 0; i < sc->n_materials; i++) {
    const PtMaterial& m = sc->materials[i];
    F4 M0{as_f(m.kind), as_f(PT_TEX_SOLID), m.param, 1.0f}, M1{m.color[0], m.color[1], m.color[2], as_f(1)},
        M2{0, 0, 0, as_f(1)}, M3{as_f(0), 0, 0, 0};
    if (m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_LIGHTSOURCE || m.kind == PT_MAT_ISOTROPIC) {
      const PtTexture& t = sc->textures[m.texture];
      M0.y = as_f(t.kind);
      M0.w = t.freq;
      M1 = {t.color0[0], t.color0[1], t.color0[2], as_f((int32_t)t.width)};
      M2 = {t.color1[0], t.color1[1], t.color1[2], as_f((int32_t)t.height)};
      M3.x = as_f((int32_t)t.offset);
      if (t.kind == PT_TEX_IMAGE) out.has_image = true;
    }
    out.mats.push_back(M0); out.mats.push_back(M1); out.mats.push_back(M2); out.mats.push_back(M3);
  }
  // runs: maximal stretches of one device kind, in list order (traversal order is semantics:
  // constant_medium draws RNG against the current closest hit, ties resolve by position)
  struct Run { int kind, first, count; };
  std::vector<Run> runs;
  for (int i = 0; i < sc->n_hittables; i++) {
    int dk = device_kind(sc->hittables[i]);
    if (dk == DK_TRI_B) out.has_badouel = true;
    if (runs.empty() || runs.back().kind != dk) runs.push_back({dk, i, 0});
    runs.back().count++;
  }
  for (int i = 0; i < sc->n_hittables; i++) {
    const PtHittable& h = sc->hittables[i];
    bool rectish = h.kind == PT_HIT_XY_RECT || h.kind == PT_HIT_XZ_RECT || h.kind == PT_HIT_YZ_RECT || h.kind == PT_HIT_BOX ||
                   (h.kind == PT_HIT_CONSTANT_MEDIUM && h.boundary_kind == PT_HIT_BOX);
    if (!rectish) continue;
    int nf = (h.kind == PT_HIT_BOX || h.kind == PT_HIT_CONSTANT_MEDIUM) ? 6 : 5;
    for (int k = 0; k < nf; k++)
      if (!(std::fabs(h.f[k]) <= 1.152921504606846976e18f)) out.fast_ok = false; // also false for NaN
    // the straight-line test clamps with v_med3 (rect_fast): every interval must be lo <= hi (an inverted rect or box can
    // never be hit in the reference either, but "median == value" would say otherwise)
    if (nf == 5) { if (!(h.f[0] <= h.f[1] && h.f[2] <= h.f[3])) out.fast_ok = false; }
    else if (!(h.f[0] <= h.f[3] && h.f[1] <= h.f[4] && h.f[2] <= h.f[5])) out.fast_ok = false;
  }
  out.coop_prefix = sc->n_hittables;
  for (int i = 0; i < sc->n_hittables; i++) {
    const PtHittable& h = sc->hittables[i];
    if (h.kind == PT_HIT_CONSTANT_MEDIUM && i < out.coop_prefix) out.coop_prefix = i;
    if (h.kind == PT_HIT_CONSTANT_MEDIUM || h.kind == PT_HIT_TRIANGLE) {
      const PtMaterial& m = sc->materials[h.material];
      bool tex = m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_LIGHTSOURCE || m.kind == PT_MAT_ISOTROPIC;
      if (tex && sc->textures[m.texture].kind == PT_TEX_IMAGE) out.coop_ok = false;
    }
  }
  out.n_runs = (int32_t)runs.size();
  std::vector<F4>& b = out.blob;
  b.resize(runs.size());
  size_t pool_at = 0, pool_x = 0;  // the current slab pool: where its table / its exact entries start, its first hittable, its entries
  int pool_first = 0, pool_n = 0;
  // ---- absorbed sphere runs (round 6) ------------------------------------------------------------------------------------------
  // The reference's default scene ends with thirteen hittables in six runs behind its 483 spheres (pyramid, light, rect, the big spheres,
  // monolith, smoke), and a short run costs a resident kernel ~600 cycles per sphere — its header, its aux record and its records are
  // dependent loads — where a LIST ENTRY of a long run costs ~100.  So a sphere run of more than two spheres also tests, through its lists,
  // the STATIC spheres of later sphere runs that only rect / box runs and short unpooled triangle runs separate it from; the later run
  // is flagged (aux flags bit 3) and the resident kernels skip it.  Why no bit changes: records stay where they are (offsets still compare
  // like list positions), a sphere's t does not depend on the running maximum, and moving a sphere's test EARLIER past kinds that accept
  // t == max (rectangle.hpp:36, triangle.hpp:91) is what the tie rule already covers — list order [X, S], equal t: X holds, S needs
  // t < max and fails; evaluated [S, X]: S holds, X accepts t <= max and replaces it — X either way; sphere against sphere is
  // sphere_finish_unordered's offset rule.  Never past a constant_medium (its draw looks at the running maximum, constant_medium.hpp:52-65),
  // a pooled triangle run (its slots order equal t by offset) or a Badouel run, and not where stale u, v are tracked.  The slab pool's
  // own out-of-order rule learnt the one new case: a LATER holder that is a sphere does not make a rect / box candidate strict.
  std::vector<int> absorbed_by(runs.size(), -1);
  struct Foreign { int run, idx; };
  std::vector<std::vector<Foreign>> foreign_of(runs.size());
  if (sphere_merge && !(out.has_image && !out.coop_ok)) {
    for (size_t ri = 0; ri < runs.size(); ri++) {
      if (runs[ri].kind != DK_SPHERE || runs[ri].count <= 2 || absorbed_by[ri] >= 0) continue;
      { // (a run whose moving spheres have different shutter intervals is scanned sphere by sphere, not through its lists: put_sphere_run_aux)
        bool uniform = true, any = false;
        float t0 = 0.0f, t1 = 0.0f;
        for (int i = 0; i < runs[ri].count; i++) {
          const float* f = sc->hittables[runs[ri].first + i].f;
          if (f[7] == f[8]) continue;
          if (!any) { t0 = f[7]; t1 = f[8]; any = true; }
          else if (std::memcmp(&t0, &f[7], 4) != 0 || std::memcmp(&t1, &f[8], 4) != 0) uniform = false;
        }
        if (!uniform) continue;
      }
      for (size_t rj = ri + 1; rj < runs.size(); rj++) {
        const Run& rr = runs[rj];
        if (rr.kind == DK_RECT || rr.kind == DK_BOX) continue;
        if (rr.kind == DK_TRI && rr.count <= 64 && rr.count < tri_tune.min_run) continue;
        if (rr.kind != DK_SPHERE || absorbed_by[rj] >= 0 || rr.count > 64) break;
        bool all_static = true;
        for (int i = 0; i < rr.count; i++) { const float* f = sc->hittables[rr.first + i].f; if (f[7] != f[8]) all_static = false; }
        if (!all_static) break;
        absorbed_by[rj] = (int)ri;
        for (int i = 0; i < rr.count; i++) foreign_of[ri].push_back({(int)rj, i});
      }
    }
  }
  struct Patch { size_t ri; std::vector<std::pair<size_t, size_t>> ranges; };
  std::vector<Patch> patches;
  for (size_t ri = 0; ri < runs.size(); ri++) {
    const Run& run = runs[ri];
    if (run.kind == DK_SPHERE) {
      std::vector<int32_t> codes;
      for (size_t k = 0; k < foreign_of[ri].size(); k++) codes.push_back(-(int32_t)(1 + k));
      Patch pt{ri, {}};
      out.grid_spheres += put_sphere_run_aux(b, &sc->hittables[run.first], run.count, allow_grid, tune, &codes, &pt.ranges);
      if (!codes.empty()) patches.push_back(std::move(pt));
      if (absorbed_by[ri] >= 0) { int32_t fl; std::memcpy(&fl, &b.back().w, 4); b.back().w = as_f(fl | 8); } // aux flags bit 3: absorbed
    }
    // slab pools (pt_device.hpp: slab_pool): a maximal stretch of consecutive rect / box runs with enough boxes gets a table
    // [n slab entries (lo, -)(hi, -), padded to an even count][n exact entries (lo', hit id)(hi', -)] in front of its first run; every rect / box run
    // carries an aux F4 at its first record - 1: (largest |coordinate| of the pool, runs the pool spans (0: not a pool head),
    // pool offset, n).
    if (run.kind == DK_BOX || run.kind == DK_RECT) {
      const bool prev_rectish = ri > 0 && (runs[ri - 1].kind == DK_BOX || runs[ri - 1].kind == DK_RECT);
      int span = 0, n = 0, n_box = 0;
      if (!prev_rectish) {
        size_t rj = ri;
        while (rj < runs.size() && (runs[rj].kind == DK_BOX || runs[rj].kind == DK_RECT)) {
          n += runs[rj].count;
          if (runs[rj].kind == DK_BOX) n_box += runs[rj].count;
          rj++;
        }
        span = (int)(rj - ri);
      }
      // worth it?  issue slots per ray: straight-line 19 per rect + 114 per box; pool 26 per entry + ~50 (leave-behind proof)
      // + ~150 (one exact trip).  Rects alone never pay; three boxes do.
      const bool pool = box_cull && out.fast_ok && span > 0 && (box_cull == 2 ? n >= 2 : 88 * n_box - 7 * (n - n_box) > 200);
      float bmax = 0.0f;
      int32_t pool_off = 0;
      if (pool) {
        pool_off = (int32_t)b.size();
        pool_first = run.first;
        pool_at = b.size();
        const int ns = n + (n & 1); // slab entries, padded to an even count with an entry no ray can be a candidate for (NaN bounds)
        b.resize(b.size() + 2 * (size_t)ns + 2 * (size_t)n);
        pool_x = pool_at + 2 * (size_t)ns;
        if (ns > n) { const float q = std::nanf(""); b[pool_at + 2 * n] = {q, q, q, 0}; b[pool_at + 2 * n + 1] = {q, q, q, 0}; }
        const float ninf = -INFINITY;
        for (int e = 0; e < n; e++) {
          const PtHittable& h = sc->hittables[run.first + e];
          const float* f = h.f;
          F4 lo, hi, xhi;
          if (h.kind == PT_HIT_BOX) { lo = {f[0], f[1], f[2], 0}; hi = {f[3], f[4], f[5], 0}; xhi = hi; }
          else if (h.kind == PT_HIT_XY_RECT) { lo = {f[0], f[2], f[4], 0}; hi = {f[1], f[3], f[4], 0}; xhi = {f[1], f[3], ninf, 0}; }
          else if (h.kind == PT_HIT_XZ_RECT) { lo = {f[0], f[4], f[2], 0}; hi = {f[1], f[4], f[3], 0}; xhi = {f[1], ninf, f[3], 0}; }
          else { lo = {f[4], f[0], f[2], 0}; hi = {f[4], f[1], f[3], 0}; xhi = {ninf, f[1], f[3], 0}; }
          for (float v : {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}) bmax = std::max(bmax, std::fabs(v));
          b[pool_at + 2 * e] = lo; b[pool_at + 2 * e + 1] = hi;
          b[pool_x + 2 * e] = lo; b[pool_x + 2 * e + 1] = xhi; // .w of the first = hit id: filled in with the record
        }
        pool_n = n;
        out.pooled += n;
      }
      b.push_back({bmax, as_f(pool ? span : 0), as_f(pool_off), as_f(pool ? n : 0)});
    }
    if (run.kind == DK_TRI) { // (Badouel-strategy runs have no pool: one parity-completeness loop)
      int32_t hdr = 0;
      bool pooled = false;
      if (allow_tri_pool) {
        TriPool tp = build_tri_pool(&sc->hittables[run.first], run.count, tri_tune);
        unsigned long long map_f4 = 0; // (the maps count too — a pool's tables are addressed by 32-bit F4 offsets)
        for (const TriDirMap& dm : tp.maps) { map_f4 += (dm.cand.size() + dm.first.size()) / 4 + 140; tri_tune.dm_budget -= (long long)dm.cand.size(); } // the next run gets what is left of the budget
        tri_tune.dm_budget = std::max(0ll, tri_tune.dm_budget);
        if (tp.ok && out.pool.size_f4 + 12ull * (size_t)run.count + tp.cell_cand.size() / 4 + tp.cell_first.size() / 4 + map_f4 + 1024 < (1ull << 31)) {
          out.tri_cells_per_triangle = tp.mean_cells_per_triangle;
          out.tri_wide = tp.wide;
          out.tri_maps = (int)tp.maps.size();
          for (size_t k = 0; k < tp.maps.size() && k < 3; k++) { out.tri_map_entries[k] = (long long)tp.maps[k].cand.size(); out.tri_map_res[k] = tp.maps[k].R; }
          hdr = put_tri_pool(b, out.pool, tp, &sc->hittables[run.first], run.count);
          pooled = true;
          out.tri_pooled += run.count;
          out.tri_pool_runs++;
          out.tri_pool_run = (int)ri; out.tri_pool_hdr = hdr; out.tri_pool_goff = (int32_t)b.size() + 1; out.tri_pool_count = run.count;
        }
      }
      b.push_back({as_f(pooled ? 1 : 0), as_f(hdr), 0, 0});
    }
    b[ri] = {as_f(run.kind | (absorbed_by[ri] >= 0 ? DK_ABSORBED : 0)), as_f((int32_t)b.size()), as_f(run.count), as_f(run.first)};
    for (int i = run.first; i < run.first + run.count; i++) {
      const PtHittable& h = sc->hittables[i];
      const float* f = h.f;
      if (pool_n > 0 && i >= pool_first && i < pool_first + pool_n && (run.kind == DK_BOX || run.kind == DK_RECT))
        b[pool_x + 2 * (size_t)(i - pool_first)].w = as_f((int32_t)((run.kind << kHitKindShift) | (int32_t)b.size())); // hit_pack(kind, 0, offset)
      switch (h.kind) {
        case PT_HIT_SPHERE: put_sphere(b, f, h.material, i); break;
        case PT_HIT_XY_RECT: case PT_HIT_XZ_RECT: case PT_HIT_YZ_RECT: {
          int axis = h.kind == PT_HIT_XY_RECT ? 0 : h.kind == PT_HIT_XZ_RECT ? 1 : 2;
          b.push_back({f[0], f[1], f[2], f[3]});
          b.push_back({f[4], as_f(h.material), as_f(axis), as_f(i)});
          break;
        }
        case PT_HIT_TRIANGLE:
          b.push_back({f[0], f[1], f[2], as_f(h.material)});
          b.push_back({f[3] - f[0], f[4] - f[1], f[5] - f[2], as_f(i)}); // edge1 (triangle.hpp:65)
          b.push_back({f[6] - f[0], f[7] - f[1], f[8] - f[2], 0.0f});    // edge2 (triangle.hpp:66)
          break;
        case PT_HIT_BOX: put_box(b, f, h.material, i); break;
        default: { // constant_medium
          out.has_medium = true;
          int bk = h.boundary_kind == PT_HIT_SPHERE ? DK_SPHERE : DK_BOX;
          b.push_back({as_f(bk), f[9], as_f(h.material), as_f(i)});
          size_t before = b.size();
          if (bk == DK_SPHERE) put_sphere(b, f, h.material, i); else put_box(b, f, h.material, i);
          while (b.size() < before + 3) b.push_back({0, 0, 0, 0});
          break;
        }
      }
    }
  }
  // the absorbed spheres' offsets, now that every run is laid out (relative to the absorbing run's first record, like the run's own)
  for (const Patch& pt : patches) {
    int32_t base;
    std::memcpy(&base, &b[pt.ri].y, 4);
    for (const auto& rg : pt.ranges)
      for (size_t q = rg.first; q < rg.first + rg.second; q++) {
        float* w = &b[q].x;
        for (int j = 0; j < 4; j++) {
          int32_t v;
          std::memcpy(&v, &w[j], 4);
          if (v >= 0) continue;
          const Foreign& fo = foreign_of[pt.ri][(size_t)(-v - 1)];
          int32_t off_j;
          std::memcpy(&off_j, &b[(size_t)fo.run].y, 4);
          w[j] = as_f(off_j + 3 * fo.idx - base);
        }
      }
    out.absorbed_spheres += (int)foreign_of[pt.ri].size();
  }
  if (b.size() >= (1u << kHitOffBits)) { err = "scene too large for 25-bit record offsets (33.5 M records of 16 bytes)"; return PT_ERR_TOO_LARGE; }
  return PT_OK;
}

} // namespace ptf
