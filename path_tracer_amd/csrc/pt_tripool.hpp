// pt_tripool.hpp — host side of the EXACT culling structure for long runs of Moller-Trumbore triangles (SURVEY.md §8f-4:
// "order-preserving culling for large N", config 5: 100 k triangles).  Pure host C++; the device query that reads these
// tables is tri_pool_scan in pt_device.hpp.
//
// The reference scans every triangle for every ray (render.hpp:37-49 -> triangle.hpp:58-100).  A spatial structure alone
// cannot reproduce that scan bit for bit: the binary32 test accepts a triangle whenever its COMPUTED u, v, u + v pass the
// comparisons against the COMPUTED a, and for a ray that grazes the triangle's plane (|a| barely above the 1e-7 cut-off of
// triangle.hpp:71) the rounding noise in u and v exceeds |a| — such a triangle can be "hit" although the ray's line passes
// far from it.  So the candidate set of a ray is built from TWO exact parts (rounds 3-4: three), and a triangle the reference could
// accept is always in at least one of them:
//
//   (1) GRID.   Pairs (ray, triangle) that are NOT grazing, |a^| >= thr_i (a^ = the exact e1 . (d x e2) = -d . N_i).  For those
//       the computed barycentrics are within 1/M of the exact ones, so the exact point P^ where the ray's line meets the
//       triangle's plane lies within sigma_i of the triangle, and the computed t is within the same distance (along the ray,
//       plus a relative 1/M_a) of P^'s exact parameter.  The triangle is listed in every cell of a fine uniform grid that meets
//       its bounding box grown by sigma'_i AND the slab |n^_i . (x - v0)| <= sigma_t,i around its plane (the point the bound below
//       speaks of is in both); every lane walks ITS ray through the cells of its segment [0, closest (1 + kappa)].
//   (2) BAND.   Pairs that ARE grazing, |d^ . n^_i| < tau_i(rho).  Round 5: indexed by the RAY'S DIRECTION, not by the
//       triangle's normal.  A cube map over unit directions (three faces, antipodes identified: the test is even in d); bin D
//       lists every triangle i for which SOME direction of D satisfies |d^ . n^_i| <= tau_i(rho_max) — the triangle's band is a
//       great-circle strip of directions, which central projection turns into a STRAIGHT strip on each face, rasterised per
//       triangle at build time.  A ray reads ONE contiguous list (the bin of its direction); tau grows with rho, so there is a
//       map per class of rho (rho <= rho_max_k) and rays beyond the last class stream every triangle's band record.  Slivers —
//       bands so wide that they cover most directions (rounds 3-4: an "always list") — are simply listed in most bins.
//   Every candidate then runs the reference's own test (tri_eval / tri_finish, the same instructions as the brute-force
//   scan) with the unordered acceptance rule (the last triangle in list order wins an equal t: triangle.hpp:91 accepts
//   t == max), so testing a triangle twice or out of order changes nothing.
//
// ---- the bound (u = 2^-24; binary32 without contraction, gradual underflow; regular ray: 2^-40 <= |d_c| <= 2^40,
//      |o_c| <= 2^60; pool scenes: every triangle coordinate finite, |x| <= 2^20) ---------------------------------------------
// Exact quantities carry a hat.  s^ = o - v0, N = e1 x e2 (e1, e2 the stored edges, exact), a^ = e1 . (d x e2) = -d . N,
// u^ = s^ . (d x e2), v^ = d . (s^ x e1), w^ = e2 . (s^ x e1); the line meets the plane at t^ = w^ / a^ with barycentrics
// beta = u^ / a^, gamma = v^ / a^:  P^ = o + t^ d = v0 + beta e1 + gamma e2.
// Rounding (standard model, dot products of three terms, cross products as two products and a difference):
//     |h - h^| <= sqrt(3) gamma_2 |d| |e2|,  h = fl(d x e2)          |s - s^| <= u |s^|
//     |a - a^| <= 7 u |d| |e1| |e2|                                   =: da
//     |u - u^| <= 8 u |s^| |d| |e2|                                   =: du
//     |v - v^| <= 9.5 u |s^| |d| |e1|,  |w - w^| <= 9.5 u |s^| |e1| |e2|   =: dv, dw
// (+ an absolute 2^-80 for products that underflow: it is folded into Q_i below).  With L_i = max(|e1|, |e2|) and
// rho >= |s^| for every triangle (rho = |o - c| + R, c and R the centre and radius of the v0's):
//     thr_i = |d| (rho P_i + Q_i),   P_i = M 17.5 u L_i SAFE,   Q_i = (M_a 7 u + 4 u) |e1| |e2| SAFE + 2^-40
// bounds M (du + dv) + M_a da from above (SAFE = 1.5 covers the second-order terms and the rounding of P_i, Q_i, rho
// themselves; the 4 u is the cheap test's own a' = fl(d . N'_i) against a^).
// NOT grazing, |a^| >= thr_i, and accepted by the reference (0 <= u/a <= 1, 0 <= v/a, (u + v)/a <= 1 + u, |a| >= 1e-7,
// min <= t <= max) implies
//     eb := |beta - u/a| <= (du + da) / |a^|,  eg := |gamma - v/a| <= (dv + da) / |a^|.
// thr_i bounds the SUM: M SAFE (du + dv) + M_a SAFE da <= thr_i <= |a^| with du + dv <= 17.5 u |s^| |d| L_i and rho >= |s^|.  SAFE = 1.5 was
// chosen for the second-order terms of du, dv, da — a relative 10^-5 — so for this side of the split M_g := 1.4 M and M_ag := 1.4 M_a hold
// with a 7 % reserve for them (round 6; rounds 3-5 let the whole factor go unused here):
//     eb + eg <= S := 1/M_g + 2/M_ag,  each <= T := 1/M_g + 1/M_ag:        beta >= -eb,  gamma >= -eg,  beta + gamma <= 1 + u + eb + eg
// — a triangle of the (beta, gamma) plane with corners A = (-eb, -eg), B = (1 + u + eb + 2 eg, -eg), C = (-eb, 1 + u + 2 eb + eg).  The
// distance to the (convex) triangle is a convex function, so over that region it is largest at a corner: at A, |P^ - v0| <= eb |e1| + eg |e2|;
// at B, |P^ - v1| = |(u + eb + 2 eg) e1 - eg e2| <= (u + eb + eg) |e1| + eg (|e1| + |e2|) <= (u + S) |e1| + T (|e1| + |e2|); C likewise with |e2|:
//     => dist(P^, triangle) <= sigma_b,i := (S + u) L_i + T (|e1| + |e2|)          (<= (3/M_g + 4/M_ag + u) L_i)
// (rounds 3-5 bounded every barycentric separately, clamped in two steps and used M itself: (6/M + 6/M_a) L_i — the same structure, three times as fat.)
//     |t - t^| |d| <= |d| dw / |a| + |t^| |d| da / |a| + u |t| |d|,   |d| dw / |a| <= 9.5 |e1| |e2| / (17.5 M_g L_i (1 - 1/M_ag)) <= 0.545 l_i / M_g
// with l_i = min(|e1|, |e2|) (|e1| |e2| = l_i L_i), the other two relative: <= |t^| |d| / (M_ag - 1) + u |t| |d|;
// so with kappa = 2.2 / (M_a - 1) the exact parameter t^ lies in [-0.55 l_i / (M_g |d|), max (1 + kappa) + 0.55 l_i / (M_g |d|)]
// and the point P' = o + clamp(t^, 0, max (1 + kappa)) d of the WALKED segment is within
//     sigma_t,i = 0.56 l_i / (M_g - 1)  of P^ (along the ray),  hence within
//     sigma'_i  = sigma_b,i + sigma_t,i        (0.43 L_i at M = 6 for |e1| = |e2|; M_a >= 64; the builder keeps M >= 4)
// of the triangle — more precisely P^ lies in the ENLARGED triangle T+_i = { v0 + beta e1 + gamma e2 : beta >= -T, gamma >= -T, beta + gamma <= 1 + u + S }
// (the union of the regions above over the admissible eb, eg) and P' within sigma_t,i of T+_i, which is what the cells are listed by (build_tri_pool:
// for_cells) — hence inside its bounding box grown by sigma'_i: the cell that contains P' lists the triangle, and the walk
// visits that cell (cells are assigned with a further absolute slack for the walk's own rounding, as the sphere grid's).
// P^ lies IN the triangle's plane (it is where the line meets it), so P' is within sigma_t,i of that plane: |n^_i . (P' - v0)| <=
// sigma_t,i (round 6; rounds 3-5 used their sigma'_i = 0.63 L_i here as well, where sigma_t,i is 0.076 l_i at M = 6): a cell with centre m and half
// edge h that contains a point within `slack` of P' has |n^_i . (m - v0)| <= sigma_t,i + slack + h (|n^x| + |n^y| + |n^z|) — cells of
// the grown box that fail this are not listed (the slab is what decides most cells) — and likewise the centre of
// such a cell is within sigma'_i + slack + the cell's half diagonal of the TRIANGLE itself (point-triangle distance), which rounds the
// box's corners off.  Every entry also records which of the cell's six face neighbours list the triangle too: a walk that enters a
// cell through a face whose other side listed the triangle has tested the pair already (a pair's test does not depend on the cell).
// Grazing, |a^| < thr_i, is |d^ . N_i| < rho P_i + Q_i: the band test, evaluated with N'_i = N_i rounded to binary32.
// The direction map lists triangle i in bin D = (face k, [p0, p1] x [q0, q1]) — directions d ~ e_k + p e_a + q e_b — iff
//     min over the bin's rectangle (grown by eps_bin for the device's own rounding of p, q) of |n_k + p n_a + q n_b|
//         <=  tau_i(rho_max) sqrt(1 + p^2 + q^2)max over the rectangle,
// which holds whenever some direction of the bin passes the real band test |d^ . n^_i| <= tau_i(rho) for a rho <= rho_max
// (tau_i grows with rho).  The device picks face k = argmax |d_c| (exact comparisons), p = d_a / d_k, q = d_b / d_k through a
// reciprocal (|error| < 4e-7 for |p|, |q| <= 1) and reads bin (floor((p + 1) R / 2), floor((q + 1) R / 2)), clamped: eps_bin = 4e-6.
// tests/test_tripool_cpu.py checks these inequalities on float32 emulations of the reference's test (random and adversarial
// grazing rays); the GPU suite checks the walked structure against the oracle's brute-force scan, bit for bit.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pt_render.h"

namespace ptf {

struct TriPoolTuning {
  float M = 6.0f;       // PT_TRI_M: barycentric slack 1/M — the grid's boxes grow with 1/M (sigma'), the bands with M.  Swept on cfg5 every round (docs/EXPERIMENTS.md); round 6, as the grid's slack was re-derived and its listing made exact (three, then five times fewer entries at the same M), the optimum moved from 12 to 8 to 6: 1080p x 32 spp 711 / 680 / 660 ms at M = 10 / 8 / 6
  float Ma = 256.0f;    // relative slack of t: the walk runs to max (1 + 2.2 / (Ma - 1))
  float cell = 0.30f;   // PT_TRI_CELL: grid cell edge in units of the median grown box extent (round 6, M = 8: 746 / 723 / 744 / 768 ms at 0.22 / 0.35 / 0.5 / 0.7)
  float grid_budget = 160.0f;         // cell entries per triangle the grid may take (cells are enlarged until the estimate fits)
  // Direction maps, one per class of rho (tau_i grows with rho = |o - c| + R, and a map lists by the tau of its class's largest rho):
  // class 0 serves the rays that START ON THE MESH OR NEXT TO IT (|o - c| <= R + 2 L: rho <= 2 R + 2 L — every secondary ray off a
  // triangle, half of all rays; its bands are half as wide as the next class's), class 1 a camera a few radii out, class 2 the far
  // rest (ground hits towards the horizon); beyond the last class a ray streams every band record.
  // Round 6: {256, 256, 64} -> {128, 64, 32}.  Camera rays take their candidates from their pixel's cache (pt_device.hpp: TriPrimCtx), so the
  // class they fall into is enumerated once per pixel, not once per sample, and the maps' size stopped paying for itself: 1080p x 32 spp 953 ms
  // with 3.0 GB of tables built in 2.1 s, 979 ms with 0.62 GB built in 0.9 s (profiles/r06_tri_map_res_sweep.txt).
  int dm_res[3] = {128, 64, 32};            // PT_TRI_RES=a,b,c: resolution of the maps (<= 1024; 0: no such map)
  float dm_rho[3] = {2.12f, 4.0f, 16.0f};   // PT_TRI_RHO=a,b,c: class k serves rays with rho <= dm_rho[k] * R (R = radius of the v0's)
  long long dm_budget = 400ll << 20;  // entries (4 bytes each) the direction maps may take together (all pooled runs of a scene: the flattener hands the rest on); a map that does not fit is built at half the resolution, or not at all
  int min_run = 4096;   // PT_TRI_MIN: shorter triangle runs are scanned as before (PT_TRICULL=1: 256)
  int threads = 0;      // build threads (0: hardware concurrency, at most 16); the tables do not depend on it
};

struct TriDirMap {
  int R = 0;
  float rho_max = 0;                // rays with rho <= rho_max read this map
  std::vector<uint32_t> first;      // 3 R R + 1 prefix offsets; bin = (face * R + row(q)) * R + column(p)
  std::vector<uint32_t> cand;       // position in the Morton-ordered copy of the run, ascending within a bin
};

struct TriPool {
  bool ok = false;
  float origin[3] = {0, 0, 0}, inv_cell = 0, cell = 0;
  int n[3] = {1, 1, 1};
  float centre[3] = {0, 0, 0}, R = 0, rlimit2 = 0, kappa = 0;
  std::vector<uint32_t> cell_first; // n cells + 1
  std::vector<uint32_t> cell_cand;  // [25:0] position in the Morton-ordered copy of the run, ascending within a cell; [31:26] which face neighbours list it too
  std::vector<uint32_t> order;      // position in the Morton-ordered copy -> triangle index in the run
  std::vector<TriDirMap> maps;      // by ascending rho_max
  std::vector<float> ball;          // 4 floats per triangle: centroid C, L = longest stored edge (every vertex is within L of C)
  std::vector<char> dead;           // triangles no ray can hit (an edge of zero length): in no table
  float p_per_L = 0, ball_abs = 0, kr_a = 0, kr_b = 0, ea = 0; // constants of the noise-radius filter (see build_tri_pool)
  // COMPRESSED band records (what the device gathers: the filters are necessary conditions, so any relaxation of them is
  // still exact — see "compressed records" in build_tri_pool): 4 dwords per triangle
  float cq_lo[3] = {0, 0, 0}, cq_step[3] = {0, 0, 0}, eps_c = 0, eps_n = 0, kq = 0, kt = 0;
  std::vector<uint32_t> band_q; // (nq.x | nq.y << 16) (nq.z | bf16(pn) << 16) (cq.x | cq.y << 16) (cq.z | bf16(L) << 16)
  // READY band records (round 6; what the binned band stage reads, pt_render.hip: band_kernel — one record serves the 64 rays of a packet, so
  // it is stored decoded, 12 floats per triangle): (nq.x, nq.y, nq.z, pn~) (C~.x, C~.y, C~.z, L~ + ball_abs + eps_c) (G, nlow, E2, KR) with
  // the per-triangle sub-expressions of the compressed record's filter precomputed from the SAME decoded values, each rounded to the side
  // that relaxes the filter: see "ready records" in build_tri_pool
  std::vector<float> band_ready;
  // statistics for the tests / DESIGN
  double mean_cells_per_triangle = 0;
  int wide = 0; // triangles whose band at the first map's rho_max covers every direction (tau >= 1, or no normal at all)
};

namespace detail {
// squared distance from the point p to the triangle (a, a + e1, a + e2) — the closest-point regions of a triangle (vertex / edge / face)
inline double point_triangle_dist2(const double p[3], const double a[3], const double e1[3], const double e2[3]) {
  auto dot = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
  double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
  const double d1 = dot(e1, ap), d2 = dot(e2, ap);
  auto len2 = [&](double s, double t) { double w[3]; for (int k = 0; k < 3; k++) w[k] = ap[k] - s * e1[k] - t * e2[k]; return dot(w, w); };
  if (d1 <= 0 && d2 <= 0) return len2(0, 0);                                   // vertex a
  double bp[3] = {ap[0] - e1[0], ap[1] - e1[1], ap[2] - e1[2]};
  const double d3 = dot(e1, bp), d4 = dot(e2, bp);
  if (d3 >= 0 && d4 <= d3) return len2(1, 0);                                  // vertex b
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) return len2(d1 / (d1 - d3), 0);           // edge ab
  double cp[3] = {ap[0] - e2[0], ap[1] - e2[1], ap[2] - e2[2]};
  const double d5 = dot(e1, cp), d6 = dot(e2, cp);
  if (d6 >= 0 && d5 <= d6) return len2(0, 1);                                  // vertex c
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) return len2(0, d2 / (d2 - d6));           // edge ac
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); return len2(1 - w, w); } // edge bc
  const double den = va + vb + vc;
  if (!(den > 0)) return std::min(len2(0, 0), std::min(len2(1, 0), len2(0, 1))); // (degenerate: a vertex is never farther than the true distance + an edge)
  return len2(vb / den, vc / den);                                             // the face
}
// Does the axis-aligned box (centre m, half extents hh) meet the triangle (a, b, c)?  Separating axes: the box's three, the triangle's
// normal, the nine products of a box axis and a triangle edge (Akenine-Moller).  Touching counts as meeting; a degenerate triangle's
// vanishing axes separate nothing (conservative).
inline bool tri_box_overlap(const double m[3], const double hh[3], const double a[3], const double b[3], const double c[3]) {
  double v[3][3];
  for (int k = 0; k < 3; k++) { v[0][k] = a[k] - m[k]; v[1][k] = b[k] - m[k]; v[2][k] = c[k] - m[k]; }
  for (int k = 0; k < 3; k++) {
    const double mn = std::min({v[0][k], v[1][k], v[2][k]}), mx = std::max({v[0][k], v[1][k], v[2][k]});
    if (mn > hh[k] || mx < -hh[k]) return false;
  }
  double f[3][3];
  for (int k = 0; k < 3; k++) { f[0][k] = v[1][k] - v[0][k]; f[1][k] = v[2][k] - v[1][k]; f[2][k] = v[0][k] - v[2][k]; }
  const double n[3] = {f[0][1] * f[1][2] - f[0][2] * f[1][1], f[0][2] * f[1][0] - f[0][0] * f[1][2], f[0][0] * f[1][1] - f[0][1] * f[1][0]};
  if (std::fabs(n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2]) > hh[0] * std::fabs(n[0]) + hh[1] * std::fabs(n[1]) + hh[2] * std::fabs(n[2])) return false;
  for (int j = 0; j < 3; j++) {
    const double ax[3][3] = {{0.0, -f[j][2], f[j][1]}, {f[j][2], 0.0, -f[j][0]}, {-f[j][1], f[j][0], 0.0}}; // e_i x f_j
    for (int i = 0; i < 3; i++) {
      const double p0 = ax[i][0] * v[0][0] + ax[i][1] * v[0][1] + ax[i][2] * v[0][2];
      const double p1 = ax[i][0] * v[1][0] + ax[i][1] * v[1][1] + ax[i][2] * v[1][2];
      const double p2 = ax[i][0] * v[2][0] + ax[i][1] * v[2][1] + ax[i][2] * v[2][2];
      const double r = hh[0] * std::fabs(ax[i][0]) + hh[1] * std::fabs(ax[i][1]) + hh[2] * std::fabs(ax[i][2]);
      if (std::min({p0, p1, p2}) > r || std::max({p0, p1, p2}) < -r) return false;
    }
  }
  return true;
}
// f(p, q) = nk + p na + nb q over the rectangle [pl, ph] x [ql, qh]: is min |f| <= W possible, and for which p?  Returns false when no p qualifies.
inline bool strip_columns(double nk, double na, double nb, double ql, double qh, double W, double& pa, double& pb) {
  const double lo_c = nk + std::min(ql * nb, qh * nb), hi_c = nk + std::max(ql * nb, qh * nb); // f_min(p) = p na + lo_c, f_max(p) = p na + hi_c
  // need f_min(p) <= W and f_max(p) >= -W
  if (na == 0.0) { pa = -2.0; pb = 2.0; return lo_c <= W && hi_c >= -W; }
  const double x0 = (-W - hi_c) / na, x1 = (W - lo_c) / na;
  pa = std::min(x0, x1); pb = std::max(x0, x1);
  return true;
}
} // namespace detail

// one direction map: triangle i (unit normal nrm, band half-width tau[i] at rho_max; tau = INFINITY: every bin) -> bins
// `order`: position in the Morton-ordered copy of the run -> triangle; the map lists POSITIONS, ascending within a bin (a bin's gathers walk
// the record arrays forwards), which falls out of visiting the triangles in that order.
inline bool build_dir_map(TriDirMap& dm, int R, const std::vector<double>& nrm, const std::vector<double>& tau, const std::vector<char>& dead, int count,
                          const std::vector<uint32_t>& order, long long budget, int n_threads) {
  dm.R = R;
  const size_t nb = (size_t)3 * R * R;
  const double eps_bin = 4e-6, step = 2.0 / R;
  const int T = std::max(1, n_threads);
  // every thread walks its own contiguous slice of the triangles, in order; pass 0 counts per (thread, bin), pass 1 writes: the table is
  // the single-threaded one whatever T is
  auto raster = [&](int i, auto&& emit) { // emit(first bin of a row, columns c0 .. c1)
    const double* nn = &nrm[(size_t)i * 3];
    const double t = tau[(size_t)i];
    if (!(t < 1.0)) { for (size_t b = 0; b < nb; b += (size_t)R) emit(b, 0, R - 1); return; } // covers every direction
    for (int k = 0; k < 3; k++) {
      const int a = (k + 1) % 3, b = (k + 2) % 3;
      const double nk = nn[k], na = nn[a], nbv = nn[b];
      const double W3 = t * 1.7320508075688772 * (1 + 1e-9) + 1e-12;
      if (std::fabs(nk) - (std::fabs(na) + std::fabs(nbv)) * (1.0 + eps_bin) > W3) continue; // the strip misses this face
      for (int j = 0; j < R; j++) {
        const double ql = -1.0 + j * step - eps_bin, qh = -1.0 + (j + 1) * step + eps_bin;
        const double qm2 = std::max(ql * ql, qh * qh);
        double W = t * std::sqrt(2.0 + qm2 + 4 * eps_bin) * (1 + 1e-9) + 1e-12, pa, pb; // |p| <= 1 + eps_bin
        if (!detail::strip_columns(nk, na, nbv, ql, qh, W, pa, pb)) continue;
        pa = std::max(pa, -1.0 - eps_bin); pb = std::min(pb, 1.0 + eps_bin);
        if (pa > pb) continue;
        { // once more with the |p| the first pass allows (still conservative: every qualifying p lies inside [pa, pb])
          const double pm2 = std::max(pa * pa, pb * pb);
          W = t * std::sqrt(1.0 + pm2 + qm2) * (1 + 1e-9) + 1e-12;
          double pa2, pb2;
          if (!detail::strip_columns(nk, na, nbv, ql, qh, W, pa2, pb2)) continue;
          pa = std::max(pa, pa2); pb = std::min(pb, pb2);
          if (pa > pb) continue;
        }
        const int c0 = std::max(0, std::min(R - 1, (int)std::floor((pa - eps_bin + 1.0) * 0.5 * R)));
        const int c1 = std::max(0, std::min(R - 1, (int)std::floor((pb + eps_bin + 1.0) * 0.5 * R)));
        emit(((size_t)k * R + j) * R, c0, c1);
      }
    }
  };
  std::vector<std::vector<uint32_t>> cnt((size_t)T);
  auto slice = [&](int t, int& i0, int& i1) { i0 = (int)((long long)count * t / T); i1 = (int)((long long)count * (t + 1) / T); };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        cnt[(size_t)t].assign(nb, 0);
        int i0, i1;
        slice(t, i0, i1);
        uint32_t* const cn = cnt[(size_t)t].data();
        for (int p = i0; p < i1; p++) {
          const int i = (int)order[(size_t)p];
          if (!dead[(size_t)i]) raster(i, [&](size_t row, int c0, int c1) { for (int c = c0; c <= c1; c++) cn[row + (size_t)c]++; });
        }
      });
    for (auto& x : th) x.join();
  }
  dm.first.assign(nb + 1, 0);
  unsigned long long total = 0;
  for (size_t b = 0; b < nb; b++) {
    dm.first[b] = (uint32_t)total;
    for (int t = 0; t < T; t++) { const uint32_t c = cnt[(size_t)t][b]; cnt[(size_t)t][b] = (uint32_t)total; total += c; } // -> this thread's cursor in the bin
    if (total > (unsigned long long)budget || total >= (1ull << 32)) return false;
  }
  dm.first[nb] = (uint32_t)total;
  dm.cand.resize((size_t)total);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        int i0, i1;
        slice(t, i0, i1);
        uint32_t* const cur = cnt[(size_t)t].data();
        uint32_t* const out = dm.cand.data();
        for (int p = i0; p < i1; p++) {
          const int i = (int)order[(size_t)p];
          if (!dead[(size_t)i]) raster(i, [&](size_t row, int c0, int c1) { for (int c = c0; c <= c1; c++) out[cur[row + (size_t)c]++] = (uint32_t)p; });
        }
      });
    for (auto& x : th) x.join();
  }
  return true;
}

inline TriPool build_tri_pool(const PtHittable* h, int count, TriPoolTuning tune = TriPoolTuning()) {
  TriPool tp;
  const bool timing = std::getenv("PT_TRI_TIMING") != nullptr; // (diagnostics: stage times on stderr)
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "build_tri_pool: %-28s %.3f s\n", what, std::chrono::duration<double>(t - t_prev).count());
    t_prev = t;
  };
  if (count < tune.min_run || count >= (1 << 26)) return tp; // (table entries carry a triangle's position in 26 bits)
  const double u = std::ldexp(1.0, -24), SAFE = 1.5;
  // (the bound of the header holds for any M > 1; sigma' below is its general form.  M >= 4 keeps the second-order terms SAFE covers small.)
  const double M = std::max(4.0, (double)tune.M), Ma = std::max(64.0, (double)tune.Ma);
  std::vector<double> P((size_t)count), Q((size_t)count), pn((size_t)count), qn((size_t)count), sig((size_t)count);
  std::vector<double> nrm((size_t)count * 3);
  std::vector<char> slab_ok((size_t)count, 0); // the unit normal is well conditioned in binary64 (sin of the edges' angle >= 1e-6): the plane slab may be used
  std::vector<char>& dead = tp.dead;
  dead.assign((size_t)count, 0);
  double c[3] = {0, 0, 0};
  for (int i = 0; i < count; i++) {
    const float* f = h[i].f;
    for (int k = 0; k < 9; k++)
      if (!(std::fabs(f[k]) <= 1048576.0f)) return tp; // also NaN: no pool
    for (int k = 0; k < 3; k++) c[k] += f[k];
  }
  for (int k = 0; k < 3; k++) c[k] /= count;
  double R = 0;
  std::vector<double> ext;
  ext.reserve((size_t)count);
  tp.ball.assign((size_t)count * 4, 0.0f);
  int n_live = 0;
  const double Mg = 1.4 * M, Mag = 1.4 * Ma;                     // what thr_i (with its SAFE = 1.5) gives the grid's side of the split (header)
  const double bS = 1.0 / Mg + 2.0 / Mag, bT = 1.0 / Mg + 1.0 / Mag;
  std::vector<double> sigt((size_t)count, 0.0);                  // sigma_t,i: P' from P^, along the ray — the plane slab's half thickness
  for (int i = 0; i < count; i++) {
    const float* f = h[i].f;
    // the edges as the flattener stores them (binary32 differences: triangle.hpp:65-66)
    const float e1f[3] = {f[3] - f[0], f[4] - f[1], f[5] - f[2]}, e2f[3] = {f[6] - f[0], f[7] - f[1], f[8] - f[2]};
    const double e1[3] = {e1f[0], e1f[1], e1f[2]}, e2[3] = {e2f[0], e2f[1], e2f[2]};
    const double N[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    const double L = std::max(l1, l2), nN = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
    double dv = 0;
    for (int k = 0; k < 3; k++) dv += (f[k] - c[k]) * (f[k] - c[k]);
    R = std::max(R, std::sqrt(dv));
    // a = e1 . (d x e2) is a sum of products of edge components: an edge pair whose cross product is exactly 0 in every
    // component the reference can form (both edges zero, or one zero) gives a = +-0 for every ray: |a| < 1e-7, never accepted
    if (!(L > 0.0) || l1 == 0.0 || l2 == 0.0) { dead[(size_t)i] = 1; continue; }
    n_live++;
    P[(size_t)i] = M * 17.5 * u * L * SAFE;
    Q[(size_t)i] = (Ma * 7.0 + 4.0) * u * l1 * l2 * SAFE + std::ldexp(1.0, -40);
    sigt[(size_t)i] = 0.56 * std::min(l1, l2) / (Mg - 1.0) * (1 + 8 * u);
    sig[(size_t)i] = ((bS + 2 * u) * L + bT * (l1 + l2)) * (1 + 8 * u) + sigt[(size_t)i]; // sigma'_i: P' from the triangle
    if (nN > 0.0) {
      pn[(size_t)i] = P[(size_t)i] / nN; qn[(size_t)i] = Q[(size_t)i] / nN;
      for (int k = 0; k < 3; k++) nrm[(size_t)i * 3 + k] = N[k] / nN;
      slab_ok[(size_t)i] = nN >= 1e-6 * l1 * l2;
    } else { pn[(size_t)i] = qn[(size_t)i] = INFINITY; }
    for (int k = 0; k < 3; k++) tp.ball[(size_t)i * 4 + k] = (float)(f[k] + (e1[k] + e2[k]) / 3.0); // centroid of v0, v0 + e1, v0 + e2
    tp.ball[(size_t)i * 4 + 3] = (float)(L * (1 + 2 * u));
    double emax = 0;
    for (int k = 0; k < 3; k++) {
      const double lo = std::min({(double)f[k], (double)f[3 + k], (double)f[6 + k]}), hi = std::max({(double)f[k], (double)f[3 + k], (double)f[6 + k]});
      emax = std::max(emax, hi - lo + 2 * sig[(size_t)i]);
    }
    ext.push_back(emax);
  }
  if (n_live < tune.min_run) return tp;
  R *= 1.0 + 8 * u;
  lap("per-triangle constants");
  // ---- (1) the grid -----------------------------------------------------------------------------------------------------
  std::nth_element(ext.begin(), ext.begin() + ext.size() / 2, ext.end());
  double cell = std::max(1e-6, (double)tune.cell * ext[ext.size() / 2]);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  auto box_of = [&](int i, double slack, double blo[3], double bhi[3]) {
    const float* f = h[i].f;
    for (int k = 0; k < 3; k++) {
      // the vertices as the reference sees them: v0, v0 + e1, v0 + e2 with the STORED (rounded) edges
      const double a = f[k], b = (double)f[k] + (double)(float)(f[3 + k] - f[k]), cc = (double)f[k] + (double)(float)(f[6 + k] - f[k]);
      blo[k] = std::min({a, b, cc}) - sig[(size_t)i] - slack; bhi[k] = std::max({a, b, cc}) + sig[(size_t)i] + slack;
    }
  };
  for (int i = 0; i < count; i++) {
    if (dead[(size_t)i]) continue;
    double blo[3], bhi[3];
    box_of(i, 0.0, blo, bhi);
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[k]); hi[k] = std::max(hi[k], bhi[k]); }
  }
  for (;;) { // at most 2^22 cells, at most 512 per axis
    double total = 1;
    bool fits = true;
    for (int k = 0; k < 3; k++) {
      const double nk = std::max(1.0, std::ceil((hi[k] - lo[k] + 1.6e-2 * cell) / cell));
      if (nk > 512) fits = false;
      tp.n[k] = (int)std::min(nk, 512.0);
      total *= nk;
    }
    // ... and a bounded candidate table: a few triangles that span the whole grid (cells are sized for the median one) would
    // otherwise list themselves in every cell (the estimate counts box cells; the plane slab keeps about a third of them)
    double entries = 0;
    if (fits && total <= 4194304.0) {
      for (int i = 0; i < count && entries <= 1e10; i++) {
        if (dead[(size_t)i]) continue;
        double blo[3], bhi[3], e = 1;
        box_of(i, 0.0, blo, bhi);
        for (int k = 0; k < 3; k++) e *= std::min((double)tp.n[k], (bhi[k] - blo[k]) / cell + 2.0);
        entries += e;
      }
      if (entries <= std::max((double)tune.grid_budget * (double)n_live, 65536.0)) break;
    }
    cell *= 1.25;
  }
  const double slack = 4e-3 * cell; // the walk's own rounding (the ray must start within rlimit: below)
  double half_diag2 = 0;
  for (int k = 0; k < 3; k++) {
    const double extk = tp.n[k] * cell, mid = 0.5 * (lo[k] + hi[k]);
    tp.origin[k] = (float)(mid - 0.5 * extk);
    tp.centre[k] = (float)c[k];
    half_diag2 += 0.25 * extk * extk;
  }
  tp.cell = (float)cell;
  tp.inv_cell = (float)(1.0 / cell);
  tp.R = (float)R;
  tp.kappa = (float)(2.2 / (Ma - 1.0));
  // ---- the distance filter in front of the exact test of a BAND candidate (a necessary condition of an acceptance, so it cannot lose one)
  // Whatever |a^| is, an accepted pair has |beta - u/a| <= (du + da) / |a^| (same for gamma), so P^ — on the ray's line — lies
  // within r = 6 (du + dv + da) L_i / |a^| of the triangle: the line passes within L_i + r of the centroid C_i (every point of the
  // triangle is within L_i of it), with
  //     r <= kr(L_i) rho |d| / (|a'| - ea(L_i) |d|),  kr = 6 SAFE u L^2 (17.5 + 7 L / R),  ea = 4 u L^2  (|a' - a^| <= ea |d|;
  //     rho >= R lets the da term ride on rho).  The device evaluates |(C - o) x d|^2 <= radius^2 |d|^2 in binary32: its
  //     rounding (and the centroid's) is covered by ball_abs = 64 u (rlimit + R + diagonal) added to every radius.
  tp.p_per_L = (float)(M * 17.5 * u * SAFE);
  tp.kr_a = (float)(6 * SAFE * u * 17.5 * (1 + 8 * u));
  tp.kr_b = (float)(6 * SAFE * u * 7.0 / std::max(R, 1e-30) * (1 + 8 * u));
  tp.ea = (float)(4 * u * (1 + 8 * u));
  // a ray's cell coordinates carry ~4 u (|o - origin| + |t d|) of rounding: with the origin within rl of the v0's centre that
  // is <= 8 u (rl + R + diagonal), which must stay below slack / 4
  {
    double cd2 = 0;
    for (int k = 0; k < 3; k++) cd2 += (c[k] - (tp.origin[k] + 0.5 * tp.n[k] * cell)) * (c[k] - (tp.origin[k] + 0.5 * tp.n[k] * cell));
    const double rl = slack / (32 * u) - R - 2 * std::sqrt(half_diag2) - std::sqrt(cd2);
    if (!(rl > 0)) return tp;
    tp.rlimit2 = (float)(rl * rl * 0.99);
    tp.ball_abs = (float)(64 * u * (rl + R + 2 * std::sqrt(half_diag2)));
  }
  const double inv = (double)tp.inv_cell; // assign with the float value the device uses
  const size_t ncell = (size_t)tp.n[0] * tp.n[1] * tp.n[2];
  // Which cells list triangle i (header, "the region of P^"): P^ lies in the ENLARGED triangle T+ — the image of beta >= -T, gamma >= -T,
  // beta + gamma <= 1 + u + S, a copy of the triangle scaled by 1 + u + S + 2 T from the corner v0 - T (e1 + e2) — and P' within r = sigma_t +
  // slack of it, so a cell is listed iff its box grown by r meets T+ (an exact triangle-box test: no bounding ball around the cell) AND
  // meets the slab |n^ . (x - v0)| <= sigma_t + slack around the plane.  (Rounds 3-5 and the first half of round 6 listed the cells whose CENTRE
  // is within sigma' + half a cell diagonal of the triangle: the rounded offset of the triangle by the corners' distance, and a ball around the cell.)
  auto for_cells = [&](int i, auto&& emit) {
    const float* f = h[i].f;
    const double v0d[3] = {f[0], f[1], f[2]};
    const double e1d[3] = {(double)(float)(f[3] - f[0]), (double)(float)(f[4] - f[1]), (double)(float)(f[5] - f[2])};
    const double e2d[3] = {(double)(float)(f[6] - f[0]), (double)(float)(f[7] - f[1]), (double)(float)(f[8] - f[2])};
    const double Tq = bT * (1 + 1e-6), top = (1.0 + 2 * u + bS + bT) * (1 + 1e-6);
    double A[3], B[3], C[3];
    for (int k = 0; k < 3; k++) {
      A[k] = v0d[k] - Tq * e1d[k] - Tq * e2d[k];
      B[k] = v0d[k] + top * e1d[k] - Tq * e2d[k];
      C[k] = v0d[k] - Tq * e1d[k] + top * e2d[k];
    }
    const double sig_t = sigt[(size_t)i], r = sig_t + slack;
    int c0[3], c1[3];
    for (int k = 0; k < 3; k++) {
      const double blo = std::min({A[k], B[k], C[k]}) - r, bhi = std::max({A[k], B[k], C[k]}) + r;
      c0[k] = std::max(0, std::min(tp.n[k] - 1, (int)std::floor((blo - tp.origin[k]) * inv - 1e-9)));
      c1[k] = std::max(0, std::min(tp.n[k] - 1, (int)std::floor((bhi - tp.origin[k]) * inv + 1e-9)));
    }
    const bool flat = slab_ok[(size_t)i] != 0; // has a normal that binary64 resolves (else: every cell of T+'s grown box)
    const double* nn = &nrm[(size_t)i * 3];
    const double hc = 0.5 / inv, reach = r + hc * (std::fabs(nn[0]) + std::fabs(nn[1]) + std::fabs(nn[2])) * (1 + 1e-9) + 1e-9 * cell;
    const double hx = hc * (1 + 1e-9) + r + 1e-9 * cell;
    const double hh[3] = {hx, hx, hx};
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) {
          if (flat) {
            const double m[3] = {tp.origin[0] + (x + 0.5) / inv, tp.origin[1] + (y + 0.5) / inv, tp.origin[2] + (z + 0.5) / inv};
            if (std::fabs(nn[0] * (m[0] - f[0]) + nn[1] * (m[1] - f[1]) + nn[2] * (m[2] - f[2])) > reach) continue;
            if (!detail::tri_box_overlap(m, hh, A, B, C)) continue;
          }
          emit(((size_t)z * tp.n[1] + y) * tp.n[0] + x);
        }
  };
  // The survivors of the filters gather their triangle's records, and a cell's candidates are neighbours in space: the device reads a
  // copy of the run's records in MORTON order of the centroids (21 bits per axis over the centroids' box), and every table lists a triangle
  // by its position in that copy, ascending — which falls out of filling the tables in that order.
  {
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < count; i++)
      for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], (double)tp.ball[(size_t)i * 4 + k]); chi[k] = std::max(chi[k], (double)tp.ball[(size_t)i * 4 + k]); }
    auto spread = [](uint64_t x) { x &= 0x1fffffull; x = (x | (x << 32)) & 0x1f00000000ffffull; x = (x | (x << 16)) & 0x1f0000ff0000ffull;
                                   x = (x | (x << 8)) & 0x100f00f00f00f00full; x = (x | (x << 4)) & 0x10c30c30c30c30c3ull; return (x | (x << 2)) & 0x1249249249249249ull; };
    std::vector<uint64_t> code((size_t)count);
    for (int i = 0; i < count; i++) {
      uint64_t q[3];
      for (int k = 0; k < 3; k++) { const double w = chi[k] - clo[k]; q[k] = w > 0 ? (uint64_t)std::min(2097151.0, std::max(0.0, ((double)tp.ball[(size_t)i * 4 + k] - clo[k]) / w * 2097151.0)) : 0; }
      code[(size_t)i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
    }
    tp.order.resize((size_t)count);
    for (int i = 0; i < count; i++) tp.order[(size_t)i] = (uint32_t)i;
    std::stable_sort(tp.order.begin(), tp.order.end(), [&](uint32_t x, uint32_t y) { return code[x] < code[y]; });
  }
  lap("grid sizing + Morton order");
  // Two passes over the triangles in Morton order, each thread its own contiguous slice (pass 0 counts per (thread, cell), pass 1 writes):
  // a cell's candidates ascend, and the table is the single-threaded one whatever the number of threads (round 6: this was one thread,
  // a third of the scene's build time).
  int GT = tune.threads > 0 ? tune.threads : (int)std::thread::hardware_concurrency();
  GT = std::max(1, std::min(GT, 16));
  if (ncell * (size_t)GT > (64u << 20)) GT = std::max(1, (int)((64u << 20) / ncell)); // (per-thread counters: at most 256 MB)
  auto gslice = [&](int t, int& p0, int& p1) { p0 = (int)((long long)count * t / GT); p1 = (int)((long long)count * (t + 1) / GT); };
  std::vector<std::vector<uint32_t>> gcnt((size_t)GT);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < GT; t++)
      th.emplace_back([&, t]() {
        gcnt[(size_t)t].assign(ncell, 0);
        int p0, p1;
        gslice(t, p0, p1);
        uint32_t* const cn = gcnt[(size_t)t].data();
        for (int p = p0; p < p1; p++) {
          const int i = (int)tp.order[(size_t)p];
          if (!dead[(size_t)i]) for_cells(i, [&](size_t ci) { cn[ci]++; });
        }
      });
    for (auto& x : th) x.join();
  }
  size_t total_entries = 0;
  tp.cell_first.assign(ncell + 1, 0);
  for (size_t k = 0; k < ncell; k++) {
    tp.cell_first[k] = (uint32_t)total_entries;
    for (int t = 0; t < GT; t++) { const uint32_t c = gcnt[(size_t)t][k]; gcnt[(size_t)t][k] = (uint32_t)total_entries; total_entries += c; } // -> this thread's cursor in the cell
    if (total_entries >= (1u << 30)) return tp;
  }
  tp.cell_first[ncell] = (uint32_t)total_entries;
  tp.cell_cand.assign(total_entries, 0);
  {
    // Each entry also says in which of the cell's six face neighbours the triangle is listed as well (bits 26 ... 31: -x +x -y +y -z +z):
    // a walk steps from cell to cell through faces, and a triangle that the cell it comes from lists has been tested there already
    // (or where that cell's predecessor listed it, and so on back to the first cell of the chain) — a (ray, triangle) pair's test does
    // not depend on the cell it is made in, so the device skips it (tri_pool_scan).  2^26 positions: build_tri_pool's caller checks the count.
    const long long sx = 1, sy = tp.n[0], sz = (long long)tp.n[0] * tp.n[1];
    std::vector<std::thread> th;
    for (int t = 0; t < GT; t++)
      th.emplace_back([&, t]() {
        int p0, p1;
        gslice(t, p0, p1);
        uint32_t* const cur = gcnt[(size_t)t].data();
        std::vector<uint32_t> own; // the cells of one triangle, sorted
        for (int p = p0; p < p1; p++) { // in Morton order: a cell's candidates ascend
          const int i = (int)tp.order[(size_t)p];
          if (dead[(size_t)i]) continue;
          own.clear();
          for_cells(i, [&](size_t ci) { own.push_back((uint32_t)ci); });
          std::sort(own.begin(), own.end());
          for (uint32_t ci : own) {
            const int x = (int)(ci % (uint32_t)tp.n[0]), y = (int)((ci / (uint32_t)tp.n[0]) % (uint32_t)tp.n[1]), z = (int)(ci / (uint32_t)(tp.n[0] * tp.n[1]));
            auto has = [&](bool in_grid, long long c2) { return in_grid && std::binary_search(own.begin(), own.end(), (uint32_t)c2); };
            uint32_t bits = 0;
            bits |= has(x > 0, (long long)ci - sx) ? 1u : 0u;
            bits |= has(x + 1 < tp.n[0], (long long)ci + sx) ? 2u : 0u;
            bits |= has(y > 0, (long long)ci - sy) ? 4u : 0u;
            bits |= has(y + 1 < tp.n[1], (long long)ci + sy) ? 8u : 0u;
            bits |= has(z > 0, (long long)ci - sz) ? 16u : 0u;
            bits |= has(z + 1 < tp.n[2], (long long)ci + sz) ? 32u : 0u;
            tp.cell_cand[cur[ci]++] = (uint32_t)p | (bits << 26);
          }
        }
      });
    for (auto& x : th) x.join();
  }
  tp.mean_cells_per_triangle = (double)total_entries / std::max(1, n_live);
  lap("grid cells (count + fill)");
  // ---- compressed records --------------------------------------------------------------------------------------------------
  // Both band filters are NECESSARY conditions of an acceptance: any relaxation keeps the pool exact.  So the device reads them
  // from quantised records, every quantity rounded to the safe side:
  //   centroid  C~ = cq_lo + k cq_step, k a 16-bit integer per axis; |C~ - C| <= eps_c (measured below on the device's own
  //             binary32 decode) is added to every radius;
  //   L / pn  as bfloat16 rounded UP (relative 2^-7);
  //   unit normal  n~ = (kx, ky, kz) / 32767, |n~ - N/|N|| <= eps_n (measured): |d . n~| <= |d . N|/|N| + |d| eps_n.
  // Band test in normalised form: |d . N'| <= |d| (rho P + Q) <=> |d . N/|N|| <= |d| (rho pn + qn), pn = P/|N|, qn = Q/|N|, and
  // with 1/|N| = pn / (kP L):  qn = pn (kQ l1 l2 / (kP L) + 2^-40 / (kP L)) <= pn (KQ L + KT / L)   (l1 l2 <= L^2) — so the
  // record needs pn and L only.  The noise radius needs |a'| = |d . N'| >= (|d . n~| - |d| eps_n) |N| and |N| >= 0.98 kP L~/pn~.
  // A triangle without a normal (|N| = 0: parallel edges) gets the record that passes every filter: n~ = 0, pn~ = the largest bfloat16.
  {
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < count; i++) {
      if (dead[(size_t)i]) continue;
      for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], (double)tp.ball[(size_t)i * 4 + k]); chi[k] = std::max(chi[k], (double)tp.ball[(size_t)i * 4 + k]); }
    }
    for (int k = 0; k < 3; k++) { tp.cq_lo[k] = (float)clo[k]; tp.cq_step[k] = (float)((chi[k] - clo[k]) / 65535.0); }
    auto bf16_up = [](float x) -> uint32_t { // smallest bfloat16 >= x (x >= 0, finite)
      uint32_t b;
      std::memcpy(&b, &x, 4);
      if (b & 0xffffu) b += 0x10000u;
      return b >> 16;
    };
    auto bf16_val = [](uint32_t hh) { const uint32_t b = hh << 16; float f; std::memcpy(&f, &b, 4); return f; };
    std::vector<uint32_t> cq((size_t)count * 3, 0);
    double dev_c = 0, dev_n = 0;
    for (int i = 0; i < count; i++) {
      if (dead[(size_t)i]) continue;
      double d2 = 0;
      for (int k = 0; k < 3; k++) {
        const double C = tp.ball[(size_t)i * 4 + k];
        long q = tp.cq_step[k] > 0 ? std::lrint((C - (double)tp.cq_lo[k]) / (double)tp.cq_step[k]) : 0;
        q = std::max(0l, std::min(65535l, q));
        cq[(size_t)i * 3 + k] = (uint32_t)q;
        const float prod = (float)q * tp.cq_step[k]; // the device's decode, operation by operation (no contraction)
        const float dec = tp.cq_lo[k] + prod;
        d2 += ((double)dec - C) * ((double)dec - C);
      }
      dev_c = std::max(dev_c, std::sqrt(d2));
    }
    tp.eps_c = (float)(dev_c * (1 + 1e-6) + 1e-37);
    tp.kq = (float)((Ma * 7.0 + 4.0) / (M * 17.5) * 1.001);
    tp.kt = (float)(std::ldexp(1.0, -40) / (M * 17.5 * u * SAFE) * 1.02);
    tp.band_q.assign((size_t)count * 4, 0);
    for (int i = 0; i < count; i++) {
      if (dead[(size_t)i]) continue; // (an all-zero record: pn~ = L~ = 0 makes the band test compare against a NaN, which fails)
      const uint32_t* c3 = &cq[(size_t)i * 3];
      const uint32_t Lh = bf16_up(tp.ball[(size_t)i * 4 + 3]);
      tp.band_q[(size_t)i * 4 + 2] = c3[0] | (c3[1] << 16);
      tp.band_q[(size_t)i * 4 + 3] = c3[2] | (Lh << 16);
      if (!(pn[(size_t)i] < 1e30)) { tp.band_q[(size_t)i * 4 + 1] = 0x7f7fu << 16; continue; } // no (usable) normal: passes every filter
      int32_t nq[3];
      double d2 = 0;
      for (int k = 0; k < 3; k++) {
        nq[k] = (int32_t)std::lrint(nrm[(size_t)i * 3 + k] * 32767.0);
        d2 += (nq[k] / 32767.0 - nrm[(size_t)i * 3 + k]) * (nq[k] / 32767.0 - nrm[(size_t)i * 3 + k]);
      }
      dev_n = std::max(dev_n, std::sqrt(d2));
      const uint32_t pnh = bf16_up((float)(pn[(size_t)i] * (1 + 1e-6)));
      tp.band_q[(size_t)i * 4] = ((uint32_t)nq[0] & 0xffffu) | ((uint32_t)nq[1] << 16);
      tp.band_q[(size_t)i * 4 + 1] = ((uint32_t)nq[2] & 0xffffu) | (pnh << 16);
      // the closed form really bounds this triangle's qn (and pn~, L~ are finite): otherwise no pool
      const double pnv = bf16_val(pnh), Lv = bf16_val(Lh);
      if (!(pnv < 1e30 && Lv < 1e30 && qn[(size_t)i] * (1 + 8 * u) <= pnv * ((double)tp.kq * Lv + (double)tp.kt / Lv))) return tp;
    }
    tp.eps_n = (float)(dev_n * (1 + 1e-6) + 3e-6); // + the binary32 rounding of d . (kx, ky, kz) / 32767 and of N'/|N'| against N/|N|
    // ---- ready records -------------------------------------------------------------------------------------------------------
    // The filter of a compressed record (pt_device.hpp: tri_pool_scan, band_pass), on the record's decoded values n~ = (kx, ky, kz) / 32767,
    // pn~, L~, C~ (the device's own decode, operation by operation):
    //     band    |d . n~| <= |d| (pn~ (rho + kq L~ + kt / L~) + eps_n)
    //     radius  a1 = (|d . n~| - |d| eps_n) nlow - ea L~^2 |d|,  nlow = 0.98 p_per_L L~ / pn~;  a1 <= 0, or the ray's line within
    //             L~ + ball_abs + eps_c + (kr_a + kr_b L~) L~^2 rho |d| / a1 of C~
    // The binned band stage evaluates the same two conditions from G = pn~ (kq L~ + kt / L~) + eps_n, nlow, E2 = ea L~^2,
    // KR = (kr_a + kr_b L~) L~^2 and Lr = L~ + ball_abs + eps_c, computed HERE in binary64 and rounded to binary32 on the side that makes
    // the condition easier to pass (G, E2, KR, Lr up, nlow down; a relative 4e-6 on top, and the device keeps band_pass's own factors
    // 1.00001 / 1.001): every pair band_pass lets through, the ready record lets through — a necessary condition stays one.
    {
      auto up = [](double x) { return std::nextafter((float)(x * (1 + 4e-6)), INFINITY); };
      auto down = [](double x) { return std::max(0.0f, std::nextafter((float)(x * (1 - 4e-6)), -INFINITY)); };
      tp.band_ready.assign((size_t)count * 12, 0.0f);
      for (int i = 0; i < count; i++) {
        float* r = &tp.band_ready[(size_t)i * 12];
        if (dead[(size_t)i]) { r[8] = -1.0f; continue; } // n~ = 0 and G < 0: |d . n~| = 0 <= |d| (0 rho - 1) never holds
        const uint32_t* q = &tp.band_q[(size_t)i * 4];
        const int16_t kx = (int16_t)(q[0] & 0xffffu), ky = (int16_t)(q[0] >> 16), kz = (int16_t)(q[1] & 0xffffu);
        const double pnv = bf16_val(q[1] >> 16), Lv = bf16_val(q[3] >> 16);
        const uint32_t ck[3] = {q[2] & 0xffffu, q[2] >> 16, q[3] & 0xffffu};
        r[0] = (float)kx; r[1] = (float)ky; r[2] = (float)kz; r[3] = (float)pnv;
        for (int k = 0; k < 3; k++) { const float prod = (float)ck[k] * tp.cq_step[k]; r[4 + k] = tp.cq_lo[k] + prod; } // the device's decode (tri_centroid)
        r[7] = up(Lv + (double)tp.ball_abs + (double)tp.eps_c);
        const double g = pnv * ((double)tp.kq * Lv + (double)tp.kt / Lv * 1.00002) + (double)tp.eps_n;
        r[8] = g < 3e38 ? up(g) : INFINITY;
        r[9] = down(0.98 * (double)tp.p_per_L * Lv / pnv);
        r[10] = up((double)tp.ea * Lv * Lv);
        r[11] = up(((double)tp.kr_a + (double)tp.kr_b * Lv) * Lv * Lv);
      }
    }
  }
  lap("compressed + ready records");
  // ---- (2) the direction maps ------------------------------------------------------------------------------------------------
  // class k: rays with rho <= rho_max_k; triangle i is listed by tau_i = rho_max_k pn_i + qn_i (what the real band test admits for such a ray)
  {
    int T = tune.threads > 0 ? tune.threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(T, 16));
    long long budget = tune.dm_budget;
    std::vector<double> tau((size_t)count, 0.0);
    for (int k = 0; k < 3; k++) {
      if (!(tune.dm_rho[k] > 0.0f) || tune.dm_res[k] < 4) continue;
      const double rho_max = (double)tune.dm_rho[k] * R;
      if (!tp.maps.empty() && !(rho_max > tp.maps.back().rho_max)) continue;
      int wide = 0;
      for (int i = 0; i < count; i++) {
        if (dead[(size_t)i]) continue;
        tau[(size_t)i] = pn[(size_t)i] < 1e30 ? (rho_max * pn[(size_t)i] + qn[(size_t)i]) * (1 + 1e-6) + 1e-9 : INFINITY;
        if (!(tau[(size_t)i] < 1.0)) wide++;
      }
      if (tp.maps.empty()) tp.wide = wide;
      for (int res = std::min(1024, tune.dm_res[k]); res >= 16; res /= 2) { // a map over budget is tried at half the resolution
        if (count >= 65536) { // (a cheap estimate first — every 32nd triangle — so that a hopeless resolution does not cost a full counting pass)
          std::vector<uint32_t> sample;
          for (int p = 0; p < count; p += 32) sample.push_back(tp.order[(size_t)p]);
          TriDirMap probe;
          if (!build_dir_map(probe, res, nrm, tau, dead, (int)sample.size(), sample, budget / 24, 1)) continue;
        }
        TriDirMap dm;
        dm.rho_max = (float)(rho_max * (1 - 1e-6));
        if (build_dir_map(dm, res, nrm, tau, dead, count, tp.order, budget, T)) { budget -= (long long)dm.cand.size(); tp.maps.push_back(std::move(dm)); break; }
      }
    }
  }
  lap("direction maps");
  tp.ok = true;
  return tp;
}

} // namespace ptf
