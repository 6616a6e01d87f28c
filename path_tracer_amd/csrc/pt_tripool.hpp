// pt_tripool.hpp — host side of the EXACT culling structure for long runs of Moller-Trumbore triangles (SURVEY.md §8f-4:
// "order-preserving culling for large N", config 5: 100 k triangles).  Pure host C++; the device query that reads these
// tables is tri_pool_scan in pt_device.hpp.
//
// The reference scans every triangle for every ray (render.hpp:37-49 -> triangle.hpp:58-100).  A spatial structure alone
// cannot reproduce that scan bit for bit: the binary32 test accepts a triangle whenever its COMPUTED u, v, u + v pass the
// comparisons against the COMPUTED a, and for a ray that grazes the triangle's plane (|a| barely above the 1e-7 cut-off of
// triangle.hpp:71) the rounding noise in u and v exceeds |a| — such a triangle can be "hit" although the ray's line passes
// far from it.  So the candidate set of a ray is built from THREE exact parts, and a triangle the reference could accept is
// always in at least one of them:
//
//   (1) GRID.   Pairs (ray, triangle) that are NOT grazing, |a^| >= thr_i (a^ = the exact e1 . (d x e2) = -d . N_i).  For those
//       the computed barycentrics are within 1/M of the exact ones, so the exact point P^ where the ray's line meets the
//       triangle's plane lies within sigma_i of the triangle, and the computed t is within the same distance (along the ray,
//       plus a relative 1/M_a) of P^'s exact parameter.  The triangle is listed in every cell of a uniform grid that its
//       bounding box, grown by sigma'_i, touches; the ray walks the cells of its segment [0, closest (1 + kappa)].
//   (2) BAND.   Pairs that ARE grazing, |d^ . n^_i| < tau_i(rho): the unit normals of the triangles with a narrow band are
//       bucketed on a cube map (three faces, antipodes identified); the set { n : |d^ . n| <= tau } is a great-circle strip,
//       which central projection turns into a STRAIGHT strip on each face: rasterised per ray, per face, row by row.
//   (3) ALWAYS. Triangles whose band is too wide for a map (slivers: tau_i ~ 1 / sin(angle between the edges)) are kept in
//       a plain list that every ray scans — through the same cheap band test, so that only the grazing ones are tested.
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
//     |beta - u/a| <= (du + da) / |a^| <= 1/M + 1/M_a, same for gamma: beta, gamma >= -1/M', beta + gamma <= 1 + 2/M' + u
//     => dist(P^, triangle) <= (6/M + 6/M_a + u) L_i
//     |t - t^| |d| <= |d| dw / |a| + |t^| |d| da / |a| + u |t| |d| <= 1.2 L_i / (M - 1) + |t^| |d| / (M_a - 1) + u |t| |d|
// so with kappa = 2.2 / (M_a - 1) the exact parameter t^ lies in [-1.2 L_i / ((M-1) |d|), max (1 + kappa) + 1.2 L_i / ((M-1) |d|)]
// and the point P' = o + clamp(t^, 0, max (1 + kappa)) d of the WALKED segment is within
//     sigma'_i = (6/M + 6/M_a + 1.2/(M-1)) L_i  <=  8.5 L_i / (M - 1)        (M >= 8, M_a >= 64)
// of the triangle, hence inside its bounding box grown by sigma'_i: the cell that contains P' lists the triangle, and the walk
// visits that cell (cells are assigned with a further absolute slack for the walk's own rounding, as the sphere grid's).
// Grazing, |a^| < thr_i, is |d^ . N_i| < rho P_i + Q_i: the band test, evaluated with N'_i = N_i rounded to binary32.
// tests/test_tripool_cpu.py checks these inequalities on float32 emulations of the reference's test (random and adversarial
// grazing rays); the GPU suite checks the walked structure against the oracle's brute-force scan, bit for bit.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pt_render.h"

namespace ptf {

struct TriPoolTuning {
  float M = 12.0f;      // PT_TRI_M (swept 8 ... 32 on cfg5: profiles/r03_tripool_sweep*.log): barycentric slack 1/M; the band width grows with M, the boxes' growth sigma' with 1/M
  float Mg = 96.0f;     // PT_TRI_MG: the grid's TIGHT slack (pairs with |a^| >= thr(Mg) are found within sigma'(Mg) of the triangle; the others of the grid's share, thr(M) <= |a^| < thr(Mg), pass a band test at Mg: see "compressed records")
  float Ma = 256.0f;    // relative slack of t: the walk runs to max (1 + 2.2 / (Ma - 1))
  float cell = 0.7f;    // PT_TRI_CELL (swept 0.35 ... 3.0; 0.5 / 0.7 / 1.0 / 1.4: 2.51 / 2.50 / 2.53 / 2.56 s at 1080p x 32 spp): grid cell edge in units of the median grown box extent
  int res[3] = {128, 64, 32}; // PT_TRI_RES=a,b,c (128,32,16 / 128,64,32 / 128,128,64: 2.53 / 2.41 / 2.40 s at 1080p x 32 spp): cube-map resolution of the three band levels (powers of two <= 128: the device deals a level's rows to the 64 lanes)
  int min_run = 4096;   // PT_TRI_MIN: shorter triangle runs are scanned as before (PT_TRICULL=1: 256)
};

struct TriPoolLevel {
  int R = 0;
  float pn_max = 0.0f, qn_max = 0.0f; // the strip of a ray is |A p + B q + C| <= sqrt(3) (rho pn_max + qn_max)
  // Two copies of the map, one per way a ray can walk a face: orientation 0 has the cells of a q-row (fixed cj) contiguous in
  // ci, orientation 1 the cells of a p-column (fixed ci) contiguous in cj — so that the cells a ray's strip covers in one row
  // are ONE contiguous candidate range either way.  first: 3 R R + 1 prefix offsets; cand: triangle index in the run.
  // (The band record of every candidate rides inline beside its index, in candidate order: see put_tri_pool.)
  std::vector<uint32_t> first[2];
  std::vector<uint32_t> cand[2];
};

struct TriPool {
  bool ok = false;
  float origin[3] = {0, 0, 0}, inv_cell = 0, cell = 0;
  int n[3] = {1, 1, 1};
  float centre[3] = {0, 0, 0}, R = 0, rlimit2 = 0, kappa = 0;
  std::vector<uint32_t> cell_first; // n cells + 1
  std::vector<uint32_t> cell_cand;  // triangle index in the run
  std::vector<TriPoolLevel> levels;
  std::vector<uint32_t> always;
  std::vector<float> cheap;         // 4 floats per triangle: g = N' / P, c = Q / P      band test: |d . g| < |d| (rho + c)
  std::vector<float> ball;          // 4 floats per triangle: centroid C, L = longest stored edge (every vertex is within L of C)
  std::vector<float> grid_radius;   // per triangle: Rv + sigma' + ball_abs, Rv = the largest distance of a vertex from C (the grid filter's radius)
  float p_per_L = 0, k_sigma = 0, ball_abs = 0, kr_a = 0, kr_b = 0, ea = 0; // constants of the two distance filters (see build_tri_pool)
  // COMPRESSED filter records (what the device streams: the filters are necessary conditions, so any relaxation of them is
  // still exact — see "compressed records" in build_tri_pool): per triangle 2 dwords for the grid, 4 for the band
  float cq_lo[3] = {0, 0, 0}, cq_step[3] = {0, 0, 0}, eps_c = 0, eps_n = 0, kq = 0, kt = 0, k_loose = 0, m_scale = 0;
  std::vector<uint32_t> grid_q; // (cq.x | cq.y << 16) (cq.z | bf16(tight radius) << 16)
  std::vector<uint32_t> grid_n; // (nq.x | nq.y << 16) (nq.z | bf16(pn_eff) << 16)
  std::vector<uint32_t> band_q; // (nq.x | nq.y << 16) (nq.z | bf16(pn) << 16) (cq.x | cq.y << 16) (cq.z | bf16(L) << 16)
  // statistics for the tests / DESIGN
  double mean_cells_per_triangle = 0;
};

inline TriPool build_tri_pool(const PtHittable* h, int count, TriPoolTuning tune = TriPoolTuning()) {
  TriPool tp;
  if (count < tune.min_run) return tp;
  const double u = std::ldexp(1.0, -24), SAFE = 1.5;
  const double M = std::max(8.0, (double)tune.M), Ma = std::max(64.0, (double)tune.Ma);
  std::vector<double> P((size_t)count), Q((size_t)count), pn((size_t)count), qn((size_t)count), sig((size_t)count);
  std::vector<double> nrm((size_t)count * 3);
  std::vector<char> dead((size_t)count, 0);
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
  tp.cheap.assign((size_t)count * 4, 0.0f);
  tp.ball.assign((size_t)count * 4, 0.0f);
  std::vector<double> rv_of; // per LIVE triangle, in order
  std::vector<int> live_index((size_t)count, -1);
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
    live_index[(size_t)i] = (int)rv_of.size();
    P[(size_t)i] = M * 17.5 * u * L * SAFE;
    Q[(size_t)i] = (Ma * 7.0 + 4.0) * u * l1 * l2 * SAFE + std::ldexp(1.0, -40);
    sig[(size_t)i] = 8.5 * L / (M - 1.0);
    const float Np[3] = {(float)N[0], (float)N[1], (float)N[2]}; // N' = N rounded to binary32 (the 4 u of Q_i)
    for (int k = 0; k < 3; k++) tp.cheap[(size_t)i * 4 + k] = (float)((double)Np[k] / P[(size_t)i]);
    tp.cheap[(size_t)i * 4 + 3] = (float)(Q[(size_t)i] / P[(size_t)i] * (1.0 + 4 * u));
    if (nN > 0.0) {
      pn[(size_t)i] = P[(size_t)i] / nN; qn[(size_t)i] = Q[(size_t)i] / nN;
      for (int k = 0; k < 3; k++) nrm[(size_t)i * 3 + k] = N[k] / nN;
    } else { pn[(size_t)i] = qn[(size_t)i] = INFINITY; }
    for (int k = 0; k < 3; k++) tp.ball[(size_t)i * 4 + k] = (float)(f[k] + (e1[k] + e2[k]) / 3.0); // centroid of v0, v0 + e1, v0 + e2
    tp.ball[(size_t)i * 4 + 3] = (float)(L * (1 + 2 * u));
    {
      double rv = 0;
      const double cx[3] = {tp.ball[(size_t)i * 4], tp.ball[(size_t)i * 4 + 1], tp.ball[(size_t)i * 4 + 2]}; // the ROUNDED centroid the device uses
      for (int v = 0; v < 3; v++) {
        double d2 = 0;
        for (int k = 0; k < 3; k++) { const double pv = (double)f[k] + (v == 1 ? e1[k] : v == 2 ? e2[k] : 0.0); d2 += (pv - cx[k]) * (pv - cx[k]); }
        rv = std::max(rv, std::sqrt(d2));
      }
      rv_of.push_back(rv);
    }
    double emax = 0;
    for (int k = 0; k < 3; k++) {
      const double lo = std::min({(double)f[k], (double)f[3 + k], (double)f[6 + k]}), hi = std::max({(double)f[k], (double)f[3 + k], (double)f[6 + k]});
      emax = std::max(emax, hi - lo + 2 * sig[(size_t)i]);
    }
    ext.push_back(emax);
  }
  if (ext.size() < (size_t)tune.min_run) return tp;
  R *= 1.0 + 8 * u;
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
  for (;;) { // at most 2^21 cells, at most 256 per axis
    double total = 1;
    bool fits = true;
    for (int k = 0; k < 3; k++) {
      const double nk = std::max(1.0, std::ceil((hi[k] - lo[k] + 4e-3 * cell) / cell));
      if (nk > 256) fits = false;
      tp.n[k] = (int)std::min(nk, 256.0);
      total *= nk;
    }
    // ... and a bounded candidate table: every (cell, triangle) entry carries the triangle's three records inline, and a few
    // triangles that span the whole grid (cells are sized for the median one) would otherwise list themselves in every cell
    double entries = 0;
    if (fits && total <= 2097152.0) {
      for (int i = 0; i < count && entries <= 1e9; i++) {
        if (dead[(size_t)i]) continue;
        double blo[3], bhi[3], e = 1;
        box_of(i, 0.0, blo, bhi);
        for (int k = 0; k < 3; k++) e *= std::min((double)tp.n[k], (bhi[k] - blo[k]) / cell + 2.0);
        entries += e;
      }
      if (entries <= std::max(24.0 * (double)ext.size(), 65536.0)) break;
    }
    cell *= 1.25;
  }
  const double slack = 1e-3 * cell; // the walk's own rounding (the ray must start within rlimit: below)
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
  // ---- two distance filters in front of the exact test (both necessary conditions of an acceptance, so neither can lose one)
  // (i) grid candidates (pairs that are not grazing): P' — a point of the ray's line — is within sigma'_i of the triangle, and
  //     every point of the triangle is within L_i of its centroid C_i: the line passes within L_i (1 + 8.5 / (M - 1)) of C_i.
  // (ii) band candidates: whatever |a^| is, an accepted pair has |beta - u/a| <= (du + da) / |a^| (same for gamma), so P^ — on the
  //     ray's line — lies within r = 6 (du + dv + da) L_i / |a^| of the triangle: the line passes within L_i + r of C_i, with
  //     r <= kr(L_i) rho |d| / (|a'| - ea(L_i) |d|),  kr = 6 SAFE u L^2 (17.5 + 7 L / R),  ea = 4 u L^2  (|a' - a^| <= ea |d|;
  //     rho >= R lets the da term ride on rho).  The device evaluates |(C - o) x d|^2 <= radius^2 |d|^2 in binary32: its
  //     rounding (and the centroid's) is covered by ball_abs = 64 u (rlimit + R + diagonal) added to every radius.
  tp.p_per_L = (float)(M * 17.5 * u * SAFE);
  tp.k_sigma = (float)((1.0 + 8.5 / (M - 1.0)) * (1 + 8 * u));
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
    // (i) tightened: every point of the triangle is within Rv_i (its farthest vertex) of the centroid, so the line passes within
    // Rv_i + sigma'_i of it
    tp.grid_radius.assign((size_t)count, 0.0f);
    for (int i = 0; i < count; i++)
      if (!dead[(size_t)i]) tp.grid_radius[(size_t)i] = (float)((rv_of[(size_t)live_index[(size_t)i]] + sig[(size_t)i]) * (1 + 8 * u) + tp.ball_abs);
  }
  const double inv = (double)tp.inv_cell; // assign with the float value the device uses
  const size_t ncell = (size_t)tp.n[0] * tp.n[1] * tp.n[2];
  std::vector<uint32_t> cnt(ncell + 1, 0);
  auto cells_of = [&](int i, int c0[3], int c1[3]) {
    double blo[3], bhi[3];
    box_of(i, slack, blo, bhi);
    for (int k = 0; k < 3; k++) {
      c0[k] = std::max(0, std::min(tp.n[k] - 1, (int)std::floor((blo[k] - tp.origin[k]) * inv)));
      c1[k] = std::max(0, std::min(tp.n[k] - 1, (int)std::floor((bhi[k] - tp.origin[k]) * inv)));
    }
  };
  size_t total_entries = 0;
  for (int i = 0; i < count; i++) {
    if (dead[(size_t)i]) continue;
    int c0[3], c1[3];
    cells_of(i, c0, c1);
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) { cnt[((size_t)z * tp.n[1] + y) * tp.n[0] + x]++; total_entries++; }
  }
  if (total_entries >= (1u << 28)) return tp;
  tp.cell_first.assign(ncell + 1, 0);
  for (size_t k = 0; k < ncell; k++) tp.cell_first[k + 1] = tp.cell_first[k] + cnt[k];
  tp.cell_cand.assign(total_entries, 0);
  std::vector<uint32_t> cur(tp.cell_first.begin(), tp.cell_first.end() - 1);
  for (int i = 0; i < count; i++) { // in list order: a cell's candidates ascend
    if (dead[(size_t)i]) continue;
    int c0[3], c1[3];
    cells_of(i, c0, c1);
    for (int z = c0[2]; z <= c1[2]; z++)
      for (int y = c0[1]; y <= c1[1]; y++)
        for (int x = c0[0]; x <= c1[0]; x++) tp.cell_cand[cur[((size_t)z * tp.n[1] + y) * tp.n[0] + x]++] = (uint32_t)i;
  }
  tp.mean_cells_per_triangle = (double)total_entries / std::max<size_t>(1, ext.size());
  // ---- (2) band levels on the cube map of normals, (3) the always list -----------------------------------------------------
  // level k takes the triangles whose band half-width at the reference distance, tau_i = rho_ref pn_i + qn_i, is <= tau_k
  const double rho_ref = 3.0 * R;
  const double tau_cap[3] = {0.004, 0.016, 0.064};
  int res[3];
  for (int k = 0; k < 3; k++) { res[k] = 16; while (res[k] < tune.res[k] && res[k] < 128) res[k] *= 2; } // powers of two in [16, 128]
  tp.levels.resize(3);
  std::vector<int> level_of((size_t)count, -1);
  for (int i = 0; i < count; i++) {
    if (dead[(size_t)i]) continue;
    const double tau = rho_ref * pn[(size_t)i] + qn[(size_t)i];
    int lv = 3;
    for (int k = 0; k < 3; k++) if (tau <= tau_cap[k]) { lv = k; break; }
    if (lv == 3) { tp.always.push_back((uint32_t)i); continue; }
    level_of[(size_t)i] = lv;
    TriPoolLevel& L = tp.levels[(size_t)lv];
    L.pn_max = std::max(L.pn_max, (float)(pn[(size_t)i] * (1 + 4 * u)));
    L.qn_max = std::max(L.qn_max, (float)(qn[(size_t)i] * (1 + 4 * u)));
  }
  for (int lv = 0; lv < 3; lv++) {
    TriPoolLevel& L = tp.levels[(size_t)lv];
    L.R = res[lv];
    const size_t nc = (size_t)3 * L.R * L.R;
    auto map_cell = [&](int i, int orient) -> size_t {
      const double* nn = &nrm[(size_t)i * 3];
      int k = 0;
      if (std::fabs(nn[1]) > std::fabs(nn[k])) k = 1;
      if (std::fabs(nn[2]) > std::fabs(nn[k])) k = 2;
      const int a = (k + 1) % 3, b = (k + 2) % 3; // face k: (p, q) = (n_a, n_b) / n_k
      const double p = nn[a] / nn[k], q = nn[b] / nn[k];
      const int ci = std::max(0, std::min(L.R - 1, (int)std::floor((p + 1.0) * 0.5 * L.R)));
      const int cj = std::max(0, std::min(L.R - 1, (int)std::floor((q + 1.0) * 0.5 * L.R)));
      return orient == 0 ? ((size_t)k * L.R + cj) * L.R + ci : ((size_t)k * L.R + ci) * L.R + cj;
    };
    for (int orient = 0; orient < 2; orient++) {
      std::vector<uint32_t> cn(nc + 1, 0);
      for (int i = 0; i < count; i++) if (level_of[(size_t)i] == lv) cn[map_cell(i, orient)]++;
      L.first[orient].assign(nc + 1, 0);
      for (size_t k = 0; k < nc; k++) L.first[orient][k + 1] = L.first[orient][k] + cn[k];
      L.cand[orient].assign(L.first[orient][nc], 0);
      std::vector<uint32_t> cu(L.first[orient].begin(), L.first[orient].end() - 1);
      for (int i = 0; i < count; i++) if (level_of[(size_t)i] == lv) L.cand[orient][cu[map_cell(i, orient)]++] = (uint32_t)i;
    }
  }
  // ---- compressed records --------------------------------------------------------------------------------------------------
  // The scan is bound by the bytes it streams, and both filters are NECESSARY conditions of an acceptance: any
  // relaxation keeps the pool exact.  So the device reads them from quantised records, every quantity rounded to the safe side:
  //   centroid  C~ = cq_lo + k cq_step, k a 16-bit integer per axis; |C~ - C| <= eps_c (measured below on the device's own
  //             binary32 decode) is added to every radius;
  //   radius / L / pn  as bfloat16 rounded UP (relative 2^-7);
  //   unit normal  n~ = (kx, ky, kz) / 32767, |n~ - N/|N|| <= eps_n (measured): |d . n~| <= |d . N|/|N| + |d| eps_n.
  // The GRID's filter has two radii.  The bound of the header holds for any M >= 8: a pair with |a^| >= thr(Mg) (Mg = 96 >> M) has
  // its P' within sigma'(Mg) = (6/Mg + 6/Ma + 1.2/(Mg-1)) L of the triangle — the TIGHT radius Rv + sigma'(Mg), stored per candidate;
  // the rest of the grid's share, thr(M) <= |a^| < thr(Mg), lies within the LOOSE radius Rv + sigma'(M) <= tight (1 + 2 (8.5/(M-1) -
  // sigma'(Mg)/L)) (every edge is <= 2 Rv), and satisfies the band test at Mg:  |d . N/|N|| < |d| (rho pn Mg/M + qn)  — evaluated on
  // n~ with pn_eff = pn (1 + KT / (L R)) >= pn + (the 2^-40 term of qn) / rho  (rho >= R) and L <= 2 tight.  A candidate is tested
  // exactly when  within(tight) or (within(loose) and band(Mg)).  The cells list a triangle by its box grown by sigma'(M), as before.
  // Band test in normalised form: |d . N'| <= |d| (rho P + Q) <=> |d . N/|N|| <= |d| (rho pn + qn), pn = P/|N|, qn = Q/|N|, and
  // with 1/|N| = pn / (kP L):  qn = pn (kQ l1 l2 / (kP L) + 2^-40 / (kP L)) <= pn (KQ L + KT / L)   (l1 l2 <= L^2) — so the
  // record needs pn and L only.  The noise radius needs |a'| = |d . N'| >= (|d . n~| - |d| eps_n) |N| and |N| >= 0.98 kP L~/pn~.
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
    auto bf16_val = [](uint32_t h) { const uint32_t b = h << 16; float f; std::memcpy(&f, &b, 4); return f; };
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
    const double Mg = std::max(M, (double)tune.Mg);
    const double sig_g = (6.0 / Mg + 6.0 / Ma + 1.2 / (Mg - 1.0)) * (1 + 8 * u); // sigma'(Mg) / L
    tp.k_loose = (float)((1.0 + 2.0 * std::max(0.0, 8.5 / (M - 1.0) - sig_g)) * (1 + 1e-6));
    tp.m_scale = (float)(Mg / M * (1 + 1e-6));
    tp.grid_q.assign((size_t)count * 2, 0);
    tp.grid_n.assign((size_t)count * 2, 0);
    tp.band_q.assign((size_t)count * 4, 0);
    std::vector<double> nq_dev((size_t)count, 0.0);
    for (int i = 0; i < count; i++) {
      if (dead[(size_t)i]) continue;
      const uint32_t* c3 = &cq[(size_t)i * 3];
      const double Ld = tp.ball[(size_t)i * 4 + 3];
      const float rg = (float)((rv_of[(size_t)live_index[(size_t)i]] + sig_g * Ld) * (1 + 8 * u) + tp.ball_abs + tp.eps_c);
      tp.grid_q[(size_t)i * 2] = c3[0] | (c3[1] << 16);
      tp.grid_q[(size_t)i * 2 + 1] = c3[2] | (bf16_up(rg * (1 + 2e-7f)) << 16);
      tp.grid_n[(size_t)i * 2 + 1] = 0x7f7fu << 16; // (|N| = 0: no direction — the band test at Mg always passes)
      if (!(pn[(size_t)i] < 1e30)) continue; // (|N| = 0: always list, exact records)
      int32_t nq[3];
      double d2 = 0;
      for (int k = 0; k < 3; k++) {
        nq[k] = (int32_t)std::lrint(nrm[(size_t)i * 3 + k] * 32767.0);
        d2 += (nq[k] / 32767.0 - nrm[(size_t)i * 3 + k]) * (nq[k] / 32767.0 - nrm[(size_t)i * 3 + k]);
      }
      dev_n = std::max(dev_n, std::sqrt(d2));
      const uint32_t pnh = bf16_up((float)(pn[(size_t)i] * (1 + 1e-6))), Lh = bf16_up(tp.ball[(size_t)i * 4 + 3]);
      {
        const double pe = pn[(size_t)i] * (1.0 + (double)tp.kt / (Ld * std::max(R, 1e-30))) * (1 + 1e-6);
        tp.grid_n[(size_t)i * 2] = ((uint32_t)nq[0] & 0xffffu) | ((uint32_t)nq[1] << 16);
        tp.grid_n[(size_t)i * 2 + 1] = ((uint32_t)nq[2] & 0xffffu) | ((pe < 3e38 ? bf16_up((float)pe) : 0x7f7fu) << 16);
      }
      tp.band_q[(size_t)i * 4] = ((uint32_t)nq[0] & 0xffffu) | ((uint32_t)nq[1] << 16);
      tp.band_q[(size_t)i * 4 + 1] = ((uint32_t)nq[2] & 0xffffu) | (pnh << 16);
      tp.band_q[(size_t)i * 4 + 2] = c3[0] | (c3[1] << 16);
      tp.band_q[(size_t)i * 4 + 3] = c3[2] | (Lh << 16);
      // the closed form really bounds this triangle's qn (and pn~, L~ are finite): otherwise no pool
      const double pnv = bf16_val(pnh), Lv = bf16_val(Lh);
      if (!(pnv < 1e30 && Lv < 1e30 && qn[(size_t)i] * (1 + 8 * u) <= pnv * ((double)tp.kq * Lv + (double)tp.kt / Lv))) return tp;
    }
    tp.eps_n = (float)(dev_n * (1 + 1e-6) + 3e-6); // + the binary32 rounding of d . (kx, ky, kz) / 32767 and of N'/|N'| against N/|N|
  }
  tp.ok = true;
  return tp;
}

} // namespace ptf
