// pt_device.hpp — device functions of the render() hot path for gfx950 (CDNA4).
//
// What is computed follows the reference line by line (citations below are into
// /root/reference/include); how it is computed does not:
//
//  * the std::variant<hittable> list (render.hpp:22-23, 624 B per element) is a
//    flattened blob of 16-byte records grouped in order-preserving RUNS of one
//    kind; the blob lives in LDS (or is fetched with scalar loads), the primitive
//    index is wave-uniform, so every primitive fetch is an LDS broadcast / SGPR
//    operand and every kind dispatch is a scalar branch (visit.hpp:51-67 becomes
//    s_cbranch);
//  * the traversal keeps only (closest t, hit id [, u, v]) per lane; hit_record
//    (hitable.hpp:8-24) is materialised once per ray for the final nearest hit —
//    a pure function of (ray, primitive, t), so deferring it cannot change a bit;
//  * uniform sub-expressions are hoisted or precomputed at flatten time
//    (dot(d,d), radius^2, triangle edges) — same IEEE operations on the same
//    operands, so same bits.
//
// Compiled with -ffp-contract=off: no fused multiply-add except the explicit
// sycl::fma of vec.hpp:12.  Division and sqrt are the correctly rounded forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_math.hpp"

#ifdef PT_STAMPS
extern __device__ unsigned long long g_stamps[8]; // diagnostic build only (pt_render.hip)
#endif
#ifdef PT_STAMPS_TRI
// diagnostic build (triangle pool): [0] scans (waves) [1] live rays [2] grid wave-steps [3] grid lane-steps (cells visited) [4] grid pairs (exact tests) [5] grid batches [6] direction-map entries enumerated [7] past the integer band test [8] past the noise radius (exact tests) [9] rays through the third map [12] through the second [10] rays that streamed every band record [11] band trips (128 entries)
extern __device__ unsigned long long g_tri[16];
#endif
#ifdef PT_STAMPS_RUNS
extern __device__ unsigned long long g_runs[16];
#endif
#ifdef PT_STAMPS_WALK
// diagnostic build: per-workgroup counters of the sphere-grid walk in LDS (cheap ds_add; global atomics per step distort the
// timing they are meant to explain), flushed once by render_kernel.  [0] cycles inside walks (wave leader's clock) [1] walks
// (waves) [2] wave-steps [3] split phases [4] lane-steps (cells visited by live walks) [5] lane sphere tests [6] wave test trips
extern __device__ unsigned long long g_walk[8];
__device__ __forceinline__ unsigned long long* walk_ctr() { __shared__ unsigned long long c[8]; return c; }
#define PT_WALK_COUNT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((threadIdx.x & 63) == 0) atomicAdd(&walk_ctr()[i], v_); } while (0)
#else
#define PT_WALK_COUNT(i, v) do { } while (0)
#endif

namespace ptd {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f4* lds_f4p; // LDS-resident blob
typedef const __attribute__((address_space(4))) f4* cst_f4p; // global blob via scalar (SMEM) loads
typedef const __attribute__((address_space(1))) f4* glb_f4p; // the same blob for per-lane (VMEM) loads: the triangle pool

// device kinds of a run / hit id (not the ABI tags: the three rect axes share one kind)
enum { DK_SPHERE = 0, DK_RECT = 1, DK_TRI = 2, DK_BOX = 3, DK_MEDIUM = 4, DK_TRI_B = 5 /* Badouel-strategy triangles */,
       DK_ABSORBED = 8 /* flag in a run header's kind: pt_flatten.hpp "absorbed sphere runs" */ };

// record sizes in f4 units
enum { SZ_SPHERE = 3, SZ_RECT = 2, SZ_TRI = 3, SZ_BOX = 2, SZ_MEDIUM = 4, SZ_MATERIAL = 4 };

// hit id: [24:0] record offset in the blob (f4 units: 33.5 M records, i.e. 11 M triangles), [27:25] box side, [30:28] device kind; -1 = none
// (pt_flatten.hpp: kHitOffBits / kHitSideShift / kHitKindShift are the same numbers)
#define PT_HIT_SIDE_SHIFT 25
#define PT_HIT_KIND_SHIFT 28
#define PT_HIT_OFF_MASK 0x1ffffff
__device__ __forceinline__ int hit_pack(int kind, int side, int off) { return (kind << PT_HIT_KIND_SHIFT) | (side << PT_HIT_SIDE_SHIFT) | off; }
__device__ __forceinline__ int hit_kind(int h) { return (h >> PT_HIT_KIND_SHIFT) & 7; }
__device__ __forceinline__ int hit_side(int h) { return (h >> PT_HIT_SIDE_SHIFT) & 7; }
__device__ __forceinline__ int hit_off(int h) { return h & PT_HIT_OFF_MASK; }

__device__ __forceinline__ float as_f(int i) { return __int_as_float(i); }
__device__ __forceinline__ int as_i(float f) { return __float_as_int(f); }

#ifndef PT_STRIDED_K
#define PT_STRIDED_K 4 /* spheres per trip of the cooperative (strided) scan: 4 or 2 */
#endif
#define PT_INF (__builtin_inff())
#define PT_PI 3.1415926535897932385f /* rtweekend.hpp:22 */
#define PT_TMIN 0.001f               /* render.hpp:40 */

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 xyz(f4 v) { return mk(v.x, v.y, v.z); }
// vec.hpp:11-13 (explicit fma)
__device__ __forceinline__ float length_squared(V3 v) {
  return __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z));
}
// Correctly rounded fp32 sqrt: __builtin_sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt (the HIP header's
// __fsqrt_rn is the 1-ulp native v_sqrt_f32 unless OCML_BASIC_ROUNDED_OPERATIONS is defined).
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }

// Correctly rounded sqrt for x = 0 or 2^-60 <= x <= 4 (what 1 - x*x and maxy*maxy - y*y of rtweekend.hpp:60-67,83-88 can
// be: multiples of 2^-48 in [0, 1]): the hardware estimate (v_sqrt_f32, <= 1 ulp) and the compiler's own two-sided
// neighbour test — is s - 1ulp or s + 1ulp the better root? — without the denormal rescaling and the inf/zero class test
// its general expansion carries (17 -> 9 issue slots, three times per lambertian bounce / camera ray).  Checked against
// the IEEE result for EVERY float of the range (tests/test_gpu_parity.py::test_unit_range_sqrt_is_correctly_rounded).
__device__ __forceinline__ float sqrt_rn_unit(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_lo = as_f(as_i(s) - 1), s_hi = as_f(as_i(s) + 1);
  const float r_lo = __builtin_fmaf(-s_lo, s, x); // x - s_lo*s <= 0: s is too large
  const float r_hi = __builtin_fmaf(-s_hi, s, x); // x - s_hi*s  > 0: s is too small
  float r = (r_lo <= 0.0f) ? s_lo : s;
  r = (r_hi > 0.0f) ? s_hi : r;
  return r;
}

struct Ray {
  V3 o, d;
  float tm;
};

// ---- RNG: xorshift.hpp:72-74 + rtweekend.hpp:39-88 -------------------------------------
__device__ __forceinline__ float rng_float(uint32_t& s) {
  s ^= s >> 7;
  s ^= s << 1;
  s ^= s >> 9;
  return (float)s * 2.3283064365386963e-10f; // v_cvt_f32_u32 (RNE) * 2^-32
}
__device__ __forceinline__ float rng_float(uint32_t& s, float mn, float mx) { return mn + (mx - mn) * rng_float(s); }

__device__ __forceinline__ V3 rng_unit_vec(uint32_t& s) {
  float x = rng_float(s, -1.0f, 1.0f);
  float maxy = sqrt_rn_unit(1.0f - x * x);
  float y = rng_float(s, -maxy, maxy);
  float absz = sqrt_rn_unit(maxy * maxy - y * y);
  float z = (rng_float(s) > 0.5f) ? absz : -absz;
  return mk(x, y, z);
}
__device__ __forceinline__ V3 rng_in_unit_ball(uint32_t& s) {
  float r = rng_float(s);
  float theta = rng_float(s, 0.0f, 2.0f * PT_PI);
  float phi = rng_float(s, 0.0f, PT_PI);
#ifdef PT_NO_SINCOS
  float plan_seed = r * ptm::sinf_(phi);
  float z = r * ptm::cosf_(phi);
  return mk(plan_seed * ptm::cosf_(theta), plan_seed * ptm::sinf_(theta), z);
#else
  float sp, cp, st, ct; // sin / cos of the two angles: one range reduction per angle (pt_math.hpp: sincosf_)
  ptm::sincosf_(phi, sp, cp);
  ptm::sincosf_(theta, st, ct);
  float plan_seed = r * sp;
  float z = r * cp;
  return mk(plan_seed * ct, plan_seed * st, z);
#endif
}

// ---- camera: render.hpp:96-99 + camera.hpp:93-100 ----------------------------------------
struct Cam { // PtCamera, passed by value in the kernarg segment (SGPRs)
  float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
  float lens_radius, time0, time1;
};

__device__ __forceinline__ float div_exact(float n, float d, float y, float q0); // below, with its proof obligations

// inv_w, inv_h: RN(1 / (float)width), RN(1 / (float)height), computed once on the host: the two quotients of render.hpp:96-97
// through the shared-reciprocal form (div_exact: bit-identical to the IEEE division for 1 <= divisor <= 2^24 and a
// numerator that is 0 or >= 2^-32 — its tested range), 5 issue slots each instead of ~12.
// pinhole (wave-uniform, decided on the host: cam_is_pinhole): lens_radius == 0 and no component of the camera's origin is a
// zero.  Then rd = 0 * (dx, dy, 0) is a vector of zeros, so is offset = rd.x u + rd.y v (dx, dy, u, v are finite), and
// origin + offset == origin, (...) - origin - offset == (...) - origin bit for bit (x + (+-0) == x unless x is -0, and a
// difference of floats is never -0 unless its first operand is): the lens arithmetic, and the square root that only scales
// the second draw, are skipped — the three draws are not (the generator's state must advance as in camera.hpp:93-100).
__device__ __forceinline__ Ray camera_ray(const Cam& c, int x, int y, int width, int height, float inv_w, float inv_h, uint32_t& rng,
                                          bool pinhole = false) {
  const float nu = (float)x + rng_float(rng);
  const float su = div_exact(nu, (float)width, inv_w, nu * inv_w);
  const float nv = (float)y + rng_float(rng);
  const float sv = div_exact(nv, (float)height, inv_h, nv * inv_h);
  V3 origin = mk(c.origin[0], c.origin[1], c.origin[2]);
  Ray r;
  if (pinhole) {
    (void)rng_float(rng); (void)rng_float(rng); // in_unit_disk's two draws
    r.o = origin;
    r.d = mk(c.llc[0], c.llc[1], c.llc[2]) + su * mk(c.horizontal[0], c.horizontal[1], c.horizontal[2]) +
          sv * mk(c.vertical[0], c.vertical[1], c.vertical[2]) - origin;
  } else {
    // in_unit_disk rtweekend.hpp:83-88
    float dx = rng_float(rng, -1.0f, 1.0f);
    float maxy = sqrt_rn_unit(1.0f - dx * dx);
    float dy = rng_float(rng, -maxy, maxy);
    V3 rd = c.lens_radius * mk(dx, dy, 0.0f);
    V3 U = mk(c.u[0], c.u[1], c.u[2]), Vv = mk(c.v[0], c.v[1], c.v[2]);
    V3 offset = rd.x * U + rd.y * Vv;
    r.o = origin + offset;
    r.d = mk(c.llc[0], c.llc[1], c.llc[2]) + su * mk(c.horizontal[0], c.horizontal[1], c.horizontal[2]) +
          sv * mk(c.vertical[0], c.vertical[1], c.vertical[2]) - origin - offset;
  }
  r.tm = rng_float(rng, c.time0, c.time1);
  return r;
}
__host__ __device__ inline bool cam_is_pinhole(const Cam& c) {
  return c.lens_radius == 0.0f && c.origin[0] != 0.0f && c.origin[1] != 0.0f && c.origin[2] != 0.0f;
}

// ---- per-ray context ------------------------------------------------------------------------------
// Everything the primitive loop needs that is constant for one ray.
//
// Exact division without dividing (rect/box sides, 43 per ray in the Cornell-style scene): the
// reference computes t = (k - o_c) / d_c per side (rectangle.hpp:34,72,110).  For a "regular" ray
// (o, d finite, 2^-40 <= |d_c| <= 2^40, |o_c| <= 2^60, scene coordinates <= 2^60) the correctly rounded
// quotient is obtained from ONE correctly rounded reciprocal per axis, y = RN(1/d_c), shared by all sides:
//     q0 = RN(n*y); r0 = fma(-d,q0,n); q1 = fma(r0,y,q0); r1 = fma(-d,q1,n); t = fma(r1,y,q1)
// — the Markstein-style correction the compiler's own fdiv expansion ends with, with a better y and no
// scaling needed inside the guarded range (no overflow / denormal intermediate).  Checked exhaustively-ish:
// 3.2e9 random + adversarial (n, d) pairs, zero mismatches vs n/d, and on the GPU by
// tests/test_gpu_parity.py::test_fast_division_is_exact.  The reference's comparisons are then applied to
// that t unchanged, as straight-line predicated code (rect_fast/box_fast).  A wave with any irregular live
// ray (axis-parallel, NaN, huge) takes the plain-division path for that iteration (wave-uniform branch).
struct RayCtx {
  Ray r;
  float a;          // dot(d,d), hoisted (sphere.hpp:69)
  float yx, yy, yz; // RN(1/d_c) for regular rays
  bool reg;
  bool live;        // this lane's ray is wanted (idle lanes scan along and their outcome is dropped: they must not vote)
};

// RN(1/d) for 2^-40 <= |d| <= 2^40: the hardware estimate (v_rcp_f32, <= 1 ulp) and ONE Newton step in fma arithmetic.
// Correctly rounded for EVERY significand: checked exhaustively (all 2^23, both signs, eight exponents across the range;
// the computation is scale-invariant while nothing over- or underflows) by tests/test_gpu_parity.py::
// test_guarded_reciprocal_is_correctly_rounded against the IEEE quotient.  3 instructions instead of the ~10 of the full
// division expansion (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup), three times per ray.
__device__ __forceinline__ float rcp_rn_guarded(float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
}

__device__ __forceinline__ RayCtx make_ctx(const Ray& r, bool scene_fast_ok) {
  RayCtx c;
  c.r = r;
  c.live = true;
  c.a = dot(r.d, r.d);
  const float lo = 9.094947017729282e-13f, hi = 1.099511627776e12f, ohi = 1.152921504606846976e18f; // 2^-40 2^40 2^60
  float ax = __builtin_fabsf(r.d.x), ay = __builtin_fabsf(r.d.y), az = __builtin_fabsf(r.d.z);
  c.reg = scene_fast_ok && ax >= lo && ax <= hi && ay >= lo && ay <= hi && az >= lo && az <= hi &&
          __builtin_fabsf(r.o.x) <= ohi && __builtin_fabsf(r.o.y) <= ohi && __builtin_fabsf(r.o.z) <= ohi;
  c.yx = rcp_rn_guarded(r.d.x); // only used when the ray is regular
  c.yy = rcp_rn_guarded(r.d.y);
  c.yz = rcp_rn_guarded(r.d.z);
  __builtin_amdgcn_sched_barrier(0); // keep the context ahead of the scan: interleaving it with the record loops spills
  return c;
}

// n / d for a regular ray axis; y = RN(1/d).  Bit-identical to the IEEE quotient inside the guarded range.
__device__ __forceinline__ float div_exact(float n, float d, float y, float q0) {
  float r0 = __builtin_fmaf(-d, q0, n);
  float q1 = __builtin_fmaf(r0, y, q0);
  float r1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(r1, y, q1);
}

// ---- sphere.hpp -----------------------------------------------------------------------------
// record: R0 = (c0.xyz, +-radius^2: sign bit set = moving)  R1 = (radius, mat, time0, time1)  R2 = (c1-c0, hittable index)

// The time fraction (time - time0) / (time1 - time0) of sphere.hpp:54 depends on the ray and on (time0, time1) only, and
// consecutive moving spheres of a list nearly always share their shutter interval: the quotient is kept while the
// interval stays the same (one IEEE division per ray instead of one per moving sphere; same operation, same bits).
struct TimeFrac {
  float t0, t1, frac;
};
__device__ __forceinline__ TimeFrac time_frac_none() { return TimeFrac{__builtin_nanf(""), __builtin_nanf(""), 0.0f}; }
__device__ __forceinline__ float time_frac(TimeFrac& m, float time, float t0, float t1) {
  if (!(t0 == m.t0 && t1 == m.t1)) { m.t0 = t0; m.t1 = t1; m.frac = (time - t0) / (t1 - t0); }
  return m.frac;
}

// sphere.hpp:51-56
__device__ __forceinline__ V3 sphere_center(f4 R0, f4 R1, f4 R2, float time) {
  V3 c0 = xyz(R0);
  if (R1.z == R1.w) return c0; // wave-uniform
  return c0 + ((time - R1.z) / (R1.w - R1.z)) * xyz(R2);
}

// sphere.hpp:13-24
__device__ __forceinline__ void mercator(V3 p, float& u, float& v) {
  float phi = ptm::atan2f_(p.z, p.x);
  float theta = ptm::asinf_(p.y);
  u = 1.0f - (phi + PT_PI) / (2.0f * PT_PI);
  v = (theta + PT_PI / 2.0f) / PT_PI;
}

// ---- an image texture on a sphere needs only the TEXEL its (u, v) select (texture.hpp:140-157), and the mercator pair costs two binary64
// transcendentals with divisions (atan2f_, asinf_: ~135 binary64 instructions, entered by a wave whenever ONE lane shades such a hit: 7 % of
// the 496-hittable scene's frame, profiles/r04_ab_texel.txt).  The texel indices are floor()s of  c_i = fmod1(u freq) (w - 1),
// c_j = (1 - fmod1(v freq)) (h - 1),  so an approximation of u, v that is provably within E_uv of the exact chain's values decides them
// unless c lies within E of an integer — and then (and for every argument outside the approximation's domain: NaN, a pole, |y| > 1,
// freq <= 0, a texture of one texel) the lane takes the exact chain, as before.
// The approximation, all binary32: u~ = 1/2 - atan2(z, x) / 2pi as  a P_A(a^2), a = min(|x|,|z|) / max(|x|,|z|) <= 1  (degree 8 in a^2, the
// 1/2pi folded into the coefficients; |error| < 1e-8) with the octant reflections done in turns (values <= 1/2: every rounding <= 2^-25);
// v~ = 1/2 + asin(y) / pi as  s P_S(s^2)  with s = |y| for |y| <= 1/2 and s = sqrt((1 - |y|) / 2) (exact difference, 1-ulp root) otherwise
// (degree 5, |error| < 3e-9).  Against the exact chain (atan2f_ / asinf_ rounded to binary32, then the reference's own binary32 operations)
// the deviation is 1.2e-7 at most in u and in v on 2 x 10^8 simulated directions (numpy binary32) and is MEASURED ON THE DEVICE by
// tests/test_gpu_parity.py::test_sphere_texel_fast_path_is_exact (pt_debug_sphere_texel returns both pairs; poles, the seam and the axes
// over-represented; the test's bar is 2.5e-7); E_uv = 1e-6, four times that bar, is what the ambiguity test assumes, and the chain behind
// u adds its own roundings: |c~ - c| <= (w - 1) (freq E_uv + 2 ulp(freq)) + 2 ulp(c)
// < (w - 1) (freq + 1) 1.25e-6 =: E for w <= 65536.  With E < 1/4, c~ finite and E <= c~ - floor(c~) <= 1 - E, 0 < c~ < w - 1: floor(c) = floor(c~)
// (an argument of fmod1 on the other side of an integer than its approximation puts c~ within E of 0 or of w - 1: integers).
// the two approximations: u~, v~ of the unit normal n (NaN / garbage outside the domain: sphere_texel_fast's `ok` covers that)
__device__ __forceinline__ void sphere_uv_fast(V3 n, float& u_out, float& v_out) {
  const float ax = __builtin_fabsf(n.x), az = __builtin_fabsf(n.z), ay = __builtin_fabsf(n.y);
  const float mxv = __builtin_fmaxf(ax, az), mnv = __builtin_fminf(ax, az);
  const float a = mnv * __builtin_amdgcn_rcpf(mxv);
  const float a2 = a * a;
  float pa = 0.0004402677f;
  pa = __builtin_fmaf(pa, a2, -0.00250370614f);
  pa = __builtin_fmaf(pa, a2, 0.00670641102f);
  pa = __builtin_fmaf(pa, a2, -0.0118679535f);
  pa = __builtin_fmaf(pa, a2, 0.0168996621f);
  pa = __builtin_fmaf(pa, a2, -0.0225964971f);
  pa = __builtin_fmaf(pa, a2, 0.0318180509f);
  pa = __builtin_fmaf(pa, a2, -0.0530511774f);
  pa = __builtin_fmaf(pa, a2, 0.159154937f);
  float t = a * pa;                 // atan(a) / 2pi in [0, 1/8]
  t = az > ax ? 0.25f - t : t;      // first quadrant, in turns
  t = n.x < 0.0f ? 0.5f - t : t;
  t = n.z < 0.0f ? -t : t;          // atan2(z, x) / 2pi in [-1/2, 1/2]
  const float u = 0.5f - t;         // 1 - (phi + pi) / 2pi
  const bool big = ay > 0.5f;
  const float sv = big ? __builtin_amdgcn_sqrtf((1.0f - ay) * 0.5f) : ay;
  const float s2 = sv * sv;
  float ps = 0.0135018695f;
  ps = __builtin_fmaf(ps, s2, 0.00763752731f);
  ps = __builtin_fmaf(ps, s2, 0.0144896684f);
  ps = __builtin_fmaf(ps, s2, 0.0238563605f);
  ps = __builtin_fmaf(ps, s2, 0.0530520156f);
  ps = __builtin_fmaf(ps, s2, 0.318309873f);
  float wv = sv * ps;               // asin(s) / pi
  wv = big ? __builtin_fmaf(-2.0f, wv, 0.5f) : wv;
  wv = n.y < 0.0f ? -wv : wv;       // asin(y) / pi in [-1/2, 1/2]
  u_out = u;
  v_out = 0.5f + wv;                // (theta + pi/2) / pi
}
__device__ __forceinline__ bool sphere_texel_fast(V3 n, float freq, uint32_t w, uint32_t h, uint32_t& i, uint32_t& j) {
#ifdef PT_NO_TEXEL_SHORTCUT
  return false;
#else
  const float ax = __builtin_fabsf(n.x), az = __builtin_fabsf(n.z), ay = __builtin_fabsf(n.y);
  const float mxv = __builtin_fmaxf(ax, az);
  float u, v;
  sphere_uv_fast(n, u, v);
  const float wm = (float)(w - 1u), hm = (float)(h - 1u);
  const float E = 1.25e-6f * (freq + 1.0f);
  const float Ei = wm * E, Ej = hm * E;
  const float xu = u * freq, xv = v * freq;
  const float ci = (xu - __builtin_floorf(xu)) * wm;
  const float cj = (1.0f - (xv - __builtin_floorf(xv))) * hm;
  const float fi = __builtin_floorf(ci), fj = __builtin_floorf(cj);
  const float di = ci - fi, dj = cj - fj;
  // every comparison is written so that a NaN anywhere fails it
  const bool ok = freq > 0.0f && freq <= 65536.0f && w >= 2u && h >= 2u && w <= 65536u && h <= 65536u && Ei < 0.25f && Ej < 0.25f &&
                  ay <= 1.0f && ax <= 3.0e38f && az <= 3.0e38f /* (fmin / fmax drop a NaN operand) */ && mxv > 0.0f && ci > 0.0f && ci < wm && cj > 0.0f && cj < hm && di >= Ei && di <= 1.0f - Ei && dj >= Ej && dj <= 1.0f - Ej;
  i = (uint32_t)fi; j = (uint32_t)fj;
  return ok;
#endif
}

// Roots of sphere.hpp:68-93: calls accept(t) if one lies in (mn, mx) and `valid`.  The acceptance runs INSIDE the
// discriminant branch, which a wave enters only when some lane's line meets the sphere: returning a flag instead made the
// compiler update the caller's state with three v_cndmask on every sphere of the list.
// first half: centre, b, discriminant — branch-free except the wave-uniform "moving" read
struct SphereEval { float b, disc; };
template <typename P>
__device__ __forceinline__ SphereEval sphere_eval(P recs, int off, const RayCtx& c, TimeFrac& tf) {
  const Ray& r = c.r;
  f4 R0 = recs[off]; // the only read on the miss path of a static sphere
  V3 center = xyz(R0);
  if (as_i(R0.w) < 0) { // moving (flatten stores -(radius^2) when time0 != time1): wave-uniform
    const f4 R1 = recs[off + 1], R2 = recs[off + 2]; // one LDS round trip for both
    center = center + time_frac(tf, r.tm, R1.z, R1.w) * xyz(R2);
  }
  V3 oc = r.o - center;
  float b = dot(oc, r.d);
  float cc = dot(oc, oc) - __builtin_fabsf(R0.w);
  return SphereEval{b, b * b - c.a * cc};
}
// second half: the roots.  (A straight-line version was tried: 2.5x slower — most spheres of a long list are missed by
// all 64 lines of a wave, so the branch skips the block for the whole wave.)
template <typename Accept>
__device__ __forceinline__ void sphere_finish(SphereEval e, const RayCtx& c, float mn, float mx, bool valid, Accept accept) {
  if (e.disc > 0) {
    float sq = sqrt_rn(e.disc);
    float temp = (-e.b - sq) / c.a;
    bool ok = temp < mx && temp > mn;
    if (!ok) {
      temp = (-e.b + sq) / c.a;
      ok = temp < mx && temp > mn;
    }
    if (ok && valid) accept(temp);
  }
}
template <typename P, typename Accept>
__device__ __forceinline__ void sphere_roots(P recs, int off, const RayCtx& c, float mn, float mx, bool valid, TimeFrac& tf,
                                             Accept accept) {
  sphere_finish(sphere_eval(recs, off, c, tf), c, mn, mx, valid, accept);
}

template <typename P>
__device__ __forceinline__ bool sphere_t(P recs, int off, const RayCtx& c, float mn, float mx, float& t, TimeFrac& tf) {
  bool hit = false;
  sphere_roots(recs, off, c, mn, mx, true, tf, [&](float temp) { t = temp; hit = true; });
  return hit;
}

// ---- rectangle.hpp:31-49,69-87,107-125 --------------------------------------------------------
// record: R0 = (a0, a1, b0, b1)  R1 = (k, mat, axis, hittable index); axis 0 xy, 1 xz, 2 yz
// AX is a compile-time constant so the operand selection costs nothing (a run-time axis index into the ray
// puts the seven selected floats in scratch memory).
template <int AX> struct AxisSel;
template <> struct AxisSel<0> { // xy_rect: plane z = k
  static __device__ __forceinline__ float ok(const RayCtx& c) { return c.r.o.z; }
  static __device__ __forceinline__ float dk(const RayCtx& c) { return c.r.d.z; }
  static __device__ __forceinline__ float yk(const RayCtx& c) { return c.yz; }
  static __device__ __forceinline__ float oa(const RayCtx& c) { return c.r.o.x; }
  static __device__ __forceinline__ float da(const RayCtx& c) { return c.r.d.x; }
  static __device__ __forceinline__ float ob(const RayCtx& c) { return c.r.o.y; }
  static __device__ __forceinline__ float db(const RayCtx& c) { return c.r.d.y; }
};
template <> struct AxisSel<1> { // xz_rect: plane y = k
  static __device__ __forceinline__ float ok(const RayCtx& c) { return c.r.o.y; }
  static __device__ __forceinline__ float dk(const RayCtx& c) { return c.r.d.y; }
  static __device__ __forceinline__ float yk(const RayCtx& c) { return c.yy; }
  static __device__ __forceinline__ float oa(const RayCtx& c) { return c.r.o.x; }
  static __device__ __forceinline__ float da(const RayCtx& c) { return c.r.d.x; }
  static __device__ __forceinline__ float ob(const RayCtx& c) { return c.r.o.z; }
  static __device__ __forceinline__ float db(const RayCtx& c) { return c.r.d.z; }
};
template <> struct AxisSel<2> { // yz_rect: plane x = k
  static __device__ __forceinline__ float ok(const RayCtx& c) { return c.r.o.x; }
  static __device__ __forceinline__ float dk(const RayCtx& c) { return c.r.d.x; }
  static __device__ __forceinline__ float yk(const RayCtx& c) { return c.yx; }
  static __device__ __forceinline__ float oa(const RayCtx& c) { return c.r.o.y; }
  static __device__ __forceinline__ float da(const RayCtx& c) { return c.r.d.y; }
  static __device__ __forceinline__ float ob(const RayCtx& c) { return c.r.o.z; }
  static __device__ __forceinline__ float db(const RayCtx& c) { return c.r.d.z; }
};

// The reference's test as written, with the IEEE division.  Used for irregular rays and for
// constant_medium boundaries (min = -inf: tiny/zero quotients matter there).
template <int AX>
__device__ __forceinline__ bool rect_plain(float a0, float a1, float b0, float b1, float k, const RayCtx& c,
                                           float mn, float mx, float& t_out, float& a_out, float& b_out) {
  typedef AxisSel<AX> S;
  float t = (k - S::ok(c)) / S::dk(c);
  if (t < mn || t > mx) return false;
  float a = S::oa(c) + t * S::da(c);
  float b = S::ob(c) + t * S::db(c);
  if (a < a0 || a > a1 || b < b0 || b > b1) return false;
  t_out = t; a_out = a; b_out = b;
  return true;
}

// Same predicate, straight-line, for a wave whose live rays are all regular and min = PT_TMIN > 0:
// t comes from the shared reciprocal (div_exact) — identical bits whenever |t| >= 2^-60; below that both
// the exact and the computed t are < min, so the decision is the same.  No branch: with 64 incoherent rays
// per wave some lane passes almost every sub-test, so early-outs only add exec-mask bookkeeping.
template <int AX>
__device__ __forceinline__ bool rect_fast(float a0, float a1, float b0, float b1, float k, const RayCtx& c,
                                          float mx, float& t_out, float& a_out, float& b_out) {
  typedef AxisSel<AX> S;
  const float n = k - S::ok(c);
  const float t = div_exact(n, S::dk(c), S::yk(c), n * S::yk(c));
  const float a = S::oa(c) + t * S::da(c);
  const float b = S::ob(c) + t * S::db(c);
  t_out = t; a_out = a; b_out = b;
  // !(t < min || t > max) && !(a < a0 || a > a1 || b < b0 || b > b1).  A regular ray on a fast_ok scene cannot produce a
  // NaN here (t is finite, a and b at worst +-inf) and every interval is lo <= hi (flatten; closest >= min always), so
  // "x inside [lo, hi]" is "median(x, lo, hi) == x": three v_med3 + three v_cmp_eq + two s_and instead of six v_cmp +
  // five s_and — the scalar unit is shared by the four SIMDs of a CU and this loop runs 0.43 SALU per VALU.
  return (__builtin_amdgcn_fmed3f(t, PT_TMIN, mx) == t) & (__builtin_amdgcn_fmed3f(a, a0, a1) == a) &
         (__builtin_amdgcn_fmed3f(b, b0, b1) == b);
}

// run-time axis (top-level rect records): wave-uniform dispatch to the three instantiations
__device__ __forceinline__ bool rect_any(bool fast, int axis, float a0, float a1, float b0, float b1, float k,
                                         const RayCtx& c, float mx, float& t, float& a, float& b) {
  if (fast) {
    if (axis == 0) return rect_fast<0>(a0, a1, b0, b1, k, c, mx, t, a, b);
    if (axis == 1) return rect_fast<1>(a0, a1, b0, b1, k, c, mx, t, a, b);
    return rect_fast<2>(a0, a1, b0, b1, k, c, mx, t, a, b);
  }
  if (axis == 0) return rect_plain<0>(a0, a1, b0, b1, k, c, PT_TMIN, mx, t, a, b);
  if (axis == 1) return rect_plain<1>(a0, a1, b0, b1, k, c, PT_TMIN, mx, t, a, b);
  return rect_plain<2>(a0, a1, b0, b1, k, c, PT_TMIN, mx, t, a, b);
}

// ---- box.hpp:29-50: nearest of the six sides in constructor order (box.hpp:20-25) ---------------
// record: R0 = (x0,y0,z0, mat)  R1 = (x1,y1,z1, hittable index)
#define PT_BOX_SIDES(R0, R1)                          \
  PT_BOX_SIDE(0, 0, R0.x, R1.x, R0.y, R1.y, R1.z)     \
  PT_BOX_SIDE(1, 0, R0.x, R1.x, R0.y, R1.y, R0.z)     \
  PT_BOX_SIDE(2, 1, R0.x, R1.x, R0.z, R1.z, R1.y)     \
  PT_BOX_SIDE(3, 1, R0.x, R1.x, R0.z, R1.z, R0.y)     \
  PT_BOX_SIDE(4, 2, R0.y, R1.y, R0.z, R1.z, R1.x)     \
  PT_BOX_SIDE(5, 2, R0.y, R1.y, R0.z, R1.z, R0.x)

template <bool UV>
__device__ __forceinline__ bool box_plain(f4 R0, f4 R1, const RayCtx& c, float mn, float mx, float& t_out, int& side_out,
                                          float& u_out, float& v_out) {
  bool hit = false;
  float closest = mx;
  float t, a, b;
#define PT_BOX_SIDE(S, AX, A0, A1, B0, B1, K)                          \
  if (rect_plain<AX>(A0, A1, B0, B1, K, c, mn, closest, t, a, b)) {    \
    hit = true; closest = t; side_out = S;                             \
    if (UV) { u_out = (a - (A0)) / ((A1) - (A0)); v_out = (b - (B0)) / ((B1) - (B0)); } \
  }
  PT_BOX_SIDES(R0, R1)
#undef PT_BOX_SIDE
  t_out = closest;
  return hit;
}

template <bool UV>
__device__ __forceinline__ bool box_fast(f4 R0, f4 R1, const RayCtx& c, float mx, float& t_out, int& side_out,
                                         float& u_out, float& v_out) {
  bool hit = false;
  float closest = mx;
  float t, a, b;
#define PT_BOX_SIDE(S, AX, A0, A1, B0, B1, K)                          \
  {                                                                    \
    const bool acc = rect_fast<AX>(A0, A1, B0, B1, K, c, closest, t, a, b); \
    hit |= acc;                                                        \
    closest = acc ? t : closest;                                       \
    side_out = acc ? S : side_out;                                     \
    if (UV) { if (acc) { u_out = (a - (A0)) / ((A1) - (A0)); v_out = (b - (B0)) / ((B1) - (B0)); } } \
  }
  PT_BOX_SIDES(R0, R1)
#undef PT_BOX_SIDE
  t_out = closest;
  return hit;
}

// ---- the same six sides with the acceptance as an exec-narrowing chain ------------------------------------------------
// A SIMD issues ONE instruction per two cycles whatever its kind, so the scalar instructions of a side count like its
// vector ones.  rect_fast spends, besides the ten arithmetic instructions of t, a, b: 3 v_med3 + 3 v_cmp + 2 s_and + the
// select of closest, and the box another ~9 for the side index and the hit flag — ~22 issue slots per side.  Here the
// reference's own six comparisons (rectangle.hpp:36,40: !(t < min) !(t > max) !(a < a0) !(a > a1) !(b < b0) !(b > b1), NaN
// behaviour included) narrow EXEC one after the other (v_cmpx writes EXEC), the two moves that record an accepted side
// run under the narrowed mask, and one s_mov restores it: 10 + 6 + 2 + 1 = 19 slots per side and no epilogue.  The block
// is ONE asm statement, so nothing the compiler schedules can run under the narrowed mask; it declares what it touches
// (closest, hit in/out; vcc clobbered: v_cmpx_e32 also writes it; EXEC is restored before the block ends).
// `hit_base` = hit id of this box with side 0, in a VGPR; side S adds S << PT_HIT_SIDE_SHIFT (hit_pack).
template <int AX, int S>
__device__ __forceinline__ void rect_side_cmpx(float a0, float a1, float b0, float b1, float k, const RayCtx& c,
                                               unsigned long long exec_all, int hit_base, float& closest, int& hit) {
  typedef AxisSel<AX> Sel;
  const float n = k - Sel::ok(c);
  const float t = div_exact(n, Sel::dk(c), Sel::yk(c), n * Sel::yk(c));
  const float a = Sel::oa(c) + t * Sel::da(c);
  const float b = Sel::ob(c) + t * Sel::db(c);
  asm volatile(
      "v_cmpx_le_f32_e32 vcc, %[tmin], %[t]\n\t"   // min <= t  ==  !(t < min): t is never NaN here (a slab pool's rect entry relies on NaN failing)
      "v_cmpx_ngt_f32_e32 vcc, %[t], %[cl]\n\t"    // !(t > max)
      "v_cmpx_nlt_f32_e32 vcc, %[a], %[a0]\n\t"    // !(a < a0)
      "v_cmpx_ngt_f32_e32 vcc, %[a], %[a1]\n\t"    // !(a > a1)
      "v_cmpx_nlt_f32_e32 vcc, %[b], %[b0]\n\t"    // !(b < b0)
      "v_cmpx_ngt_f32_e32 vcc, %[b], %[b1]\n\t"    // !(b > b1)
      "v_mov_b32_e32 %[cl], %[t]\n\t"
      "v_or_b32_e32 %[hit], %[sbits], %[base]\n\t"
      "s_mov_b64 exec, %[all]"
      : [cl] "+v"(closest), [hit] "+v"(hit)
      : [t] "v"(t), [a] "v"(a), [b] "v"(b), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [tmin] "s"(PT_TMIN),
        [sbits] "n"(S << PT_HIT_SIDE_SHIFT), [base] "v"(hit_base), [all] "s"(exec_all)
      : "vcc");
}

// box.hpp:29-50 for a regular ray on a fast_ok scene; closest / hit are the traversal's own running state
__device__ __forceinline__ void box_cmpx(f4 R0, f4 R1, const RayCtx& c, unsigned long long exec_all, int hit_base, float& closest, int& hit) {
#define PT_BOX_SIDE(S, AX, A0, A1, B0, B1, K) rect_side_cmpx<AX, S>(A0, A1, B0, B1, K, c, exec_all, hit_base, closest, hit);
  PT_BOX_SIDES(R0, R1)
#undef PT_BOX_SIDE
}

// ---- triangle.hpp:58-100 (Moller-Trumbore) ---------------------------------------------------------
// record: R0 = (v0.xyz, mat)  R1 = (edge1.xyz, hittable index)  R2 = (edge2.xyz, 0); edges = v1-v0, v2-v0
// first half: a = e1.(d x e2), u = s.(d x e2) and the first two rejections of triangle.hpp:71-81 evaluated together
struct TriEval { float a, u; bool pass; };
__device__ __forceinline__ TriEval tri_eval(f4 R0, f4 R1, f4 R2, const Ray& r) {
  const float epsilon = 0.0000001f;
  V3 edge1 = xyz(R1), edge2 = xyz(R2);
  V3 h = cross(r.d, edge2);
  float a = dot(edge1, h);
  float a_abs = __builtin_fabsf(a);
  bool a_pos = a > 0.0f;
  V3 s = r.o - xyz(R0);
  float u = dot(s, h);
  bool u_pos = u > 0.0f;
  return TriEval{a, u, !((a_abs < epsilon) | (u_pos != a_pos) | (__builtin_fabsf(u) > a_abs))};
}
// second half, under ONE exec-mask branch that a wave rarely enters; the acceptance runs inside it (returning a flag made
// the compiler update the caller's state with selects on every triangle of the list)
template <typename Accept>
__device__ __forceinline__ void tri_finish(f4 R0, f4 R1, f4 R2, const Ray& r, TriEval e, float mn, float mx, bool valid,
                                           Accept accept) {
  if (e.pass) {
    const float a_abs = __builtin_fabsf(e.a);
    const bool a_pos = e.a > 0.0f;
    V3 edge1 = xyz(R1), edge2 = xyz(R2);
    V3 s = r.o - xyz(R0);
    V3 q = cross(s, edge1);
    float v = dot(r.d, q);
    bool v_pos = v > 0.0f;
    if (!((v_pos != a_pos) | (__builtin_fabsf(e.u + v) > a_abs))) {
      float length = dot(edge2, q) / e.a;
      if (!(length < mn || length > mx) && valid) accept(length);
    }
  }
}
__device__ __forceinline__ bool tri_t(f4 R0, f4 R1, f4 R2, const Ray& r, float mn, float mx, float& t_out) {
  bool hit = false;
  tri_finish(R0, R1, R2, r, tri_eval(R0, R1, R2, r), mn, mx, true, [&](float t) { t_out = t; hit = true; });
  return hit;
}

// ---- triangle.hpp:14-56 (Badouel strategy: the alternative argument of _triangle<>) -----------------------------------
// Same record as the Moller-Trumbore triangle (edges precomputed with the same subtractions, triangle.hpp:19-20).  A
// parity-completeness path (main.cpp never instantiates it): one triangle at a time, compiled only into the kernels that
// scenes with such triangles run (template flag BADOUEL), so it costs the others neither code nor registers.
template <typename Accept>
__device__ __forceinline__ void badouel_test(f4 R0, f4 R1, f4 R2, const Ray& r, float mn, float mx, Accept accept) {
  const V3 u = xyz(R1), v = xyz(R2);
  const V3 outward_normal = cross(u, v);
  const V3 w0 = r.o - xyz(R0);
  const float a = -dot(outward_normal, w0);
  const float b = dot(outward_normal, r.d);
  if (__builtin_fabsf(b) < 0.000001f) return; // parallel to the plane
  const float length = a / b;
  if (length < 0) return;
  if (length < mn || length > mx) return;
  const V3 hit_pt = r.o + length * r.d;
  const float uu = dot(u, u), uv = dot(u, v), vv = dot(v, v);
  const V3 w = hit_pt - xyz(R0);
  const float wu = dot(w, u), wv = dot(w, v);
  const float D = uv * uv - uu * vv;
  const float s = (uv * wv - vv * wu) / D;
  const float t = (uv * wu - uu * wv) / D;
  if (s < 0.0f || s > 1.0f || t < 0.0f || (s + t) > 1.0f) return;
  accept(length);
}

// ---- constant_medium.hpp:28-78 -------------------------------------------------------------------------
// record: R0 = (boundary kind, neg_inv_density, mat, hittable index)  R1.. = boundary (sphere 3 f4 | box 2 f4)
template <typename P>
__device__ __forceinline__ bool boundary_t(P recs, int off, int bkind, const RayCtx& c, float mn, float mx, float& t) {
  if (bkind == DK_SPHERE) { TimeFrac tf = time_frac_none(); return sphere_t(recs, off, c, mn, mx, t, tf); }
  int side; float u, v;
  return box_plain<false>(recs[off], recs[off + 1], c, mn, mx, t, side, u, v);
}

template <typename P>
__device__ __forceinline__ bool medium_t(P recs, int off, const RayCtx& c, float mn, float mx, uint32_t& rng,
                                         float& t_out) {
  f4 R0 = recs[off];
  int bkind = as_i(R0.x);
  float t1, t2;
  if (bkind == DK_SPHERE) {
    // The two boundary hits of constant_medium.hpp:35-41 — hit(r, -inf, inf, rec1), hit(r, rec1.t + 0.0001, inf, rec2) — are the SAME sphere
    // against the SAME ray: centre, b, discriminant, square root and both quotients come out identical in the second call, only the
    // window differs.  They are evaluated once (one discriminant, one square root, two divisions instead of two, two and up to four) and
    // sphere.hpp:74-91's selection is applied twice, to the same values: same bits (round 4; the generic two-call form stays for boxes).
    TimeFrac tf = time_frac_none();
    const SphereEval e = sphere_eval(recs, off + 1, c, tf);
    if (!(e.disc > 0)) return false;
    const float sq = sqrt_rn(e.disc);
    const float r1 = (-e.b - sq) / c.a, r2 = (-e.b + sq) / c.a;
    const bool in1 = r1 < PT_INF && r1 > -PT_INF;                 // first root inside (-inf, inf)?
    if (!in1 && !(r2 < PT_INF && r2 > -PT_INF)) return false;
    t1 = in1 ? r1 : r2;
    const float mn2 = t1 + 0.0001f;
    const bool in2 = r1 < PT_INF && r1 > mn2;                     // the second call looks at the first root again, then at the second
    if (!in2 && !(r2 < PT_INF && r2 > mn2)) return false;
    t2 = in2 ? r1 : r2;
  } else {
    if (!boundary_t(recs, off + 1, bkind, c, -PT_INF, PT_INF, t1)) return false;
    if (!boundary_t(recs, off + 1, bkind, c, t1 + 0.0001f, PT_INF, t2)) return false;
  }
  if (t1 < mn) t1 = mn;
  if (t2 > mx) t2 = mx;
  if (t1 >= t2) return false;
  if (t1 < 0) t1 = 0;
  float a_here = c.a;
  asm volatile("" : "+v"(a_here)); // (opaque: the square root stays HERE, behind the boundary tests — hoisted to the top of the iteration it ran for every ray of every wave)
  const float ray_length = sqrt_rn(a_here); // sycl::length(r.direction())
  const float distance_inside_boundary = (t2 - t1) * ray_length;
  const float hit_distance = R0.y * ptm::logf_(rng_float(rng)); // the in-traversal draw (:65)
  if (hit_distance > distance_inside_boundary) return false;
  t_out = t1 + hit_distance / ray_length;
  return true;
}

// ---- hit_world: render.hpp:30-51 --------------------------------------------------------------------------
// Per-lane traversal state.  IMG: the scene has an image texture, so u,v of every accepted candidate are
// tracked exactly as the reference's temp_rec does (incl. the stale values triangles/media leave behind).
struct HitState {
  float closest;
  int hit;
  float u, v;
};
__device__ __forceinline__ void hit_begin(HitState& h) { h.closest = PT_INF; h.hit = -1; h.u = 0.0f; h.v = 0.0f; }

// ---- a run of spheres ------------------------------------------------------------------------------------------------
// What bounds this loop is instruction ISSUE, all kinds together: a SIMD issues one instruction per two cycles, and a scalar
// instruction takes that slot like a vector one (PMC, 496-hittable scene: VALU 0.62 + SALU 0.26 + LDS/branch/waits of the
// issue slots; hiding the LDS latency alone changed nothing).  So the scan is organised to need the fewest instructions per
// sphere of ANY kind: no per-sphere "is it moving" test, no per-sphere loop control.
// In front of a sphere run's records the flattener puts (pt_flatten.hpp: put_sphere_run_aux)
//     [static list][moving list][aux f4 = (time0, time1, number of static spheres, flags)]
// the record offsets (f4 units, relative to the run) of the run's static and of its moving spheres, each list in list order
// and padded to a multiple of four entries by repeating its last entry.  flags bit 0: every moving sphere of the run has
// the shutter interval (time0, time1) — then (time - time0) / (time1 - time0) of sphere.hpp:54 is ONE division per ray and
// run.  The lists are read through the scalar cache, four offsets per s_load; a trip issues the first records (and for the
// moving list the R2 = center1 - center0 records) of four spheres together, evaluates the discriminants and takes the roots.
// Scanning the static spheres before the moving ones is not list order, so acceptance carries the reference's tie rule
// explicitly: the sequential scan keeps the FIRST sphere in list order among equal t (sphere.hpp:77 needs t < max), i.e. a
// candidate replaces an equal-t hit iff that hit is a record of this run with a larger offset (records stay in list order
// in the blob, so offsets compare like list positions; a hit of an earlier run has a smaller offset and stays).  Repeating
// a sphere (the padding) is therefore a no-op.  Every candidate's t is independent of max, so the result is
// the sequential scan's, bit for bit.
typedef const __attribute__((address_space(4))) i4* cst_i4p;

// sphere_finish with the tie rule (see above); off_here = blob offset of this sphere's record
// (Round 3 tried the roots through ONE reciprocal of a = d.d per ray and run + div_exact's correction and a range-guarded
// v_sqrt + neighbour test — 19 issue slots instead of 41, bit-exact in an exhaustive sweep — and the 496-hittable scene got
// 2.5 % SLOWER: its frames follow the dependent chain of an iteration, and the corrected quotient is five dependent fma
// where the compiler's expansion overlaps its steps.  Not kept; DESIGN.md, rejected experiments.)
template <typename Accept>
__device__ __forceinline__ void sphere_finish_unordered(SphereEval e, const RayCtx& c, float mn, const HitState& h, int off_here,
                                                        Accept accept) {
  if (e.disc > 0) {
    const float mx = h.closest;
    const bool later = (h.hit >= 0) & (hit_off(h.hit) > off_here);
    float sq = sqrt_rn(e.disc);
    float temp = (-e.b - sq) / c.a;
    bool ok = (temp < mx || (temp == mx && later)) && temp > mn;
    if (!ok) {
      temp = (-e.b + sq) / c.a;
      ok = (temp < mx || (temp == mx && later)) && temp > mn;
    }
    if (ok) accept(temp);
  }
}

// K spheres of a list entry (K = 4: the whole entry; K = 2: half of it — the kernels that run at a 72-register budget)
template <bool MOVING, int K, typename P, typename AcceptAt>
__device__ __forceinline__ void sphere_list_trip(P recs, const int (&o)[K], int goff, float frac, const RayCtx& c, HitState& h,
                                                 AcceptAt accept_at) {
  const Ray& r = c.r;
  f4 R0[K], R2[K];
  // the run's base address held in a VGPR the compiler cannot see through: each record address is then ONE v_lshl_add_u32
  // with the scalar offset as an operand, instead of s_lshl + s_add + v_mov (three issue slots)
  P vrecs = recs;
  asm volatile("" : "+v"(vrecs));
#pragma unroll
  for (int k = 0; k < K; k++) R0[k] = vrecs[o[k]];
  if (MOVING) {
#pragma unroll
    for (int k = 0; k < K; k++) R2[k] = vrecs[o[k] + 2];
  }
  SphereEval e[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    V3 center = xyz(R0[k]);
    if (MOVING) center = center + frac * xyz(R2[k]); // sphere.hpp:54-55
    V3 oc = r.o - center;
    float b = dot(oc, r.d);
    float cc = dot(oc, oc) - __builtin_fabsf(R0[k].w);
    e[k] = SphereEval{b, b * b - c.a * cc};
  }
#pragma unroll
  for (int k = 0; k < K; k++) sphere_finish_unordered(e[k], c, PT_TMIN, h, goff + o[k], accept_at(o[k]));
}
template <bool MOVING, int K, typename P, typename AcceptAt>
__device__ __forceinline__ void sphere_list_entry(P recs, i4 o4, int goff, float frac, const RayCtx& c, HitState& h, AcceptAt accept_at) {
  if constexpr (K >= 4) {
    const int o[4] = {o4.x, o4.y, o4.z, o4.w};
    sphere_list_trip<MOVING, 4>(recs, o, goff, frac, c, h, accept_at);
  } else {
    // Equal neighbouring offsets are the list's padding (its last entry repeated: two records never share an offset), and
    // the offsets are scalars: the repeats are skipped by scalar branches.  Short lists are what a ray of a chain-bound
    // frame spends its time on — the 496-hittable scene tests 7 spheres outside the grid (ground, glowing ball, the five
    // big ones) and did so as 16.
    const int a[2] = {o4.x, o4.y}, b[2] = {o4.z, o4.w}, a1[1] = {o4.x}, b1[1] = {o4.z};
    if (o4.y == o4.x) { sphere_list_trip<MOVING, 1>(recs, a1, goff, frac, c, h, accept_at); return; }
    sphere_list_trip<MOVING, 2>(recs, a, goff, frac, c, h, accept_at);
    if (o4.z == o4.y) return;
    if (o4.w == o4.z) { sphere_list_trip<MOVING, 1>(recs, b1, goff, frac, c, h, accept_at); return; }
    sphere_list_trip<MOVING, 2>(recs, b, goff, frac, c, h, accept_at);
  }
}

// dword i of an array of dwords packed four per f4 at `base` (cell table / candidate list of a sphere grid)
__device__ __forceinline__ unsigned int dword_at(lds_f4p base, int i) { return ((const __attribute__((address_space(3))) unsigned int*)base)[i]; }
__device__ __forceinline__ unsigned int dword_at(cst_f4p base, int i) { return ((const __attribute__((address_space(4))) unsigned int*)base)[i]; }
__device__ __forceinline__ unsigned int dword_at(const f4* base, int i) { return ((const unsigned int*)base)[i]; }
__device__ __forceinline__ unsigned int ushort_at(lds_f4p base, int i) { return ((const __attribute__((address_space(3))) unsigned short*)base)[i]; }
__device__ __forceinline__ unsigned int ushort_at(cst_f4p base, int i) { return ((const __attribute__((address_space(4))) unsigned short*)base)[i]; }
__device__ __forceinline__ unsigned int ushort_at(const f4* base, int i) { return ((const unsigned short*)base)[i]; }

// Exact culling for the small spheres of a run (pt_flatten.hpp: build_sphere_grid states why it is exact): every lane walks
// its own ray through a uniform grid (3-D DDA in cell units; t is the ray's own parameter, so it compares with closest
// directly) and tests the spheres listed in the cells it crosses, each with the unordered acceptance rule — the order of
// the tests does not matter and a sphere listed in several cells is simply re-tested.  The walk ends where the next cell's
// entry lies beyond the nearest hit so far.  SIMD shape: the wave steps all walks together (a lane past its last cell idles)
// and, per step, loops to the largest candidate count of the lanes' cells.
// (Round 3 measured the walk with in-kernel counters — `make stamps EXTRA=-DPT_STAMPS_WALK`, tools/stamps.py — and tried a
// split phase that hands the rest of the unfinished walks to idle lanes once <= 32 lanes still walk: DESIGN.md, rejected
// experiments.  The walk is 48 % of a wave-iteration on the 496-hittable scene; 6.3 wave-steps and 17.3 test trips for 2.0
// cells and 4.1 sphere tests per lane.)
template <typename P, typename AcceptAt>
__device__ __forceinline__ void sphere_grid_walk(P recs, P cells, P cand, f4 g0, f4 g1, float frac, int goff, const RayCtx& c,
                                                 HitState& h, AcceptAt accept_at) {
  const Ray& r = c.r;
  const float inv = g0.w, cell = g1.w;
  const int nx = as_i(g1.x), ny = as_i(g1.y), nz = as_i(g1.z);
  const float gx = (r.o.x - g0.x) * inv, gy = (r.o.y - g0.y) * inv, gz = (r.o.z - g0.z) * inv; // origin in cell units
  const float rx = c.yx * cell, ry = c.yy * cell, rz = c.yz * cell;                           // 1 / (direction in cell units)
  // slab clip of the line against the grid box [0, n]
  const float ax = (0.0f - gx) * rx, bx = ((float)nx - gx) * rx;
  const float ay = (0.0f - gy) * ry, by = ((float)ny - gy) * ry;
  const float az = (0.0f - gz) * rz, bz = ((float)nz - gz) * rz;
  const float t_in = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
  const float t_out = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
  const float t0 = __builtin_fmaxf(t_in, 0.0f);
  auto limit = [&]() { const float m = __builtin_fminf(t_out, h.closest); return m + (__builtin_fabsf(m) * 1e-4f + 1e-4f); };
  bool active = c.live && t0 <= limit();
  // entry cell
  const float px = gx + t0 * (r.d.x * inv), py = gy + t0 * (r.d.y * inv), pz = gz + t0 * (r.d.z * inv);
  int ix = min(max((int)__builtin_floorf(px), 0), nx - 1);
  int iy = min(max((int)__builtin_floorf(py), 0), ny - 1);
  int iz = min(max((int)__builtin_floorf(pz), 0), nz - 1);
  const bool fx = r.d.x > 0.0f, fy = r.d.y > 0.0f, fz = r.d.z > 0.0f; // walking towards larger indices?
  float tmx = ((float)(ix + (fx ? 1 : 0)) - gx) * rx; // ray parameter at the next cell boundary, per axis
  float tmy = ((float)(iy + (fy ? 1 : 0)) - gy) * ry;
  float tmz = ((float)(iz + (fz ? 1 : 0)) - gz) * rz;
  const float dtx = __builtin_fabsf(rx), dty = __builtin_fabsf(ry), dtz = __builtin_fabsf(rz);
  const int stx = fx ? 1 : -1, sty = fy ? 1 : -1, stz = fz ? 1 : -1;
  unsigned int hdr = 0;
  if (active) hdr = dword_at(cells, (iz * ny + iy) * nx + ix);
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    PT_WALK_COUNT(2, 1);
    PT_WALK_COUNT(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(active)));
    // the next cell (the axis whose boundary comes first; branch-free) and its header, requested BEFORE this cell's tests:
    // the walk of a lone wave is a chain of dependent LDS reads, this takes one of them off the chain
    const float tn = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
    const bool sx = tmx == tn, sy = !sx & (tmy == tn), sz = !sx & !sy;
    const int jx = ix + (sx ? stx : 0), jy = iy + (sy ? sty : 0), jz = iz + (sz ? stz : 0);
    const bool inside = ((unsigned)jx < (unsigned)nx) & ((unsigned)jy < (unsigned)ny) & ((unsigned)jz < (unsigned)nz);
    unsigned int hdr_next = 0;
    if (active & inside) hdr_next = dword_at(cells, (jz * ny + jy) * nx + jx);
    const int count = (int)(hdr & 255u), first = (int)(hdr >> 8);
    for (int k = 0; __builtin_amdgcn_ballot_w64(k < count) != 0; ++k) {
      PT_WALK_COUNT(6, 1);
      PT_WALK_COUNT(5, __builtin_popcountll(__builtin_amdgcn_ballot_w64(k < count)));
      if (k < count) {
        const unsigned int e = ushort_at(cand, first + k); // sphere index in the run | moving << 15
        const int o = (int)(e & 0x7fffu) * SZ_SPHERE;
        const f4 R0 = recs[o];
        V3 center = xyz(R0);
        if (e & 0x8000u) center = center + frac * xyz(recs[o + 2]); // moving: sphere.hpp:54-55
        V3 oc = r.o - center;
        float b = dot(oc, r.d);
        float cc = dot(oc, oc) - __builtin_fabsf(R0.w);
        sphere_finish_unordered(SphereEval{b, b * b - c.a * cc}, c, PT_TMIN, h, goff + o, accept_at(o));
      }
    }
    // step: the walk ends where the next cell lies outside the grid or begins beyond the nearest hit so far
    active = active & inside & !(tn > limit());
    ix = jx; iy = jy; iz = jz;
    tmx += sx ? dtx : 0.0f; tmy += sy ? dty : 0.0f; tmz += sz ? dtz : 0.0f;
    hdr = active ? hdr_next : 0u;
  }
}

// ---- the walk, round 4: the wave still steps its walks together, but the candidates go through an LDS queue and are tested 64 PAIRS at a time
// What the counters of the walk above say (496-hittable scene, profiles/r03_smoke_walk_counters.json, r04_walk_*): per wave and walk 6.4
// wave-steps and 17.7 test trips for 127 lane-cells and 268 lane-tests — a test trip runs at 15 of 64 lanes, and whenever ONE of them has a
// positive discriminant (a third of the trips) the whole wave walks through the root block: a correctly rounded square root and two
// IEEE divisions, ~75 issue slots.  Instruction counts, from the ISA and SQ_INSTS_*: a wave-step ~48, a trip ~45 + 75 x 0.35.
// Round 4 first tried per-lane cursors (a lane pushes up to two candidates of its cell per round or steps to its next cell, one ballot per
// round): 9.6 rounds of ~125 slots + 4.9 batches — bit-exact, 58 of 64 lanes per batch, and NOT fewer instructions than the walk above
// (SQ_INSTS_VALU + SALU + LDS per sample 196 -> 201): the bookkeeping of a round costs what the divergence did.  This version keeps the cheap
// part of the old shape — the wave-synchronous DDA step and the loop to the largest candidate count — but a trip of that loop only QUEUES
// (lane << 16 | sphere index | moving << 15) per candidate (a 16-bit read, a 32-bit write, the position from the trip's ballot + mbcnt:
// ~17 slots), and whenever 64 pairs are queued the wave tests them together: each lane fetches its pair's ray from the owner lane
// (ds_bpermute) and the sphere's record, evaluates the reference's test (sphere.hpp:59-93) and, for a root in range, lowers the owner's
// slot of a per-wave LDS table with ONE 64-bit atomic minimum on (bits of t) << 32 | record offset: the smallest t wins and, among equal t,
// the FIRST sphere in list order (sphere.hpp:77 needs t < max) — the tie rule of sphere_finish_unordered; the slot starts as the ray's hit
// so far with ITS record offset, so an equal t replaces it iff the holder is a later record (an earlier run's hit, or a big sphere of this
// run listed before the candidate, stays).  A candidate's t does not depend on the running maximum (first root if > min, else second root
// if > min: the second root is never smaller than the first, so a first root that fails `t < max` takes the second down with it), hence
// any order and any batching of the tests gives the scan's result.  A lane reads its slot back before a step looks at the limit: the walk
// still ends where the next cell begins beyond the nearest hit so far — later than with immediate tests when pairs are still queued, never
// earlier, and a later end only adds candidates.  Exactness of the culling itself is unchanged (pt_flatten.hpp: build_sphere_grid).
// Measured on the 496-hittable scene at 1080p x 1024 spp (profiles/r04_walk_*): 450 -> 436 ms, VALU wave-instructions per sample 131.8 ->
// 115.2 (SALU 60.8 -> 65.3, LDS 3.8 -> 6.5), lane utilisation 0.515 -> 0.59, 4.8 batches at 58 of 64 lanes instead of 17.7 trips at 15.
// Two follow-ups measured and NOT kept: the roots of a batch deferred until a lane finds a second positive discriminant or the walk ends
// (the root block then runs 1-2 times per walk instead of 4, but hits no longer end walks early: 438 -> 476 ms), and the same deferral for
// the listed big spheres (a first pass notes positive discriminants per lane, a second runs the roots per lane: VALU per sample 115 ->
// 129, 436 -> 447 ms — with 7 listed spheres the per-lane gather + recomputation costs more than the ~4 shared root blocks it replaces).
// (the triangle pool's per-wave LDS arrays — tri_pool_scan below — declared here because kernels that carry both lend them to this walk)
__device__ __forceinline__ unsigned long long* tri_slots() { __shared__ unsigned long long s[256]; return s; }
#ifndef PT_TRI_ABLATE
#define PT_TRI_ABLATE 0 /* timing experiments only: 1 no grid, 2 no direction maps (every ray streams every band record), 4 no band stage at all (wrong images) */
#endif
#define PT_TRI_QUEUE 192 /* >= 64 PT_TRI_PAIRS - 1 left over + one trip's 64 pushes (the grid's pairs; the band stage's survivors), and >= PT_SQ_CAP */
__device__ __forceinline__ int* tri_queue() { __shared__ int s[4 * PT_TRI_QUEUE]; return s; } // per wave: what waits for the exact test
#ifndef PT_MAX_WAVES_PER_BLOCK
#define PT_MAX_WAVES_PER_BLOCK 4
#endif
#define PT_SQ_CAP 128 /* 63 left over + one trip's pushes: 512 B per wave, + 512 B of slots: 4 KB of STATIC LDS per 256-thread workgroup on top of the blob image (launch_render budgets it: kQueuedWalkStaticLds) */
__device__ __forceinline__ unsigned int* sphere_queue() { __shared__ unsigned int s[PT_MAX_WAVES_PER_BLOCK * PT_SQ_CAP]; return s; }
__device__ __forceinline__ unsigned long long* sphere_slots() { __shared__ unsigned long long s[PT_MAX_WAVES_PER_BLOCK * 64]; return s; }

// slot / q: this wave's 64 slots and its queue of PT_SQ_CAP entries (kernels with a triangle pool lend the pool's arrays: the two scans
// never overlap)
template <typename P, typename AcceptAt>
__device__ __forceinline__ void sphere_grid_walk_queued(P recs, P cells, P cand, f4 g0, f4 g1, float frac, bool moves, int goff, const RayCtx& c,
                                                        HitState& h, unsigned long long* const slot, unsigned int* const q, AcceptAt accept_at) {
  const Ray& r = c.r;
  const int lane = threadIdx.x & 63;
  const float inv = g0.w, cell = g1.w;
  const int nx = as_i(g1.x), ny = as_i(g1.y), nz = as_i(g1.z);
  const float gx = (r.o.x - g0.x) * inv, gy = (r.o.y - g0.y) * inv, gz = (r.o.z - g0.z) * inv; // origin in cell units
  const float rx = c.yx * cell, ry = c.yy * cell, rz = c.yz * cell;                           // 1 / (direction in cell units)
  // slab clip of the line against the grid box [0, n]
  const float ax = (0.0f - gx) * rx, bx = ((float)nx - gx) * rx;
  const float ay = (0.0f - gy) * ry, by = ((float)ny - gy) * ry;
  const float az = (0.0f - gz) * rz, bz = ((float)nz - gz) * rz;
  const float t_in = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
  const float t_out = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
  const float t0 = __builtin_fmaxf(t_in, 0.0f);
  float closest = h.closest; // the slot's t as last read back
  auto limit = [&]() { const float m = __builtin_fminf(t_out, closest); return m + (__builtin_fabsf(m) * 1e-4f + 1e-4f); };
  bool active = c.live && t0 <= limit();
  // entry cell
  const float px = gx + t0 * (r.d.x * inv), py = gy + t0 * (r.d.y * inv), pz = gz + t0 * (r.d.z * inv);
  int ix = min(max((int)__builtin_floorf(px), 0), nx - 1);
  int iy = min(max((int)__builtin_floorf(py), 0), ny - 1);
  int iz = min(max((int)__builtin_floorf(pz), 0), nz - 1);
  const bool fx = r.d.x > 0.0f, fy = r.d.y > 0.0f, fz = r.d.z > 0.0f; // walking towards larger indices?
  float tmx = ((float)(ix + (fx ? 1 : 0)) - gx) * rx; // ray parameter at the next cell boundary, per axis
  float tmy = ((float)(iy + (fy ? 1 : 0)) - gy) * ry;
  float tmz = ((float)(iz + (fz ? 1 : 0)) - gz) * rz;
  const float dtx = __builtin_fabsf(rx), dty = __builtin_fabsf(ry), dtz = __builtin_fabsf(rz);
  const int stx = fx ? 1 : -1, sty = fy ? 1 : -1, stz = fz ? 1 : -1;
  const unsigned long long key0 = ((unsigned long long)(unsigned int)as_i(h.closest) << 32) | (unsigned long long)(unsigned int)(h.hit >= 0 ? hit_off(h.hit) : 0);
  slot[lane] = key0;
  unsigned int hdr = 0;
  if (active) hdr = dword_at(cells, (iz * ny + iy) * nx + ix);
  int qn = 0;         // pairs queued (wave-uniform)
  bool stale = false; // (wave-uniform) a batch ran since `closest` was last read back
  // 64 pairs (the top of the queue) through the reference's test
  auto test_batch = [&]() {
    __builtin_amdgcn_wave_barrier();
    const int n = min(qn, 64);
    const bool on = lane < n;
    const unsigned int e = q[qn - n + (on ? lane : 0)];
    const int src = (int)(e >> 16);
    PT_WALK_COUNT(6, 1);
    PT_WALK_COUNT(5, n);
    // the pair's ray, from its owner's registers
    const V3 o = mk(__shfl(r.o.x, src, 64), __shfl(r.o.y, src, 64), __shfl(r.o.z, src, 64));
    const V3 d = mk(__shfl(r.d.x, src, 64), __shfl(r.d.y, src, 64), __shfl(r.d.z, src, 64));
    const float a = __shfl(c.a, src, 64);
    float fr = 0.0f;
    if (moves) fr = __shfl(frac, src, 64); // (wave-uniform: the run has moving spheres)
    const int so = (int)(e & 0x7fffu) * SZ_SPHERE;
    const f4 R0 = recs[so];
    V3 center = xyz(R0);
    if (moves) { if (e & 0x8000u) center = center + fr * xyz(recs[so + 2]); } // moving: sphere.hpp:54-55
    const V3 oc = o - center;
    const float b = dot(oc, d);
    const float cc = dot(oc, oc) - __builtin_fabsf(R0.w);
    const float disc = b * b - a * cc;
    if (on && disc > 0) {
      const float sq = sqrt_rn(disc);
      float t = (-b - sq) / a;
      if (!(t > PT_TMIN)) t = (-b + sq) / a;
      if (t > PT_TMIN) atomicMin(&slot[src], ((unsigned long long)(unsigned int)as_i(t) << 32) | (unsigned long long)(unsigned int)(goff + so));
    }
    stale = true;
    qn -= n;
    __builtin_amdgcn_wave_barrier();
  };
  while (__builtin_amdgcn_ballot_w64(active) != 0) {
    PT_WALK_COUNT(2, 1);
    PT_WALK_COUNT(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(active)));
    // the next cell (the axis whose boundary comes first; branch-free) and its header, requested BEFORE this cell's candidates are queued
    const float tn = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
    const bool sx = tmx == tn, sy = !sx & (tmy == tn), sz = !sx & !sy;
    const int jx = ix + (sx ? stx : 0), jy = iy + (sy ? sty : 0), jz = iz + (sz ? stz : 0);
    const bool inside = ((unsigned)jx < (unsigned)nx) & ((unsigned)jy < (unsigned)ny) & ((unsigned)jz < (unsigned)nz);
    unsigned int hdr_next = 0;
    if (active & inside) hdr_next = dword_at(cells, (jz * ny + jy) * nx + jx);
    const int count = (int)(hdr & 255u), first = (int)(hdr >> 8);
    for (int k = 0;; ++k) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(k < count);
      if (m == 0) break;
      if (k < count) {
        const unsigned int e = ushort_at(cand, first + k); // sphere index in the run | moving << 15
        const int at = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
        q[at] = ((unsigned int)lane << 16) | e;
      }
      qn += __builtin_popcountll(m);
      if (qn >= 64) test_batch();
    }
    if (stale) { closest = as_f((int)(unsigned int)(slot[lane] >> 32)); stale = false; }
    // step: the walk ends where the next cell lies outside the grid or begins beyond the nearest hit so far
    active = active & inside & !(tn > limit());
    ix = jx; iy = jy; iz = jz;
    tmx += sx ? dtx : 0.0f; tmy += sy ? dty : 0.0f; tmz += sz ? dtz : 0.0f;
    hdr = active ? hdr_next : 0u;
  }
  if (qn > 0) test_batch();
  const unsigned long long kf = slot[lane];
  if (kf != key0) accept_at((int)(unsigned int)(kf & 0xffffffffull) - goff)(as_f((int)(unsigned int)(kf >> 32)));
}

// The whole run (recs = its first record, at blob offset goff).
// GRID = false: the kernel does not carry the grid walk (the streaming kernel: its register budget belongs to the triangle
// loop); a run with a grid is then scanned through its full lists.
template <int K, int GRID, bool TRIPOOL, typename P, typename AcceptAt>
__device__ __forceinline__ void sphere_scan(P recs, cst_f4p cblob, int n, int goff, const RayCtx& c, HitState& h, AcceptAt accept_at) {
  const f4 aux = cblob[goff - 1];
  // (aux.z: entries of the static list, the absorbed spheres of later runs — flags >> 8 of them — included: pt_flatten.hpp "absorbed sphere runs")
  const int flags = as_i(aux.w), ns = as_i(aux.z), nm = n - (ns - (flags >> 8));
  if (!(flags & 1)) { // moving spheres with different shutter intervals: one sphere at a time in list order, fraction memoised
    TimeFrac tf = time_frac_none();
    for (int i = 0, off = 0; i < n; ++i, off += SZ_SPHERE) sphere_roots(recs, off, c, PT_TMIN, h.closest, true, tf, accept_at(off));
    return;
  }
  const int qs = (ns + 3) >> 2, qm = (nm + 3) >> 2;
  int lists_off = goff - 1 - qs - qm; // blob offset of the run's static list
  float frac = 0.0f;
  if (flags & 2) frac = (c.r.tm - aux.x) / (aux.y - aux.x); // (time - time0) / (time1 - time0)  sphere.hpp:54
  int q_static = qs, q_moving = qm;
  bool walk = false;
  int w_cell = 0, w_cand = 0;
  f4 wg0 = aux, wg1 = aux;
  if (flags & 4) lists_off -= 4; // (a grid's four header records sit between the lists and aux)
  if (GRID && (flags & 4)) { // the run has a grid for its small spheres
    const f4 g0 = cblob[goff - 5], g1 = cblob[goff - 4], g2 = cblob[goff - 3], g3 = cblob[goff - 2];
    const V3 dc = c.r.o - xyz(g2);
    // the walk is exact for a regular ray that starts within rlimit of the grid and (if something moves) whose time lies in
    // the run's shutter interval, so that centres stay between centre0 and centre1; one live lane outside -> full lists
    const bool ok = c.reg && dot(dc, dc) <= g2.w && (!(flags & 2) || (c.r.tm >= aux.x && c.r.tm <= aux.y));
#if defined(PT_STAMPS) && !defined(PT_STAMPS_WALK)
    { // diagnostic build: how often a wave may walk the grid, and why not (far origin / irregular / shutter)
      const unsigned long long bad = __builtin_amdgcn_ballot_w64(c.live && !ok);
      const unsigned long long far = __builtin_amdgcn_ballot_w64(c.live && !(dot(dc, dc) <= g2.w));
      if ((threadIdx.x & 63) == 0) { atomicAdd(&g_stamps[4], 1ull); if (bad) atomicAdd(&g_stamps[5], 1ull); if (far) atomicAdd(&g_stamps[6], 1ull);
                                     atomicAdd(&g_stamps[7], (unsigned long long)__builtin_popcountll(bad)); }
    }
#endif
    if (__builtin_amdgcn_ballot_w64(c.live && !ok) == 0) {
      const int n_cell = as_i(g3.x), n_cand = as_i(g3.y), qbs = as_i(g3.z), qbm = as_i(g3.w);
      const int big_off = lists_off - qbs - qbm, cand_off = big_off - n_cand, cell_off = cand_off - n_cell;
      walk = true; w_cell = cell_off - goff; w_cand = cand_off - goff; wg0 = g0; wg1 = g1;
      lists_off = big_off; q_static = qbs; q_moving = qbm; // only the spheres that are not in the grid go through the lists
    }
  }
  const cst_i4p lists = (cst_i4p)(cblob + lists_off);
  // the next list entry is requested while the current one is processed (reading one entry past a list lands on the next
  // list or on the records behind it: valid memory, never used)
  i4 cur = lists[0];
  for (int q = 0; q < q_static; ++q) {
    const i4 nxt = lists[q + 1];
    sphere_list_entry<false, K>(recs, cur, goff, 0.0f, c, h, accept_at);
    cur = nxt;
  }
  for (int q = 0; q < q_moving; ++q) {
    const i4 nxt = lists[q_static + q + 1];
    sphere_list_entry<true, K>(recs, cur, goff, frac, c, h, accept_at);
    cur = nxt;
  }
  // the walk comes after the big spheres (any order gives the same result: the tie rule is explicit): a ground hit found
  // first ends the walks of the rays that go down where they reach it
#ifdef PT_STAMPS_WALK
  const unsigned long long walk_t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#endif
  // GRID = 1: the wave-synchronous walk (every lane tests its own candidate in place); GRID = 2: the queued walk (pairs through an LDS
  // queue, 64 per batch).  Which one a launch takes is the launcher's choice (pt_render.hip: launch_render): measured on the 496-hittable
  // scene, old / queued kernel ms — 1080p x 256 spp shards 0/1, 0/2, 0/4, 0/8: 114.6 / 113.4, 115.1 / 112.9, 95.4 / 91.6, 62.5 / 60.0; 4K x 256 spp
  // shards 0/1 ... 0/16: 351.8 / 388.8, 211.0 / 228.0, 141.4 / 149.8, 108.5 / 103.1, 71.8 / 61.6; config 1 (8 lanes per wave): 18.0 / 19.4.  The
  // queue pays where walks diverge or the launch is bound by its chains; coherent, throughput-bound frames keep the in-place test.
  // (Both in ONE kernel, chosen per trip by how many lanes it has, was measured too: the second path's registers cost 12 % on every frame.)
  if constexpr (GRID == 1) { if (walk) sphere_grid_walk(recs, recs + w_cell, recs + w_cand, wg0, wg1, frac, goff, c, h, accept_at); }
  if constexpr (GRID == 2) {
    if (walk) {
      unsigned long long* slot;
      unsigned int* q;
      if constexpr (TRIPOOL) { slot = tri_slots() + (threadIdx.x & ~63); q = (unsigned int*)(tri_queue() + (threadIdx.x >> 6) * PT_TRI_QUEUE); }
      else { slot = sphere_slots() + (threadIdx.x & ~63); q = sphere_queue() + (threadIdx.x >> 6) * PT_SQ_CAP; }
      sphere_grid_walk_queued(recs, recs + w_cell, recs + w_cand, wg0, wg1, frac, (flags & 2) != 0, goff, c, h, slot, q, accept_at);
    }
  }
#ifdef PT_STAMPS_WALK
  asm volatile("" ::"v"(h.closest), "v"(h.hit));
  __builtin_amdgcn_sched_barrier(0);
  if (GRID && walk) { PT_WALK_COUNT(0, __builtin_amdgcn_s_memtime() - walk_t0); PT_WALK_COUNT(1, 1); }
#endif
}

// n records of one kind at recs[0..): record i is blob offset goff + i*size.  `recs` is either the resident
// blob (LDS or scalar-cached) advanced to the run, or an LDS tile of a streamed run.
// TRIP / TTRIP: how many spheres / triangles a trip of those loops evaluates together (see "several records per trip" below).
// Every resident kernel contains all the loops, and the ones that run at a 72-register budget (7 waves) pay for a wide
// triangle loop with spills around EVERY scan, triangles or not: they take TRIP = 1; the cooperative and the streaming
// kernels, which have registers to spare, take 2 spheres / 4 triangles.
// WHOLE: recs[0..n) is a whole run (its aux records sit in front of it at cblob[goff - 1]); false for an LDS tile of a
// streamed run, which takes the spheres one at a time in list order.
// BADOUEL: the kernel also knows Badouel-strategy triangle runs (DK_TRI_B).
// ---- consecutive runs of rects and boxes, culled exactly ("slab pool") ----------------------------------------------------
// The straight-line scans test 19 issue slots per rect and 6 x 19 per box for EVERY record although a ray's line passes
// through few of them.  For every maximal stretch of consecutive rect / box runs (>= 2 hittables, fast_ok scene) the
// flattener adds a pool table (pt_flatten.hpp): per hittable a slab entry (lo, hi: a rect is a box of no thickness) and an
// exact entry.  Each entry first gets a conservative slab test in cheap arithmetic (~30 slots): the ray's parameter
// interval inside the box, inflated by the rounding the reference itself can commit, which yields a lower bound L on the t
// of ANY side of that hittable the reference could accept, or "cannot be hit".  The lane keeps its three smallest
// (L, entry) keys; then, nearest first and only while L <= closest, the entry's sides are tested EXACTLY (the same
// rect_side_cmpx instructions as the straight-line box scan, records fetched per lane).  Everything skipped provably
// fails the reference's own comparisons; every accepted t is computed by the same instructions as in the straight-line scan.
//
// Why the slab test is conservative (u = 2^-24).  A side on plane K of axis k is accepted by the reference
// (rectangle.hpp:34-43, also through box.hpp:29-50) only if t = RN(RN(K - o_k) / d_k) >= min and, for both in-plane axes a,
// A = RN(o_a + RN(t d_a)) lies in [lo_a, hi_a].  The two roundings move A by at most u (|t d_a| + |A|) <= u (2 B + |o_a|)
// (B = the largest |coordinate| in the pool), so in real arithmetic t lies in the ray's parameter interval over
// [lo_a - e, hi_a + e], e = u (2 B + |o_a|); for the plane's own axis t is within 2u (relative) of an interval end, i.e.
// within 2u |K - o_k| <= 2u (B + |o_k|) in position.  The interval ends are computed from an origin moved outwards by
// S = 9u |o| + 7u B per axis, as fma(lo, y, -(o + S) y) and fma(hi, y, -(o - S) y) with y = RN(1 / d) (two instructions per
// axis and entry fewer than subtract-then-multiply).  What S has to cover, in position: e (or the 2u (B + |o|) of the
// plane's own axis); ulp(o +- S)/2 <= u (|o| + S) lost when o +- S is rounded; u |o +- S| for rounding the product
// (o +- S) y; and the roundings that are relative to the interval end — y itself and the fma, 3u |lo - o| <= 3u (B + |o|)
// with room: together u (5 B + 6 |o|) (1 + ...) < S.  So an accepted t satisfies max_c lo_c <= t <= min_c hi_c over the
// computed ends, and t >= min: L = max(entry, min) is a lower bound of it and the hittable is a candidate iff L <= exit.
// (Regular rays on a fast_ok scene only: nothing here overflows or is NaN.  The margins matter little: a ray leaving a
// box's face keeps that box as a candidate when S / |d_k| reaches min — the proof in slab_chunk_pass drops those.)
//
// Testing out of list order needs the tie rule spelled out, as for the sphere lists: the sequential scan accepts a side on
// t <= closest, i.e. among equal t the LAST in list order wins.  A candidate therefore compares against closest when the
// current holder sits earlier in the list (or is this hittable), and against the next float below closest
// (t <= nextbelow(closest)  <=>  t < closest) when the holder is a later hittable of this pool that was tested first
// (records stay in list order in the blob: offsets compare like list positions).
// A rect's exact entry is a box whose other five sides cannot pass: X0 = (lo, hit id), X1 = (hi, -) with the plane in `lo`
// and -inf in `hi` on the rect's own axis — the "hi" side of that axis has t = NaN (fails the ordered min <= t), the four
// sides of the other axes have an empty in-plane interval [K, -inf], and the "lo" side IS the rect's test, instruction for
// instruction (rect_side_cmpx<AX> with the same operands as rect_fast<AX>).  A rect hit carries junk side bits: nothing
// reads them for DK_RECT.
// Keys: the float L with its low four bits replaced by the entry's index in the chunk of 16 (L >= min > 0, so the keys
// order like floats; truncation only lowers L); +inf = none.  More than three live candidates: another pass over the
// chunk for the lanes concerned, restricted to keys above the last one handled.
template <bool FILTER, typename P>
__device__ __forceinline__ bool slab_chunk_pass(P xrecs, P slrecs, cst_f4p srecs, int cn, float bmax, const RayCtx& c, V3 om, V3 op, float& kdone,
                                                HitState& h) {
  const float none = PT_INF;
  float k1 = none, k2 = none, k3 = none;
  // two entries per trip (the table is padded to an even count with an all-NaN entry: `L <= NaN` is false), one
  // s_load_dwordx16 for both; wave-uniform, the bounds are SGPR operands
  typedef float f16v __attribute__((ext_vector_type(16)));
  typedef const __attribute__((address_space(4))) f16v* cst_f16p;
  const cst_f16p pairs = (cst_f16p)srecs;
  auto entry = [&](float lx, float ly, float lz, float hx, float hy, float hz, int j) {
    // (x and y through the packed fp32 pipe — v_pk_add_f32 / v_pk_mul_f32, 8 instead of 12 instructions — measured 5 % SLOWER
    // on the same box: a packed fp32 instruction occupies the issue port for more than one slot)
    const float ax = __builtin_fmaf(lx, c.yx, om.x), bx = __builtin_fmaf(hx, c.yx, op.x); // om = -(o + S) y, op = -(o - S) y
    const float ay = __builtin_fmaf(ly, c.yy, om.y), by = __builtin_fmaf(hy, c.yy, op.y);
    const float az = __builtin_fmaf(lz, c.yz, om.z), bz = __builtin_fmaf(hz, c.yz, op.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    const float L = __builtin_fmaxf(tn, PT_TMIN);
    float key;
    asm("v_bfi_b32 %0, 15, %1, %2" : "=v"(key) : "s"(j), "v"(L)); // (j & 15) | (L & ~15)
    bool cand = L <= tf;
    if (FILTER) cand = cand & (key > kdone);
    const float kk = cand ? key : none;
    // sorted insert into (k1 <= k2 <= k3), in place, as unsigned integers (positive floats and +inf order the same way; the
    // float min / med3 would first canonicalise their operands: two more instructions per entry)
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k3) : "v"(k2), "v"(kk));
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k2) : "v"(k1), "v"(kk));
    asm("v_min_u32_e32 %0, %0, %1" : "+v"(k1) : "v"(kk));
  };
  // (four entries per trip of this scalar loop — two s_load_dwordx16 in flight — measured 2.3 % SLOWER on the headline scene, 150.2
  // against 146.9 ms, three alternating runs each: profiles/r03_ab_slab_unroll4.log)
  for (int j = 0; j < cn; j += 2) {
    const f16v e = pairs[j >> 1];
    entry(e[0], e[1], e[2], e[4], e[5], e[6], j);
    entry(e[8], e[9], e[10], e[12], e[13], e[14], j + 1);
  }
  // Every trip of the loop consumes the nearest key of every lane that still has a live one (tested exactly, or dropped by
  // the proof below); a key beyond closest ends the lane's scan (keys ascend).  A lane that consumed all three keys it
  // could hold may have a fourth candidate: it asks for another pass (`more`).
  const bool had3 = k3 < none;
#pragma unroll 1
  for (;;) {
    // A ray that leaves a face of a box has that box as its nearest candidate whenever the slab margins reach `min` along the
    // face's axis (a few per cent of the lanes: nearly every wave), and the exact test then rejects all six sides: a whole
    // trip for nothing.  For nearest keys with L == min (origin inside, on or next to the box) a proof is tried first.
    // With the UNSHIFTED interval ends lw_c <= up_c of the three axes (relative error <= 3u each) and W_k >= the shifted
    // upper end of axis k (up_k + S_k |y_k|): if for some axis k
    //   (1) up_k (1 + 2^-20) < min: both planes of axis k have t < min — rejected;
    //   (2) for both other axes a: lw_a (1 + 2^-20) < min — the near plane of a has t < min — and
    //       up_a (1 - 2^-20) > W_k (1 + 2^-20): the far plane of a has a t beyond everything the slab bound admits for a
    //       side whose in-plane coordinate k is inside [lo_k, hi_k] (accepted t <= shifted upper end of EVERY axis, above);
    // then no side of the box can be accepted and the key is dropped without a trip: the ray leaves through the far plane
    // of axis k before it reaches any other plane of the box.
    {
      bool inside = (k1 < none) & ((as_i(k1) & ~15) == (as_i(PT_TMIN) & ~15)) & c.live;
#ifdef PT_NO_POOL_PROOF
      inside = false; // A/B build
#endif
      if (__builtin_amdgcn_ballot_w64(inside)) {
        const int offs = inside ? (as_i(k1) & 15) * 2 : 0;
        const f4 S0 = slrecs[offs], S1 = slrecs[offs + 1]; // the slab entry: true bounds, a rect's plane in both
        const float e = 0x1p-20f, tlo = PT_TMIN * (1.0f - 0x1p-20f);
        auto ends = [&](float o, float y, float lo, float hi, float& lw, float& us, float& ws) { // returns (1) for this axis
          const float a = (lo - o) * y, b = (hi - o) * y;
          lw = __builtin_fminf(a, b);
          const float up = __builtin_fmaxf(a, b);
          // slab_pool's shift S = 9u |o| + 7u B as it can come out of RN(o +- S) and of rounding (o +- S) y: up to
          // ulp(o +- S)/2 <= u (|o| + S) and u |o +- S| more; the fma's own rounding is in the 2^-20 below
          const float S = __builtin_fmaf(__builtin_fabsf(o), 0x1.7p-21f, 0x1.dp-22f * bmax); // 11.5u |o| + 7.25u B
          const float w = __builtin_fmaf(__builtin_fabsf(y), S, up);
          us = __builtin_fmaf(__builtin_fabsf(up), -e, up); // up lowered
          ws = __builtin_fmaf(__builtin_fabsf(w), e, w);    // W raised
          return __builtin_fmaf(__builtin_fabsf(up), e, up) < PT_TMIN;
        };
        float lwx, lwy, lwz, usx, usy, usz, wsx, wsy, wsz;
        const bool ownx = ends(c.r.o.x, c.yx, S0.x, S1.x, lwx, usx, wsx);
        const bool owny = ends(c.r.o.y, c.yy, S0.y, S1.y, lwy, usy, wsy);
        const bool ownz = ends(c.r.o.z, c.yz, S0.z, S1.z, lwz, usz, wsz);
        const float lwm = __builtin_fmaxf(__builtin_fmaxf(lwx, lwy), lwz);
        const bool near_behind = __builtin_fmaf(__builtin_fabsf(lwm), e, lwm) < PT_TMIN;
        const bool kx = ownx & (usy > wsx) & (usz > wsx);
        const bool ky = owny & (usx > wsy) & (usz > wsy);
        const bool kz = ownz & (usx > wsz) & (usy > wsz);
        const bool gone = inside & near_behind & (kx | ky | kz);
        kdone = gone ? k1 : kdone;
        k1 = gone ? k2 : k1; k2 = gone ? k3 : k2; k3 = gone ? none : k3;
      }
    }
    const bool active = (k1 < none) & (as_f(as_i(k1) & ~15) <= h.closest) & c.live;
    if (!__builtin_amdgcn_ballot_w64(active)) break;
#ifdef PT_STAMPS_POOL
    { // diagnostic build (make stamps EXTRA=-DPT_STAMPS_POOL): trips per pool scan, lanes busy per trip, repeated passes
      const unsigned long long busy = __builtin_amdgcn_ballot_w64(active);
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_stamps[5], 1ull);
        atomicAdd(&g_stamps[7], (unsigned long long)__builtin_popcountll(busy));
        if (FILTER) atomicAdd(&g_stamps[6], 1ull);
      }
    }
#endif
    const int offl = active ? (as_i(k1) & 15) * 2 : 0;
    const f4 X0 = xrecs[offl], X1 = xrecs[offl + 1];
    if (active) {
      const unsigned long long exec_now = __builtin_amdgcn_ballot_w64(true);
      int hit_base = as_i(X0.w);
      // (a later holder that is a SPHERE — an absorbed run's, tested through an earlier run's lists — is strict itself: in list order this
      // hittable takes an equal t first and the sphere then fails t < max, so the candidate keeps the non-strict comparison)
      const bool holder_later = ((h.hit >= 0) & (hit_off(h.hit) > hit_off(hit_base))) && hit_kind(h.hit) != DK_SPHERE;
      float cl = holder_later ? as_f(as_i(h.closest) - 1) : h.closest;
      int hit_now = h.hit;
      box_cmpx(X0, X1, c, exec_now, hit_base, cl, hit_now);
      h.closest = hit_now != h.hit ? cl : h.closest;
      h.hit = hit_now;
      kdone = k1;
    }
    k1 = active ? k2 : k1; k2 = active ? k3 : k2; k3 = active ? none : k3;
  }
  return had3 & !(k1 < none) & c.live;
}

// pool table at blob[pool_off]: n slab entries (2 f4 each; padded to an even count), then n exact entries (2 f4 each)
template <typename P>
__device__ __forceinline__ void slab_pool(P blob, cst_f4p cblob, int pool_off, int n, float bmax, const RayCtx& c, HitState& h) {
  const float kP = 0x1.2p-21f;           // 9u
  const float bP = 0x1.cp-22f * bmax;    // 7u B
  const V3 pp = mk(__builtin_fmaf(__builtin_fabsf(c.r.o.x), kP, bP), __builtin_fmaf(__builtin_fabsf(c.r.o.y), kP, bP),
                   __builtin_fmaf(__builtin_fabsf(c.r.o.z), kP, bP));
  const V3 yv = mk(c.yx, c.yy, c.yz);
  const V3 om = (mk(0.0f, 0.0f, 0.0f) - (c.r.o + pp)) * yv, op = (mk(0.0f, 0.0f, 0.0f) - (c.r.o - pp)) * yv; // the fma addends of the slab test
#ifdef PT_STAMPS_POOL
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[4], 1ull);
#endif
  for (int base = 0; base < n; base += 16) {
    const int cn = n - base < 16 ? n - base : 16;
    const P xrecs = blob + pool_off + 2 * (n + (n & 1)) + 2 * base;
    const cst_f4p srecs = cblob + pool_off + 2 * base; // the slab entries through the scalar cache (wave-uniform reads) ...
    const P slrecs = blob + pool_off + 2 * base;       // ... and where the lanes read them individually
    float kdone = 0.0f;
    bool more = slab_chunk_pass<false>(xrecs, slrecs, srecs, cn, bmax, c, om, op, kdone, h);
    while (__builtin_amdgcn_ballot_w64(more)) {
      if (!more) kdone = PT_INF; // lanes that are done: no key passes the filter
      more = slab_chunk_pass<true>(xrecs, slrecs, srecs, cn, bmax, c, om, op, kdone, h);
    }
  }
}

// ---- a long run of Moller-Trumbore triangles, culled exactly ("triangle pool") ----------------------------------------------
// pt_tripool.hpp states what the tables are and proves that the two candidate sources below — the fine grid for the triangles a
// ray does not graze, the direction map for the ones it does (slivers included) — contain every triangle the reference's scan
// could accept.  Every candidate runs the reference's own test (tri_param = tri_eval + the second half) with the scan's
// acceptance spelled out for any order: min <= t <= closest, and an equal t replaces the holder unless the holder is a LATER
// record (triangle.hpp:91 accepts t == max: the last in list order wins; records keep list order in the blob).  Re-testing a
// triangle is therefore a no-op, and a triangle found by both sources is harmless.
// The tables live in a buffer of their own in global memory (`pool`: up to gigabytes — this path belongs to scenes far beyond LDS).
#ifdef PT_STAMPS_TRI
#define PT_TRI_COUNT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((threadIdx.x & 63) == 0) atomicAdd(&g_tri[i], v_); } while (0)
#else
#define PT_TRI_COUNT(i, v) do { } while (0)
#endif
// bits set in a 4-bit per-lane mask, summed over the wave (diagnostic counters only)
#define PT_TRI_WAVE_BITS(m) (__builtin_popcountll(__builtin_amdgcn_ballot_w64(((m) & 1u) != 0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(((m) & 2u) != 0)) + \
                             __builtin_popcountll(__builtin_amdgcn_ballot_w64(((m) & 4u) != 0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(((m) & 8u) != 0)))

__device__ __forceinline__ unsigned int gdword(glb_f4p pool, unsigned int base_f4, unsigned int i) { return ((const __attribute__((address_space(1))) unsigned int*)(pool + base_f4))[i]; }
// a dword of a table that is streamed (a direction map's list, a cell's candidate list).  Non-temporal loads here — so that the streams
// would not push the re-read tables out of an XCD's 4 MB L2 — measured 30 % SLOWER (1080p x 8 spp: 430 against 333 ms, three alternating
// runs, profiles/r05_ab_tripool.txt): neighbouring rays read the same bins and cells, the lists ARE re-read.  PT_NT_LOADS restores the experiment.
__device__ __forceinline__ unsigned int gdword_stream(glb_f4p pool, unsigned int base_f4, unsigned int i) {
#ifdef PT_NT_LOADS
  return __builtin_nontemporal_load(((const __attribute__((address_space(1))) unsigned int*)(pool + base_f4)) + i);
#else
  return gdword(pool, base_f4, i);
#endif
}
__device__ __forceinline__ unsigned int sdword(glb_f4p pool, unsigned int base_f4, unsigned int i) { // wave-uniform address: through the scalar cache
  return ((const __attribute__((address_space(4))) unsigned int*)(unsigned long long)(pool + base_f4))[i];
}

// returns false when some live lane's ray is outside what the pool is exact for (irregular, or its origin beyond rlimit): the
// caller then scans the whole run.
//
// Round 5: TWO SIMD SHAPES, one per candidate source (rounds 3-4 took every ray through grid, band levels and always list one
// at a time, 12 500 wave-instructions per ray; profiles/r04_tripool_counters.txt).
//   (1) GRID, EVERY LANE ITS OWN RAY.  The cells are fine (a few tens of triangles; listed by box AND plane slab) and a ray ends
//       at its first hit after a handful of them, so a per-lane 3-D DDA is cheap; what the lanes find in their cells is not tested
//       lane by lane (a gather and ~100 instructions at a few lanes) but as PAIRS: per wave-step the lanes' candidate ranges are
//       concatenated (a wave scan) and the wave runs the reference's test on 64 (ray, triangle) pairs per trip, each lane fetching
//       its pair's ray from the owner lane (ds_bpermute) and the triangle's records from a Morton-ordered copy of the run (a cell's
//       triangles are neighbours in it).  A hit lowers the owner's slot of a per-wave LDS table with ONE 64-bit atomic minimum:
//       key = (bits of t) << 32 | (0xffffffff - record offset), so that the smallest t wins and, among equal t, the LAST record
//       in list order — the scan's own acceptance; the slot starts as the ray's hit so far (earlier runs have smaller offsets: an
//       equal t loses to any triangle, as in the scan).  Every candidate's t is independent of the running maximum, so the
//       order and the batching of the tests do not matter; a lane reads its slot back before a step looks at the limit.
//   (2) DIRECTION MAP, ONE RAY AT A TIME, ITS CANDIDATES ACROSS THE 64 LANES.  The grazing candidates of a ray are a long list
//       (1 000 - 3 000 entries: the bin of its direction in the map of its rho class) of which a handful survive: the wave takes
//       its live rays one after the other (the ray's context read from its lane with v_readlane: scalar operands from then on),
//       streams the bin's entries 256 per trip (coalesced), gathers each entry's 16-byte compressed band record and runs
//       stage 1 — the band test alone, in integers; survivors are queued and meet stage 2 — the noise-radius filter — 64 at a
//       time, and what survives that is queued again and runs the reference's test 64 at a time, into the same slots.
//       The grid goes first: the nearest hit it finds is nearly always the final one, and nothing here depends on that.
typedef short short2_t __attribute__((ext_vector_type(2)));
#ifndef PT_BAND_PER
#define PT_BAND_PER 2 /* entries a lane takes per trip of the direction-map loop (tri_band_one_ray) */
#endif
#define PT_TRI_BQUEUE (64 + 64 * PT_BAND_PER)
// per wave (63 + 128 entries are needed): band candidates past the integer band test — the compressed record itself (stage 2 reads it from
// here: a second gather of the record cost a TA cycle per lane, and this kernel is bound by those) and the candidate's position
__device__ __forceinline__ f4* tri_bqueue() { __shared__ f4 s[4 * PT_TRI_BQUEUE]; return s; }
__device__ __forceinline__ int* tri_bqueue_idx() { __shared__ int s[4 * PT_TRI_BQUEUE]; return s; }
__device__ __forceinline__ unsigned long long tri_key(float t, int off) {
  return ((unsigned long long)(unsigned int)as_i(t) << 32) | (unsigned long long)(0xffffffffu - (unsigned int)off);
}
// triangle.hpp:58-89: everything but the range test on t; false = rejected by |a|, u, v or u + v
__device__ __forceinline__ bool tri_param(f4 R0, f4 R1, f4 R2, const Ray& r, float& t) {
  const TriEval e = tri_eval(R0, R1, R2, r);
  if (!e.pass) return false;
  const float a_abs = __builtin_fabsf(e.a);
  const bool a_pos = e.a > 0.0f;
  const V3 edge1 = xyz(R1), edge2 = xyz(R2);
  const V3 s = r.o - xyz(R0);
  const V3 q = cross(s, edge1);
  const float v = dot(r.d, q);
  const bool v_pos = v > 0.0f;
  if ((v_pos != a_pos) | (__builtin_fabsf(e.u + v) > a_abs)) return false;
  t = dot(edge2, q) / e.a;
  return true;
}
__device__ __forceinline__ float rl_f(float v, int src) { return as_f(__builtin_amdgcn_readlane(as_i(v), src)); }

// the bin of a direction in a direction map of R bins per face edge (pt_tripool.hpp: build_dir_map lists by exactly this rule)
__device__ __forceinline__ void tri_dir_cell(V3 d, int R, int& k, int& ci, int& cj) {
  const float adx = __builtin_fabsf(d.x), ady = __builtin_fabsf(d.y), adz = __builtin_fabsf(d.z);
  // face k = the largest |component| (exact comparisons); (p, q) = (d_a, d_b) / d_k with a = k + 1, b = k + 2 (mod 3)
  k = (adx >= ady && adx >= adz) ? 0 : (ady >= adz ? 1 : 2);
  const float dk = k == 0 ? d.x : k == 1 ? d.y : d.z, da = k == 0 ? d.y : k == 1 ? d.z : d.x, db = k == 0 ? d.z : k == 1 ? d.x : d.y;
  const float rk = __builtin_amdgcn_rcpf(dk), halfR = 0.5f * (float)R;
  ci = min(max((int)__builtin_floorf((da * rk + 1.0f) * halfR), 0), R - 1);
  cj = min(max((int)__builtin_floorf((db * rk + 1.0f) * halfR), 0), R - 1);
}
__device__ __forceinline__ unsigned int tri_dir_bin(V3 d, int R) {
  int k, ci, cj;
  tri_dir_cell(d, R, k, ci, cj);
  return (unsigned int)((k * R + cj) * R + ci);
}

// centroid of a compressed band record: lo + k step per axis, one multiplication and one addition each, NOT fused — the host measures
// the deviation eps_c of exactly this decode (pt_tripool.hpp "compressed records")
__device__ __forceinline__ V3 tri_centroid(unsigned int kx, unsigned int ky, unsigned int kz, f4 H7, f4 H8) {
  return mk(H7.x + (float)kx * H8.x, H7.y + (float)ky * H8.y, H7.z + (float)kz * H8.z);
}

// ONE RAY (wave-uniform: in scalar operands) through one list of a direction map, the list's entries across the 64 lanes: stage 1 (the band
// test in integers on the gathered compressed records), stage 2 (the noise-radius filter, 64 queued survivors at a time), then the
// reference's test, 64 at a time; on_hit(key) receives every accepted candidate's tri_key.  Called by tri_pool_scan for each of a wave's
// rays in turn, and (round 6) by the binned band stage for the rays of packets too small to share a list (pt_binned.hpp).
struct TriBandCtx {
  f4 H5, H6, H7, H8;
  unsigned int band_rec, tri_sorted;
  int n_tri, goff;
};
// BUILD (round 6, the camera rays' candidate cache — tri_pool_scan): `ur` is the ray through a pixel's CENTRE and `dd` bounds |d - d_c| over
// every ray of the pixel (same origin: a pinhole camera); the two filters are evaluated in the form that holds for ALL of those rays at once
//     band    |d . n~| >= |d_c . n~| - dd |n~|  and  |d| <= |d_c| + dd:   |d_c . n~| - dd <= (|d_c| + dd) (pn (rho + ...) + eps_n)
//     radius  a1 >= (|d_c . n~| - dd - (|d_c| + dd) eps_n) nlow - ea L^2 (|d_c| + dd), the noise radius <= its value at that a1 and at |d_c| + dd,
//             and |(C - o) x d| >= |(C - o) x d_c| - |C - o| dd:   |(C - o) x d_c| <= rad (|d_c| + dd) + |C - o| dd
// — each a consequence of the pair's own condition, so what passes for some ray of the pixel passes here — and the survivors are not tested
// but handed to on_hit as the pixel's candidate list (on_hit(position), lanes below the returned count... see the caller).
template <bool BUILD = false, typename HitFn>
__device__ __forceinline__ void tri_band_one_ray(glb_f4p pool, const TriBandCtx& K, const Ray& ur, float ua, float rho, float dn, unsigned int first,
                                                 unsigned int last, unsigned int cand_off, bool listed, HitFn&& on_hit, float dd = 0.0f) {
#ifndef PT_NO_FILTER_FMA
#pragma clang fp contract(fast) /* filter arithmetic: see tri_pool_scan */
#endif
  const f4 H5 = K.H5, H6 = K.H6, H7 = K.H7, H8 = K.H8;
  const unsigned int band_rec = K.band_rec, tri_sorted = K.tri_sorted;
  const int n_tri = K.n_tri, goff = K.goff;
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  int* const tq = tri_queue() + (threadIdx.x >> 6) * PT_TRI_QUEUE; // per wave: what waits for the exact test
  int qn = 0; // entries queued for the exact test (uniform)
  auto push = [&](bool p, int e) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
    if (p) tq[qn + __builtin_popcountll(m & below)] = e;
    qn += __builtin_popcountll(m);
  };
  // the reference's test for the top min(64, qn) entries and the (uniform) ray; a hit goes to on_hit
  auto drain = [&](int keep) {
    __builtin_amdgcn_wave_barrier();
    while (qn > keep) {
      const int n = min(qn, 64);
      if constexpr (BUILD) on_hit(lane < n ? tq[qn - n + lane] : -1, n); // (wave-wide: n survivors, one per lane)
      else if (lane < n) {
        const unsigned int o = tri_sorted + 3u * (unsigned int)tq[qn - n + lane];
        const f4 R0 = pool[o], R1 = pool[o + 1], R2 = pool[o + 2];
        float t;
        if (tri_param(R0, R1, R2, ur, t) && !(t < PT_TMIN)) on_hit(tri_key(t, goff + 3 * as_i(R2.w)));
      }
      qn -= n;
    }
    __builtin_amdgcn_wave_barrier();
  };
  const float dn_hi = dn + dd * 1.0001f; // (BUILD: >= |d| for every ray of the pixel)
  // band test, then the noise-radius filter: the line within L + kr rho |d| / (|a'| - ea |d|) of the centroid
  // on the 16-byte compressed record (pt_tripool.hpp "compressed records"; every quantity rounded to the safe side)
  auto near_line = [&](V3 C, float rad) { // does the ray's LINE pass within `rad` of the point C?
    const V3 x = cross(C - ur.o, ur.d);
    if constexpr (BUILD) {
      const V3 co = C - ur.o;
      const float lim = rad * dn_hi + __builtin_amdgcn_sqrtf(dot(co, co)) * dd * 1.0001f;
      return dot(x, x) <= lim * lim * 1.00002f;
    }
    return dot(x, x) <= rad * rad * ua * 1.00001f;
  };
  auto band_pass = [&](f4 Q) {
    const unsigned int w0 = (unsigned int)as_i(Q.x), w1 = (unsigned int)as_i(Q.y), w2 = (unsigned int)as_i(Q.z), w3 = (unsigned int)as_i(Q.w);
    const float nx = (float)((int)(w0 << 16) >> 16), ny = (float)((int)w0 >> 16), nz = (float)((int)(w1 << 16) >> 16);
    const float pn = as_f((int)(w1 & 0xffff0000u)), L = as_f((int)(w3 & 0xffff0000u));
    float dq = __builtin_fabsf(ur.d.x * nx + ur.d.y * ny + ur.d.z * nz) * 3.0518509e-5f; // |d . n~|, n~ = (nx, ny, nz) / 32767
    if constexpr (BUILD) dq = __builtin_fmaxf(dq - dd * 1.0002f, 0.0f);                  // (a lower bound over the pixel's rays; |n~| <= 1 + 1e-4)
    const float rL = __builtin_amdgcn_rcpf(L) * 1.00001f;
    if (!(dq <= dn_hi * (pn * (rho + H5.y * L + H5.w * rL) + H8.w) * 1.00001f)) return false;
    const float L2 = L * L;
    const float nlow = 0.98f * H5.z * L * __builtin_amdgcn_rcpf(pn);          // <= |N|
    const float a1 = (dq - dn_hi * H8.w) * nlow - H6.w * L2 * dn_hi;           // <= |a'| - ea |d|
    const float rr = (H6.y + H6.z * L) * L2 * rho * dn_hi * __builtin_amdgcn_rcpf(a1) * 1.001f; // >= the noise radius; a1 <= 0: no bound
    const V3 C = tri_centroid(w2 & 0xffffu, w2 >> 16, w3 & 0xffffu, H7, H8);
    return !(a1 > 0.0f) || near_line(C, L + rr + H6.x + H7.w);
  };
  const V3 dh = __builtin_amdgcn_rsqf(ua) * ur.d; // unit direction (a few ulp: covered by the integer test's absolute slack)
  // Two stages.  Stage 1, on every enumerated candidate: the band test alone, in INTEGERS — the record's normal is three 16-bit
  // integers k / 32767, the ray's unit direction is rounded to the same grid once per ray, and two v_dot2_i32_i16 give
  // S = kn . kd exactly (|S| <= 32767^2 (1 + 1e-4): no overflow); |d^ . n^| <= |S| / 32767^2 + eps_n + eps_d with
  // eps_d = sqrt(3) / (2 * 32767) + 1e-6 (the rounding of the direction and of rsq), so the test below passes whenever the band
  // test of pt_tripool.hpp does.  Its survivors are queued and run the full filter 64 at a time (stage 2).
  const int kdx = (int)__builtin_rintf(dh.x * 32767.0f), kdy = (int)__builtin_rintf(dh.y * 32767.0f), kdz = (int)__builtin_rintf(dh.z * 32767.0f);
  short2_t dxy, dz0;
  { const unsigned int a = ((unsigned int)kdx & 0xffffu) | ((unsigned int)kdy << 16), b = (unsigned int)kdz & 0xffffu; __builtin_memcpy(&dxy, &a, 4); __builtin_memcpy(&dz0, &b, 4); }
  // (BUILD: |d^ - d^_c| <= 2 dd / |d_c| for the unit directions of the pixel's rays)
  const float e1s = H8.w + 2.75e-5f + (BUILD ? 2.02f * dd * __builtin_amdgcn_rcpf(dn) : 0.0f);
  auto band_stage1 = [&](f4 Q) {
    const unsigned int w0 = (unsigned int)as_i(Q.x), w1 = (unsigned int)as_i(Q.y), w3 = (unsigned int)as_i(Q.w);
    short2_t nxy, nzp;
    __builtin_memcpy(&nxy, &w0, 4); __builtin_memcpy(&nzp, &w1, 4); // (nzp's high half is pn's bits: multiplied by dz0's zero)
    const int S = __builtin_amdgcn_sdot2(nzp, dz0, __builtin_amdgcn_sdot2(nxy, dxy, 0, false), false);
    const float sa = (float)(S < 0 ? -S : S);
    const float pn = as_f((int)(w1 & 0xffff0000u)), L = as_f((int)(w3 & 0xffff0000u));
    const float rL = __builtin_amdgcn_rcpf(L) * 1.00001f;
    return sa <= (pn * (rho + H5.y * L + H5.w * rL) + e1s) * 1.0737e9f; // 32767^2 (1 + 2e-5)
  };
  f4* const bq = tri_bqueue() + (threadIdx.x >> 6) * PT_TRI_BQUEUE;
  int* const bqi = tri_bqueue_idx() + (threadIdx.x >> 6) * PT_TRI_BQUEUE;
  int bn = 0;
  auto bpush = [&](bool p, int e, f4 Q) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
    if (p) { const int at = bn + __builtin_popcountll(m & below); bq[at] = Q; bqi[at] = e; }
    bn += __builtin_popcountll(m);
  };
  auto drain_band = [&](int keep) {
    __builtin_amdgcn_wave_barrier();
    while (bn > keep) {
      const int n = min(bn, 64);
      bool pass = false;
      int e = 0;
      if (lane < n) { e = bqi[bn - n + lane]; pass = band_pass(bq[bn - n + lane]); }
      bn -= n;
      PT_TRI_COUNT(8, __builtin_popcountll(__builtin_amdgcn_ballot_w64(pass)));
      push(pass, e);
      drain(63);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // 128 entries per trip: lane l takes entries base + l, base + 64 + l.  The loop is a two-stage software pipeline — while the records
  // of trip i are tested, the records of trip i + 1 are being gathered and the indices of trip i + 2 loaded — because a ray's turn is a
  // chain of dependent loads (index -> record, a microsecond each from beyond L2) and ~20 trips long: un-pipelined, that latency was
  // what a wave waited for (both arrays carry spare entries behind their end: no clamping; what lies beyond `last` is masked).
  constexpr int PER = PT_BAND_PER;
  auto load_idx = [&](unsigned int base, int (&idx)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; j++) { const unsigned int k = base + 64u * (unsigned int)j + (unsigned int)lane; idx[j] = listed ? (int)gdword_stream(pool, cand_off, k) : (int)k; }
  };
  auto load_rec = [&](const int (&idx)[PER], f4 (&Q)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; j++) Q[j] = pool[band_rec + (unsigned int)idx[j]];
  };
  int idxA[PER], idxB[PER], idxC[PER];
  f4 QA[PER], QB[PER];
  // (indices past `last` read spare or foreign entries: clamp what they point at to a valid record)
  auto clamp_idx = [&](int (&idx)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; j++) idx[j] = min(max(idx[j], 0), n_tri + 127); // (the record tables carry >= 128 spare records)
  };
  load_idx(first, idxA); clamp_idx(idxA);
  load_idx(first + 64u * PER, idxB);
  load_rec(idxA, QA);
  for (unsigned int base = first; base < last; base += 64u * PER) {
    PT_TRI_COUNT(11, 1);
    load_idx(base + 128u * PER, idxC); // two trips ahead
    clamp_idx(idxB);
    load_rec(idxB, QB);                // one trip ahead
#ifdef PT_BAND_ONE_STAGE /* A/B: the whole filter on the gathered record at once (no integer stage, no LDS queue of records) */
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const bool pass = base + 64u * (unsigned int)j + (unsigned int)lane < last && band_pass(QA[j]);
      push(pass, idxA[j]);
      drain(63);
    }
    PT_TRI_COUNT(6, min(64u * PER, last - base));
#else
    unsigned int passmask = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) passmask |= (base + 64u * (unsigned int)j + (unsigned int)lane < last && band_stage1(QA[j])) ? (1u << j) : 0u;
    PT_TRI_COUNT(6, min(64u * PER, last - base));
    PT_TRI_COUNT(7, PT_TRI_WAVE_BITS(passmask));
#pragma unroll
    for (int j = 0; j < PER; j++) bpush((passmask >> j) & 1u, idxA[j], QA[j]);
    drain_band(63);
#endif
#pragma unroll
    for (int j = 0; j < PER; j++) { idxA[j] = idxB[j]; QA[j] = QB[j]; idxB[j] = idxC[j]; }
  }
  drain_band(0);
  drain(0);
}

// What a ray asks of the BINNED band stage (round 6; pt_render.hip: band_kernel): tri_pool_scan<true> runs the grid part in place and,
// instead of taking the wave's rays through their direction-map lists one at a time, says which list a ray needs — key = the bin of its
// direction in the map of its rho class, numbered across the maps (the last key: every band record) — so that the launcher can bring
// the rays of one bin together from all over the frame.  key < 0: no request (a dead lane, or the run was scanned in full).
struct TriDefer {
  int key;
  float rho;
};

// The camera rays' candidate cache (round 6).  Every sample of a pixel starts with a camera ray, and for a pinhole camera those rays share
// their origin and differ in direction by less than a pixel: the few dozen entries of the direction-map lists that survive the two filters
// are nearly the same for all of them.  A lane holds ONE pixel for all its samples (pt_render.hip), so it keeps that pixel's list: the first
// camera ray builds it — the bins the pixel's footprint touches, the filters in the form that holds for every ray of the pixel
// (tri_band_one_ray<true>) — and every camera ray of the pixel, that one included, runs the reference's test on the cached candidates
// instead of enumerating 2 300 map entries.  A superset of each ray's own candidate set: testing more triangles changes nothing.
//   cache line of a lane: [0] the pixel it belongs to (the kernel's pixel id), [1] entries (-1: this pixel gets no cache — footprint over a
//   face edge of the cube map, beyond the last rho class, or more than PT_TRI_CACHE_CAP survivors), [2 ...] positions in the Morton-ordered copy
#define PT_TRI_CACHE_WORDS 256
#define PT_TRI_CACHE_CAP (PT_TRI_CACHE_WORDS - 2)
struct TriPrimCtx {
  bool prim;  // per lane: this ray is a camera ray (the first of its sample) of a pinhole camera
  int xy;     // per lane: pixel x | y << 16
  int pix;    // per lane: the pixel's id in this render (the cache tag)
  const __attribute__((address_space(4))) float* foot; // uniform: (llc - origin) xyz, hor / W xyz, ver / H xyz, dd  (the kernel arguments' KArgs::foot)
  unsigned int* cache; // all lanes' cache lines; NULL: no cache
};

template <bool DEFER = false>
__device__ __forceinline__ bool tri_pool_scan(glb_f4p pool, cst_f4p cblob, int hdr, int goff, const RayCtx& c, HitState& h, TriDefer* dfr = nullptr,
                                              const TriPrimCtx* pc = nullptr) {
#ifndef PT_NO_FILTER_FMA
  // Everything written in this function is FILTER arithmetic — necessary conditions with explicit slack against exact mathematics
  // (pt_tripool.hpp), and the walk whose rounding the cells' absolute slack covers — so a product may fuse with the sum that
  // takes it: one rounding instead of two, never a larger error.  The reference's own test (tri_param -> tri_eval, functions of
  // their own) and the centroid decode above are compiled as written (-ffp-contract=off).
#pragma clang fp contract(fast)
#endif
  const f4 H0 = cblob[hdr], H1 = cblob[hdr + 1], H2 = cblob[hdr + 2], H3 = cblob[hdr + 3], H4 = cblob[hdr + 4], H5 = cblob[hdr + 5], H6 = cblob[hdr + 6];
  const f4 H7 = cblob[hdr + 7], H8 = cblob[hdr + 8]; // the quantisation of the compressed band records (pt_tripool.hpp)
  const V3 oc_own = c.r.o - xyz(H2);
  const float oc2_own = dot(oc_own, oc_own);
  const bool in_domain = c.reg && oc2_own <= H3.x;
  // (the persistent kernels: one ray outside the pool's domain and the whole wave scans the run; the binned renderer — DEFER — takes such
  // a ray out of the walk and files it under the key "every triangle, exactly": a full scan inside a generation would hold up the frame)
  if (!DEFER && __builtin_amdgcn_ballot_w64(c.live && !in_domain) != 0) return false;
  const bool c_live = c.live && (!DEFER || in_domain);
  const unsigned int cell_first = (unsigned int)as_i(H4.x), cell_cand = (unsigned int)as_i(H4.y), tri_sorted = (unsigned int)as_i(H4.z), band_rec = (unsigned int)as_i(H4.w);
  const int lane = threadIdx.x & 63;
  unsigned long long* const slot = tri_slots() + (threadIdx.x & ~63); // this wave's 64 slots
  const unsigned long long key0 = h.hit >= 0 ? tri_key(h.closest, hit_off(h.hit)) : ((unsigned long long)0x7f800000u << 32);
  slot[lane] = key0;
  const unsigned long long live = __builtin_amdgcn_ballot_w64(c_live);
  PT_TRI_COUNT(0, 1);
  PT_TRI_COUNT(1, __builtin_popcountll(live));
  int* const tq = tri_queue() + (threadIdx.x >> 6) * PT_TRI_QUEUE; // per wave: what waits for the exact test
  const unsigned long long below = (1ull << lane) - 1ull;
  const Ray& r = c.r;
  // ---- (1) the grid: every lane walks its own ray; (lane, triangle) pairs through the queue, 64 per batch ------------------------
#if !(PT_TRI_ABLATE & 1)
  {
    const float inv = H0.w, cell = H1.w, kappa = H3.y;
    const int nx = as_i(H1.x), ny = as_i(H1.y), nz = as_i(H1.z);
    const float gx = (r.o.x - H0.x) * inv, gy = (r.o.y - H0.y) * inv, gz = (r.o.z - H0.z) * inv; // origin in cell units
    const float rx = c.yx * cell, ry = c.yy * cell, rz = c.yz * cell;                         // 1 / (direction in cell units)
    const float ax = (0.0f - gx) * rx, bx = ((float)nx - gx) * rx;
    const float ay = (0.0f - gy) * ry, by = ((float)ny - gy) * ry;
    const float az = (0.0f - gz) * rz, bz = ((float)nz - gz) * rz;
    const float t_in = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float t_out = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    const float t0 = __builtin_fmaxf(t_in, 0.0f);
    float closest = h.closest; // the slot's t as last read back
    // closest (1 + kappa) — closest lowered by every hit so far — then the walk's own slack (relative 1e-4)
    auto limit = [&]() { const float m = __builtin_fminf(t_out, closest + closest * kappa); return m + (__builtin_fabsf(m) * 1e-4f + 1e-4f); };
    bool active = c_live && t0 <= limit();
    const float px = gx + t0 * (r.d.x * inv), py = gy + t0 * (r.d.y * inv), pz = gz + t0 * (r.d.z * inv);
    int ix = min(max((int)__builtin_floorf(px), 0), nx - 1);
    int iy = min(max((int)__builtin_floorf(py), 0), ny - 1);
    int iz = min(max((int)__builtin_floorf(pz), 0), nz - 1);
    const bool fx = r.d.x > 0.0f, fy = r.d.y > 0.0f, fz = r.d.z > 0.0f;
    float tmx = ((float)(ix + (fx ? 1 : 0)) - gx) * rx, tmy = ((float)(iy + (fy ? 1 : 0)) - gy) * ry, tmz = ((float)(iz + (fz ? 1 : 0)) - gz) * rz;
    const int stx = fx ? 1 : -1, sty = fy ? 1 : -1, stz = fz ? 1 : -1;
    // The cells' candidate ranges are EXPANDED over the wave: an inclusive scan of the lanes' counts gives every (lane, candidate) entry of
    // this wave-step a position p in [0, T); trip by trip the 64 lanes take positions p = base + lane, find the entry's owner (the lane
    // whose range contains p: a binary search over the scan, six ds_bpermute steps) and read it (neighbouring lanes, neighbouring
    // addresses).  An entry whose triangle the owner's PREVIOUS cell lists as well is dropped — it was tested there, or where that
    // cell's predecessor listed it: the entry carries one bit per face neighbour (pt_tripool.hpp), the owner says through which face it
    // came — which removes the two out of three tests that repeated one made a cell earlier.  What is left is queued per wave in LDS as
    // (owner << 26 | triangle) and, 64 pairs at a time, runs the reference's test for the owner's ray.
    // (The first version let every lane loop over ITS cell's list: a wave-step took as many trips as its fullest cell has candidates,
    // most of them at a handful of lanes — 1 844 ms at 1080p x 8 spp against 568 ms of round 4's kernel.)
    int qn = 0; // pairs queued (wave-uniform)
    // up to PT_TRI_PAIRS x 64 pairs (the top of the queue) through the reference's test: the records of all of them are requested before
    // the first is tested (round 6: a batch is a chain LDS -> three gathers -> test, and a wave-step queues ten of them: one batch at a
    // time, the wave waited a memory latency per batch)
#ifndef PT_TRI_PAIRS
#define PT_TRI_PAIRS 2
#endif
    constexpr int NP = PT_TRI_PAIRS; // (the persistent kernels: two in flight at five waves per SIMD, 96 VGPRs — 1080p x 32 spp 1 009 -> 979 ms; at six waves / 80 VGPRs the same code spills and loses 10 %: profiles/r06_ab_tri_pipe.txt)
    auto test_batch = [&]() {
      __builtin_amdgcn_wave_barrier();
      int nb[NP], srcs[NP];
      f4 R0[NP], R1[NP], R2[NP];
      int left = qn;
#pragma unroll
      for (int k = 0; k < NP; k++) {
        const int n = min(left, 64);
        nb[k] = n;
        const bool on = lane < n;
        const unsigned int e = (unsigned int)tq[left - n + (on ? lane : 0)];
        srcs[k] = (int)(e >> 26);
        const unsigned int o = tri_sorted + 3u * (on ? (e & 0x3ffffffu) : 0u);
        R0[k] = pool[o]; R1[k] = pool[o + 1]; R2[k] = pool[o + 2];
        left -= n;
      }
#pragma unroll
      for (int k = 0; k < NP; k++) {
        if (nb[k] == 0) continue; // (wave-uniform)
        PT_TRI_COUNT(5, 1);
        PT_TRI_COUNT(4, nb[k]);
        const int src = srcs[k];
        Ray r2; // the pair's ray, from its owner's registers
        r2.o = mk(__shfl(r.o.x, src, 64), __shfl(r.o.y, src, 64), __shfl(r.o.z, src, 64));
        r2.d = mk(__shfl(r.d.x, src, 64), __shfl(r.d.y, src, 64), __shfl(r.d.z, src, 64));
        r2.tm = 0.0f;
        if (lane < nb[k]) {
          float t;
          if (tri_param(R0[k], R1[k], R2[k], r2, t) && !(t < PT_TMIN)) atomicMin(&slot[src], tri_key(t, goff + 3 * as_i(R2[k].w)));
        }
      }
      qn = left;
      __builtin_amdgcn_wave_barrier();
    };
    unsigned int k0 = 0, k1 = 0; // this cell's candidate range
    int came = 6;                // the face through which the walk entered this cell: 0 from -x, 1 from +x, 2 -y, 3 +y, 4 -z, 5 +z; 6: the walk's first cell
    if (active) { const int ci = (iz * ny + iy) * nx + ix; k0 = gdword(pool, cell_first, (unsigned int)ci); k1 = gdword(pool, cell_first, (unsigned int)ci + 1u); }
    while (__builtin_amdgcn_ballot_w64(active) != 0) {
      PT_TRI_COUNT(2, 1);
      PT_TRI_COUNT(3, __builtin_popcountll(__builtin_amdgcn_ballot_w64(active)));
      // the next cell (the axis whose boundary comes first; branch-free) and its range, requested BEFORE this cell's candidates are tested
      const float tn = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
      const bool sx = tmx == tn, sy = !sx & (tmy == tn), sz = !sx & !sy;
      const int jx = ix + (sx ? stx : 0), jy = iy + (sy ? sty : 0), jz = iz + (sz ? stz : 0);
      const bool inside = ((unsigned)jx < (unsigned)nx) & ((unsigned)jy < (unsigned)ny) & ((unsigned)jz < (unsigned)nz);
      unsigned int n0 = 0, n1 = 0;
      if (active & inside) { const int cj = (jz * ny + jy) * nx + jx; n0 = gdword(pool, cell_first, (unsigned int)cj); n1 = gdword(pool, cell_first, (unsigned int)cj + 1u); }
      int incl = (int)(k1 - k0); // (0 for a lane that does not walk)
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) { const int o2 = __shfl_up(incl, dd, 64); incl += lane >= dd ? o2 : 0; }
      const int T = __builtin_amdgcn_readlane(incl, 63);
      const int kbase = (int)k1 - incl; // candidate index of position p for this lane's range: kbase + p
      const unsigned int skip_bit = came < 6 ? (1u << (26 + came)) : 0u;
      // (round 6: PT_TRI_TRIPS trips' entries are requested before the first is looked at — a trip was a chain owner search -> load -> ballot,
      // and a wave-step of 64 lanes has ten of them: one at a time, the wave waited a memory latency per trip)
#ifndef PT_TRI_TRIPS
#define PT_TRI_TRIPS 4
#endif
      constexpr int NT = DEFER ? PT_TRI_TRIPS : 2; // (the binned step kernel has 128 VGPRs: four; the persistent kernels two)
      for (int base = 0; base < T; base += 64 * NT) {
        unsigned int es[NT], sbs[NT];
        int srcs[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) {
          const int p = base + 64 * k + lane;
          const bool on = p < T;
          // owner: the first lane whose inclusive count exceeds p (lanes without candidates repeat their predecessor's count and are never first)
          int lo = 0;
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1) { const int v = __shfl(incl, lo + st - 1, 64); lo += v <= p ? st : 0; }
          const int src = on ? lo : lane;
          const int kb = __shfl(kbase, src, 64);
          sbs[k] = on ? (unsigned int)__shfl((int)skip_bit, src, 64) : 0xffffffffu; // (off: every entry "skipped")
          srcs[k] = src;
          es[k] = 0xffffffffu;
          if (on) es[k] = gdword_stream(pool, cell_cand, (unsigned int)(kb + p));
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
          const bool keep = (es[k] & sbs[k]) == 0u;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
          if (keep) tq[qn + __builtin_popcountll(m & below)] = (int)(((unsigned int)srcs[k] << 26) | (es[k] & 0x3ffffffu));
          qn += __builtin_popcountll(m);
          if (qn >= 64 * NP) test_batch();
        }
      }
      if (T > 0) { // (the limit below looks at the nearest hit so far: everything queued is tested first)
        while (qn > 0) test_batch();
        closest = as_f((int)(unsigned int)(slot[lane] >> 32));
      }
      // step: the walk ends where the next cell lies outside the grid or begins beyond the nearest hit so far
      active = active & inside & !(tn > limit());
      came = sx ? (fx ? 0 : 1) : sy ? (fy ? 2 : 3) : (fz ? 4 : 5); // stepped +x: entered through the new cell's -x face, ...
      ix = jx; iy = jy; iz = jz;
      // the stepped axis' next boundary, from the cell INDEX (round 6): t = (boundary - g) r carries ~2 u (|g| + |t d|) of rounding
      // whatever the number of steps — what the cells' slack is sized for (pt_tripool.hpp: rlimit) — where the running sum tmx += dtx of
      // rounds 3-5 added a rounding per step (k steps at D cells from the origin: ~k u D cells, beyond the slack for k D > 10^5)
      tmx = sx ? ((float)(ix + (fx ? 1 : 0)) - gx) * rx : tmx;
      tmy = sy ? ((float)(iy + (fy ? 1 : 0)) - gy) * ry : tmy;
      tmz = sz ? ((float)(iz + (fz ? 1 : 0)) - gz) * rz : tmz;
      k0 = active ? n0 : 0u; k1 = active ? n1 : 0u;
    }
  }
#endif
  // ---- (2) the direction map: the triangles a ray grazes, one ray at a time ------------------------------------------------------
  // band test of triangle i for a ray: |d . n~_i| <= |d| (pn (rho + KQ L + KT / L) + eps_n); rho and |d| rounded up
  const float rho_own = (__builtin_amdgcn_sqrtf(oc2_own) + H2.w) * 1.000002f;
  const float dn_own = __builtin_amdgcn_sqrtf(c.a) * 1.000002f;
  const int n_tri = as_i(H3.z), n_maps = (PT_TRI_ABLATE & 2) ? 0 : as_i(H3.w);
  TriBandCtx bctx;
  bctx.H5 = H5; bctx.H6 = H6; bctx.H7 = H7; bctx.H8 = H8; bctx.band_rec = band_rec; bctx.tri_sorted = tri_sorted; bctx.n_tri = n_tri; bctx.goff = goff;
  // Every lane looks up ITS ray's list first (the bin of its direction in the map of its rho class — or, beyond the last class, every
  // triangle): two gathers for the whole wave instead of two dependent scalar loads at the head of every ray's turn.
  unsigned int first_own = 0, last_own = (unsigned int)n_tri, cand_own = 0;
  int listed_own = 0;
  {
    const f4 D0 = cblob[hdr + 9], D1 = cblob[hdr + 10], D2 = cblob[hdr + 11];
    const bool c0 = n_maps > 0 && rho_own <= D0.y, c1 = n_maps > 1 && rho_own <= D1.y, c2 = n_maps > 2 && rho_own <= D2.y;
    if constexpr (DEFER) {
      // the binned band stage: hand the request out (keys: map 0's bins, then map 1's, then map 2's, then "every record")
      const unsigned long long kf = slot[lane];
      if (kf != key0) { h.closest = as_f((int)(unsigned int)(kf >> 32)); h.hit = hit_pack(DK_TRI, 0, (int)(0xffffffffu - (unsigned int)(kf & 0xffffffffull))); }
      const int R0 = n_maps > 0 ? as_i(D0.x) : 0, R1 = n_maps > 1 ? as_i(D1.x) : 0, R2 = n_maps > 2 ? as_i(D2.x) : 0;
      const int base1 = 3 * R0 * R0, base2 = base1 + 3 * R1 * R1, base_all = base2 + 3 * R2 * R2;
      dfr->rho = rho_own;
      dfr->key = c_live ? base_all : c.live ? base_all + 1 : -1; // (base_all: every band record; base_all + 1: every triangle, exactly)
      if (c_live && (c0 || c1 || c2)) {
        const f4 D = c0 ? D0 : c1 ? D1 : D2;
        dfr->key = (c0 ? 0 : c1 ? base1 : base2) + (int)tri_dir_bin(r.d, as_i(D.x));
      }
      return true;
    }
    if (c_live && (c0 || c1 || c2)) {
      const f4 D = c0 ? D0 : c1 ? D1 : D2;
      const unsigned int bin = tri_dir_bin(r.d, as_i(D.x));
      const unsigned int foff = (unsigned int)as_i(D.z);
      first_own = gdword(pool, foff, bin); last_own = gdword(pool, foff, bin + 1u);
      cand_own = (unsigned int)as_i(D.w);
      listed_own = c0 ? 1 : c1 ? 2 : 3;
    }
  }
  for (unsigned long long todo = (PT_TRI_ABLATE & 4) ? 0ull : live; todo != 0; todo &= todo - 1) {
    const int src = __builtin_ctzll(todo);
    Ray ur;
    ur.o = mk(rl_f(r.o.x, src), rl_f(r.o.y, src), rl_f(r.o.z, src));
    ur.d = mk(rl_f(r.d.x, src), rl_f(r.d.y, src), rl_f(r.d.z, src));
    ur.tm = 0.0f;
    const float ua = rl_f(c.a, src), rho = rl_f(rho_own, src), dn = rl_f(dn_own, src);
    const unsigned int first = (unsigned int)__builtin_amdgcn_readlane((int)first_own, src), last = (unsigned int)__builtin_amdgcn_readlane((int)last_own, src);
    const unsigned int cand_off = (unsigned int)__builtin_amdgcn_readlane((int)cand_own, src);
    const int listed_k = __builtin_amdgcn_readlane(listed_own, src);
    const bool listed = listed_k != 0;
    PT_TRI_COUNT(9, listed_k == 3 ? 1 : 0);
    PT_TRI_COUNT(12, listed_k == 2 ? 1 : 0);
    PT_TRI_COUNT(10, listed ? 0 : 1);
    if (pc != nullptr && pc->cache != nullptr && __builtin_amdgcn_readlane((int)pc->prim, src) != 0) { // a camera ray: its pixel's cached candidates
      unsigned int* const cs = pc->cache + (size_t)(blockIdx.x * blockDim.x + (threadIdx.x & ~63u) + (unsigned int)src) * PT_TRI_CACHE_WORDS;
      const int pix = __builtin_amdgcn_readlane(pc->pix, src);
      int n = -1;
      if (__builtin_amdgcn_readfirstlane((int)cs[0]) == pix) n = __builtin_amdgcn_readfirstlane((int)cs[1]);
      else {
        // build: the centre ray of the pixel, the bins its footprint touches
        const int xy = __builtin_amdgcn_readlane(pc->xy, src);
        const float sc = ((float)(xy & 0xffff) + 0.5f), tc = ((float)(xy >> 16) + 0.5f);
        const V3 fb = mk(pc->foot[0], pc->foot[1], pc->foot[2]), fh = mk(pc->foot[3], pc->foot[4], pc->foot[5]), fv = mk(pc->foot[6], pc->foot[7], pc->foot[8]);
        const float dd = pc->foot[9];
        Ray cr;
        cr.o = ur.o; cr.tm = 0.0f;
        cr.d = fb + sc * fh + tc * fv;
        const float cua = dot(cr.d, cr.d), cdn = __builtin_amdgcn_sqrtf(cua) * 1.000002f;
        n = 0;
        if (!listed) n = -1; // (beyond the last rho class: such a ray streams every record anyway)
        else {
          const f4 D = cblob[hdr + 8 + listed_k];
          const int R = as_i(D.x);
          int k0 = 0, i0 = R, i1 = -1, j0 = R, j1 = -1;
          bool one_face = true;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const V3 dq4 = cr.d + ((q & 1) ? 0.5f : -0.5f) * fh + ((q & 2) ? 0.5f : -0.5f) * fv;
            int k, ci, cj;
            tri_dir_cell(dq4, R, k, ci, cj);
            if (q == 0) k0 = k;
            one_face = one_face && k == k0;
            i0 = min(i0, ci); i1 = max(i1, ci); j0 = min(j0, cj); j1 = max(j1, cj);
          }
          if (!one_face || (i1 - i0 + 1) * (j1 - j0 + 1) > 4) n = -1;
          const unsigned int foff = (unsigned int)as_i(D.z);
          for (int cj = j0; cj <= j1 && n >= 0; cj++)
            for (int ci = i0; ci <= i1 && n >= 0; ci++) {
              const unsigned int bin = (unsigned int)((k0 * R + cj) * R + ci);
              const unsigned int bf = (unsigned int)__builtin_amdgcn_readfirstlane((int)gdword(pool, foff, bin)), bl = (unsigned int)__builtin_amdgcn_readfirstlane((int)gdword(pool, foff, bin + 1u));
              tri_band_one_ray<true>(pool, bctx, cr, cua, rho, cdn, bf, bl, (unsigned int)as_i(D.w), true,
                                     [&](int e, int cnt) {
                                       if (n >= 0 && n + cnt <= PT_TRI_CACHE_CAP) { if (lane < cnt) cs[2 + n + lane] = (unsigned int)e; n += cnt; }
                                       else n = -1;
                                     }, dd);
            }
        }
        if (lane == 0) { cs[0] = (unsigned int)pix; cs[1] = (unsigned int)n; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // (the list is read back by other lanes of this wave, now and for the pixel's later samples)
        __builtin_amdgcn_wave_barrier();
      }
      if (n >= 0) {
        PT_TRI_COUNT(13, 1);
        PT_TRI_COUNT(14, n);
        for (int base = 0; base < n; base += 64) {
          if (base + lane < n) {
            const unsigned int o = tri_sorted + 3u * cs[2 + base + lane];
            const f4 R0 = pool[o], R1 = pool[o + 1], R2 = pool[o + 2];
            float t;
            if (tri_param(R0, R1, R2, ur, t) && !(t < PT_TMIN)) atomicMin(&slot[src], tri_key(t, goff + 3 * as_i(R2.w)));
          }
        }
        continue;
      }
    }
    tri_band_one_ray(pool, bctx, ur, ua, rho, dn, first, last, cand_off, listed, [&](unsigned long long key) { atomicMin(&slot[src], key); });
  }
  // each lane reads its own ray's slot back: changed = some triangle of this run is the nearest hit so far
  const unsigned long long kf = slot[lane];
  if (kf != key0) { h.closest = as_f((int)(unsigned int)(kf >> 32)); h.hit = hit_pack(DK_TRI, 0, (int)(0xffffffffu - (unsigned int)(kf & 0xffffffffull))); }
  return true;
}

// RECTBOX: the scene holds rects and boxes only (MATS_RECTBOX_ONLY kernels): the sphere / triangle / medium loops are not compiled in —
// less code, and nothing of theirs (the medium's sqrt(d.d), say) can be hoisted into the per-iteration prologue of a kernel that never runs it.
template <bool IMG, int TRIP = 1, int TTRIP = TRIP, bool WHOLE = true, bool BADOUEL = false, int GRID = 1, bool TRIPOOL = false, bool RECTBOX = false, bool DEFER = false, typename P>
__device__ __forceinline__ void hit_records(P recs, cst_f4p cblob, int kind, int n, int goff,
                                            const RayCtx& c, bool fast, uint32_t& rng, HitState& h, glb_f4p pool = nullptr, TriDefer* dfr = nullptr,
                                            const TriPrimCtx* pc = nullptr) {
  const Ray& r = c.r;
  int off = 0;
  if (!RECTBOX && kind == DK_SPHERE) {
    auto accept_at = [&](int o) {
      return [&h, &r, recs, goff, o](float t) {
        h.closest = t;
        h.hit = hit_pack(DK_SPHERE, 0, goff + o);
        if (IMG) {
          f4 R0 = recs[o], R1 = recs[o + 1], R2 = recs[o + 2];
          V3 p = r.o + t * r.d;
          V3 n_ = (p - sphere_center(R0, R1, R2, r.tm)) / R1.x;
          bool ff = dot(r.d, n_) < 0;
          V3 nn = ff ? n_ : mk(0.0f, 0.0f, 0.0f) - n_;
          mercator(nn, h.u, h.v);
        }
      };
    };
    // (a run of one or two spheres — a lone ball between other kinds — is tested in place: the list machinery would cost it
    // two dependent scalar loads before the first record is even requested)
    if (WHOLE && n > 2) sphere_scan<(TRIP >= 2 ? 4 : 2), GRID, TRIPOOL>(recs, cblob, n, goff, c, h, accept_at);
    else {
      TimeFrac tf = time_frac_none();
      for (int i = 0; i < n; ++i, off += SZ_SPHERE) sphere_roots(recs, off, c, PT_TMIN, h.closest, true, tf, accept_at(off));
    }
  } else if (kind == DK_RECT) {
    for (int i = 0; i < n; ++i, off += SZ_RECT) {
      f4 R0 = recs[off], R1 = recs[off + 1];
      float t, ca, cb;
      const bool acc = rect_any(fast, as_i(R1.z), R0.x, R0.y, R0.z, R0.w, R1.x, c, h.closest, t, ca, cb);
      if (acc) {
        h.closest = t;
        h.hit = hit_pack(DK_RECT, 0, goff + off);
        if (IMG) { h.u = (ca - R0.x) / (R0.y - R0.x); h.v = (cb - R0.z) / (R0.w - R0.z); }
      }
    }
  } else if (!RECTBOX && kind == DK_TRI) {
    if constexpr (TRIPOOL && WHOLE) { // a long run with a triangle pool (flag + header offset in the run's aux record)
      const f4 aux = cblob[goff - 1];
      if (as_i(aux.x) != 0 && (fast || DEFER)) { // (DEFER: irregular rays are taken out lane by lane, inside)
        if (tri_pool_scan<DEFER>(pool, cblob, as_i(aux.y), goff, c, h, dfr, pc)) return;
      }
    }
    auto accept_at = [&](int o) { return [&h, goff, o](float t) { h.closest = t; h.hit = hit_pack(DK_TRI, 0, goff + o); }; };
    int i = 0;
    // two triangles per trip: six record reads in flight together, two independent arithmetic chains, half the loop
    // control; the second halves still run in list order (the second sees the first's closest)
    auto eval_at = [&](int o) { return tri_eval(recs[o], recs[o + 1], recs[o + 2], r); };
    auto finish_at = [&](int o, TriEval e) { // re-reads the records: only inside the rare branch
      if (e.pass) tri_finish(recs[o], recs[o + 1], recs[o + 2], r, e, PT_TMIN, h.closest, true, accept_at(o));
    };
    if constexpr (TTRIP >= 4) {
      for (; i + 3 < n; i += 4, off += 4 * SZ_TRI) {
        const TriEval ea = eval_at(off), eb = eval_at(off + SZ_TRI), ec = eval_at(off + 2 * SZ_TRI), ed = eval_at(off + 3 * SZ_TRI);
        finish_at(off, ea); finish_at(off + SZ_TRI, eb); finish_at(off + 2 * SZ_TRI, ec); finish_at(off + 3 * SZ_TRI, ed);
      }
    }
    if constexpr (TTRIP >= 2) {
      for (; i + 1 < n; i += 2, off += 2 * SZ_TRI) {
        const TriEval ea = eval_at(off), eb = eval_at(off + SZ_TRI);
        finish_at(off, ea); finish_at(off + SZ_TRI, eb);
      }
    }
    for (; i < n; ++i, off += SZ_TRI) finish_at(off, eval_at(off));
  } else if (BADOUEL && kind == DK_TRI_B) {
    for (int i = 0; i < n; ++i, off += SZ_TRI)
      badouel_test(recs[off], recs[off + 1], recs[off + 2], r, PT_TMIN, h.closest,
                   [&](float t) { h.closest = t; h.hit = hit_pack(DK_TRI_B, 0, goff + off); });
  } else if (kind == DK_BOX || (RECTBOX && kind != DK_RECT)) {
#ifndef PT_NO_CMPX
    if (fast && !IMG) {
      const unsigned long long exec_all = __builtin_amdgcn_ballot_w64(true); // EXEC as it is around the scan
      for (int i = 0; i < n; ++i, off += SZ_BOX) {
        int hit_base = hit_pack(DK_BOX, 0, goff + off);
        asm volatile("" : "+v"(hit_base)); // in a VGPR: the side's v_or takes the side bits as its literal
        box_cmpx(recs[off], recs[off + 1], c, exec_all, hit_base, h.closest, h.hit);
      }
    } else
#endif
    if (fast) {
      for (int i = 0; i < n; ++i, off += SZ_BOX) {
        float t, bu = 0.0f, bv = 0.0f;
        int side = 0;
        const bool acc = box_fast<IMG>(recs[off], recs[off + 1], c, h.closest, t, side, bu, bv);
        h.closest = acc ? t : h.closest;
        h.hit = acc ? hit_pack(DK_BOX, side, goff + off) : h.hit;
        if (IMG) { if (acc) { h.u = bu; h.v = bv; } }
      }
    } else {
      for (int i = 0; i < n; ++i, off += SZ_BOX) {
        float t, bu = 0.0f, bv = 0.0f;
        int side = 0;
        if (box_plain<IMG>(recs[off], recs[off + 1], c, PT_TMIN, h.closest, t, side, bu, bv)) {
          h.closest = t;
          h.hit = hit_pack(DK_BOX, side, goff + off);
          if (IMG) { h.u = bu; h.v = bv; }
        }
      }
    }
  } else {
    for (int i = 0; i < n; ++i, off += SZ_MEDIUM) {
      float t;
      if (medium_t(recs, off, c, PT_TMIN, h.closest, rng, t)) {
        h.closest = t;
        h.hit = hit_pack(DK_MEDIUM, 0, goff + off);
      }
    }
  }
}

// One run, split over a group of G = 2^logG lanes (cooperative traversal, regular rays only, never a medium run): lane j
// tests the records whose hittable index is == j (mod G).  The loop itself stays wave-uniform — ceil(cnt / G) trips for
// everybody, a lane without a record in a trip re-tests record 0 of the run and discards the outcome — so the only
// per-lane things are the record address and one predicate (divergent trip counts cost ~10 SALU + 2 branches per trip).
template <bool IMG, bool WHOLE = true, typename P>
__device__ __forceinline__ void hit_records_strided(P recs, cst_f4p cblob, int kind, int cnt, int first, int goff, int j, int logG,
                                                    const RayCtx& c, HitState& h) {
  const Ray& r = c.r;
  const int G = 1 << logG, trips = (cnt + G - 1) >> logG;
  int k = (j - first) & (G - 1);
  if (kind == DK_SPHERE) {
    auto accept_at = [&](int o) {
      return [&h, &r, recs, goff, o](float t) {
        h.closest = t;
        h.hit = hit_pack(DK_SPHERE, 0, goff + o);
        if (IMG) {
          f4 R0 = recs[o], R1 = recs[o + 1], R2 = recs[o + 2];
          V3 p = r.o + t * r.d;
          V3 n_ = (p - sphere_center(R0, R1, R2, r.tm)) / R1.x;
          bool ff = dot(r.d, n_) < 0;
          V3 nn = ff ? n_ : mk(0.0f, 0.0f, 0.0f) - n_;
          mercator(nn, h.u, h.v);
        }
      };
    };
    bool listed = false;
    if constexpr (WHOLE) {
      // Lane j takes the list ENTRIES (four spheres each) j, j + G, ... of the run's static and then of its moving list
      // (sphere_scan): four first records in flight per trip, no per-lane "is it moving" branch.  A lane past the end of a
      // list repeats the last entry — a duplicate candidate, which the tie rule (own scan) and the merge (same t, same
      // record) both ignore.
      const f4 aux = cblob[goff - 1];
      if (as_i(aux.w) & 1) {
        listed = true;
        const int ns = as_i(aux.z), nm = cnt - (ns - (as_i(aux.w) >> 8)), qs = (ns + 3) >> 2, qm = (nm + 3) >> 2;
        const P lists = recs - (1 + ((as_i(aux.w) & 4) ? 4 : 0) + qs + qm); // (a grid's four header records sit between the lists and aux)
        auto entry = [&](int e) {
          const f4 v = lists[e];
          return i4{as_i(v.x), as_i(v.y), as_i(v.z), as_i(v.w)};
        };
        for (int e = j, i = 0; i < ((qs + G - 1) >> logG); ++i, e += G)
          sphere_list_entry<false, PT_STRIDED_K>(recs, entry(min(e, qs - 1)), goff, 0.0f, c, h, accept_at);
        if (qm) {
          const float frac = (r.tm - aux.x) / (aux.y - aux.x); // sphere.hpp:54
          for (int e = j, i = 0; i < ((qm + G - 1) >> logG); ++i, e += G)
            sphere_list_entry<true, PT_STRIDED_K>(recs, entry(qs + min(e, qm - 1)), goff, frac, c, h, accept_at);
        }
      }
    }
    if (!listed) {
      TimeFrac tf = time_frac_none();
      for (int i = 0; i < trips; ++i, k += G) {
        const bool valid = k < cnt;
        const int off = (valid ? k : 0) * SZ_SPHERE;
        sphere_roots(recs, off, c, PT_TMIN, h.closest, valid, tf, accept_at(off));
      }
    }
  } else if (kind == DK_RECT) {
    for (int i = 0; i < trips; ++i, k += G) {
      const bool valid = k < cnt;
      const int off = (valid ? k : 0) * SZ_RECT;
      f4 R0 = recs[off], R1 = recs[off + 1];
      float t, ca, cb;
      const bool acc = rect_any(true, as_i(R1.z), R0.x, R0.y, R0.z, R0.w, R1.x, c, h.closest, t, ca, cb) && valid;
      if (acc) {
        h.closest = t;
        h.hit = hit_pack(DK_RECT, 0, goff + off);
        if (IMG) { h.u = (ca - R0.x) / (R0.y - R0.x); h.v = (cb - R0.z) / (R0.w - R0.z); }
      }
    }
  } else if (kind == DK_TRI) {
    for (int i = 0; i < trips; ++i, k += G) {
      const bool valid = k < cnt;
      const int off = (valid ? k : 0) * SZ_TRI;
      const f4 A0 = recs[off], A1 = recs[off + 1], A2 = recs[off + 2];
      tri_finish(A0, A1, A2, r, tri_eval(A0, A1, A2, r), PT_TMIN, h.closest, valid,
                 [&](float t) { h.closest = t; h.hit = hit_pack(DK_TRI, 0, goff + off); });
    }
  } else { // DK_BOX
    for (int i = 0; i < trips; ++i, k += G) {
      const bool valid = k < cnt;
      const int off = (valid ? k : 0) * SZ_BOX;
      float t, bu = 0.0f, bv = 0.0f;
      int side = 0;
      const bool acc = box_fast<IMG>(recs[off], recs[off + 1], c, h.closest, t, side, bu, bv) & valid;
      h.closest = acc ? t : h.closest;
      h.hit = acc ? hit_pack(DK_BOX, side, goff + off) : h.hit;
      if (IMG) { if (acc) { h.u = bu; h.v = bv; } }
    }
  }
}

__device__ __forceinline__ int record_size(int kind) {
  return kind == DK_SPHERE ? SZ_SPHERE : kind == DK_RECT ? SZ_RECT : (kind == DK_TRI || kind == DK_TRI_B) ? SZ_TRI : kind == DK_BOX ? SZ_BOX : SZ_MEDIUM;
}

// Whole list, blob resident (LDS or scalar cache): blob = [n_runs x (kind, first record offset, count, -)] [records]
// `headers` may be another view of the same blob: the LDS kernels read the run headers through the scalar cache (they are
// kernel constants: an s_load lands in SGPRs directly, no LDS round trip + v_readfirstlane per run) and the records from LDS.
// `cblob`: the blob in global memory through the scalar cache (run headers, sphere-run masks); `blob`: where the records are
// read from (LDS copy, or the same global blob).
// hit_world_range: the runs [ri0, ri1) of the list, on top of the hit h already holds (hit_world = the whole list from nothing).
// DEFER (the binned triangle-pool renderer, pt_render.hip: bin_step_kernel): a pooled triangle run does its grid part and leaves
// its direction-map part as a request in *dfr (tri_pool_scan<true>).
template <bool IMG, bool BADOUEL = false, int GRID = 1, bool TRIPOOL = false, bool RECTBOX = false, bool DEFER = false, typename P>
__device__ __forceinline__ void hit_world_range(P blob, cst_f4p cblob, int ri0, int ri1, const RayCtx& c, bool fast, uint32_t& rng, HitState& h, const f4* pool = nullptr,
                                                TriDefer* dfr = nullptr, const TriPrimCtx* pc = nullptr) {
  for (int ri = ri0; ri < ri1; ++ri) {
#ifdef PT_STAMPS_RUNS /* diagnostic build: cycles per run of the list (wave leader's clock), g_runs[min(ri, 15)] */
    const unsigned long long run_t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    struct RunStamp { unsigned long long t0; int ri; const HitState* h; __device__ ~RunStamp() { asm volatile("" ::"v"(h->closest), "v"(h->hit)); __builtin_amdgcn_sched_barrier(0);
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_runs[ri < 15 ? ri : 15], __builtin_amdgcn_s_memtime() - t0); } } run_stamp{run_t0, ri, &h};
#endif
    f4 runf = cblob[ri];
    const int off = as_i(runf.y), kind = as_i(runf.x);
    if (kind & DK_ABSORBED) continue; // a sphere run that an earlier run's lists have tested already (pt_flatten.hpp "absorbed sphere runs")
#ifdef PT_ABLATE_TAIL /* measurement-only build (a WRONG image): what would the runs behind the first cost if they were free? */
    if (ri > 0) continue;
#endif
    if constexpr (!IMG) {
      if (fast && (kind == DK_RECT || kind == DK_BOX)) { // head of a slab pool: the pool covers this run and the next span - 1
        const f4 aux = cblob[off - 1];                   // (largest |coordinate|, span, pool offset, entries)
        if (as_i(aux.y) != 0) {
          slab_pool(blob, cblob, as_i(aux.z), as_i(aux.w), aux.x, c, h);
          ri += as_i(aux.y) - 1;
          continue;
        }
      }
    }
    // (four triangles per trip in the grid kernels — 96 VGPRs — measured: nothing, 347-351 ms either way on the 496-hittable scene)
    hit_records<IMG, 1, 1, true, BADOUEL, GRID, TRIPOOL, RECTBOX, DEFER>(blob + off, cblob, kind, as_i(runf.z), off, c, fast, rng, h, (glb_f4p)pool, dfr, pc);
  }
}
template <bool IMG, bool BADOUEL = false, int GRID = 1, bool TRIPOOL = false, bool RECTBOX = false, typename P>
__device__ __forceinline__ void hit_world(P blob, cst_f4p cblob, int n_runs, const RayCtx& c, bool fast, uint32_t& rng, HitState& h, const f4* pool = nullptr,
                                          const TriPrimCtx* pc = nullptr) {
  hit_begin(h);
  hit_world_range<IMG, BADOUEL, GRID, TRIPOOL, RECTBOX, false>(blob, cblob, 0, n_runs, c, fast, rng, h, pool, nullptr, pc);
}

// Wave-uniform switch: the straight-line rect/box path is used only when every live lane's ray is regular.
__device__ __forceinline__ bool wave_all_regular(const RayCtx& c, bool live) {
  return __builtin_amdgcn_ballot_w64(live && !c.reg) == 0;
}

// ---- cooperative traversal: one ray's primitive list split over the idle lanes of its wave -----------------------
// A pixel's samples are one sequential chain (one RNG stream), so when most lanes of a wave have finished their
// pixels the stragglers set the wave's — and in the end the frame's — finishing time.  When a wave is down to
// <= 32 live lanes, each live ray is handed to a group of G = 64 / 2^ceil(log2 live) lanes: lane j of the group scans
// every G-th hittable of the list (in hittable order), the group merges the G partial winners with the reference's own
// acceptance rule — a later candidate replaces an earlier one iff t is smaller, or equal and its kind accepts
// t == max (rect/box/triangle do, rectangle.hpp:36 triangle.hpp:91; spheres need t < max, sphere.hpp:77) — which is
// exactly what the sequential scan with its shrinking max computes:
//   * every candidate's t is independent of max (sphere: first root > min; box: its own nearest side), max only
//     decides acceptance; so the scan's result is "minimum t; among equal t the last candidate that accepts
//     equality, else the first", a selection that is associative and commutative over any split of the list
//     (other_wins).
//   * constant_medium is the exception (it clamps against max and draws RNG, constant_medium.hpp:52-65): the list is
//     split only up to the first medium (coop_prefix); the rest is scanned after the merge, by every lane of the
//     group redundantly with the owner's RNG state, so it sees exactly the sequential max and draw order.
//   * stale u,v (a triangle/medium hit keeps the u,v of the previously ACCEPTED candidate) depends on acceptance
//     history, which segments do not reproduce: scenes where that value can reach an image texture disable this path
//     (PtScene::coop_ok).  For sphere/rect/box winners u,v are computed from the winner itself.
// Irregular rays (NaN/inf/axis-parallel) never take this path.
struct CoopScene {
  int n_runs;
  int coop_prefix; // hittables before the first constant_medium (== n_hittables when there is none)
};

__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) { // lane index of the n-th (0-based) set bit
  for (int k = 0; k < n; ++k) m &= m - 1;
  return __builtin_ctzll(m);
}

__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl_i(int v, int src) { return __shfl(v, src, 64); }

// The scan's selection for two candidates in ANY list positions (strided splits): smaller t wins; at equal t the scan
// keeps the LAST candidate whose kind accepts t == max if there is one, else the FIRST candidate — so: both accept
// equality -> the later one, exactly one does -> that one, none -> the earlier one.  Record offsets grow in list order.
// Commutative and associative, hence valid for a butterfly over arbitrary disjoint subsets of the list.
__device__ __forceinline__ bool other_wins(float tA, int hitA, float tB, int hitB) { // does B replace A?  (branch-free)
  const bool validA = hitA >= 0, validB = hitB >= 0;
  const bool eqA = hit_kind(hitA) != DK_SPHERE, eqB = hit_kind(hitB) != DK_SPHERE;
  const bool b_later = hit_off(hitB) > hit_off(hitA);
  const bool tie = (eqB & (!eqA | b_later)) | (!eqA & !eqB & !b_later);
  return validB & (!validA | (tB < tA) | ((tB == tA) & tie));
}

// value of lane (lane ^ STEP): DPP inside a row of 16, LDS crossbar beyond.  Steps 4 and 8 use the mirror patterns
// (lane ^ 7, lane ^ 15): equivalent for a butterfly whose earlier steps have already made every aligned group of STEP
// lanes agree.
template <int STEP>
__device__ __forceinline__ int xor_exchange(int v) {
  if constexpr (STEP == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]
  else if constexpr (STEP == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  else if constexpr (STEP == 4) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false); // row_half_mirror
  else if constexpr (STEP == 8) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); // row_mirror
  else return __shfl_xor(v, STEP, 64);
}

// one butterfly stage of the group merge: afterwards both partners hold the better of their two candidates
template <bool IMG, int STEP>
__device__ __forceinline__ void merge_stage(HitState& s) {
  const float tB = as_f(xor_exchange<STEP>(as_i(s.closest)));
  const int hB = xor_exchange<STEP>(s.hit);
  float uB = 0.0f, vB = 0.0f;
  if (IMG) { uB = as_f(xor_exchange<STEP>(as_i(s.u))); vB = as_f(xor_exchange<STEP>(as_i(s.v))); }
  if (other_wins(s.closest, s.hit, tB, hB)) {
    s.closest = tB; s.hit = hB;
    if (IMG) { s.u = uB; s.v = vB; }
  }
}

// The three steps of the dynamic cooperative mode as separate pieces (used by the LDS-tile streaming kernel, whose scan is
// spread over many workgroup-synchronous tiles; hit_world_lds below has them inline).
// group size for a wave with 1 <= nlive <= 32 live rays: G = 64 >> ceil(log2 nlive)
__device__ __forceinline__ int coop_group_log(int nlive) {
  return nlive == 1 ? 6 : 6 - (32 - __builtin_clz((unsigned)(nlive - 1)));
}
// hand every live ray (+ its context) to a group of G lanes: group g serves the g-th live lane; surplus groups shadow ray 0
__device__ __forceinline__ void coop_handoff(RayCtx& c, unsigned long long live_mask, int nlive, int logG) {
  const int group = (threadIdx.x & 63) >> logG;
  const int owner = nth_set_bit(live_mask, group < nlive ? group : 0);
  c.r.o = mk(shfl_f(c.r.o.x, owner), shfl_f(c.r.o.y, owner), shfl_f(c.r.o.z, owner));
  c.r.d = mk(shfl_f(c.r.d.x, owner), shfl_f(c.r.d.y, owner), shfl_f(c.r.d.z, owner));
  c.r.tm = shfl_f(c.r.tm, owner);
  c.a = shfl_f(c.a, owner);
  c.yx = shfl_f(c.yx, owner); c.yy = shfl_f(c.yy, owner); c.yz = shfl_f(c.yz, owner);
  c.reg = true; // cooperative mode is entered only when every live ray is regular
}
// butterfly over the G partial winners of every group, then each owner reads its group's result
template <bool IMG>
__device__ __forceinline__ void coop_merge_handback(HitState& s, unsigned long long live_mask, bool live, int logG) {
  merge_stage<IMG, 1>(s);
  if (logG > 1) merge_stage<IMG, 2>(s);
  if (logG > 2) merge_stage<IMG, 4>(s);
  if (logG > 3) merge_stage<IMG, 8>(s);
  if (logG > 4) merge_stage<IMG, 16>(s);
  if (logG > 5) merge_stage<IMG, 32>(s);
  const int lane = threadIdx.x & 63;
  const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(live_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)live_mask, 0u));
  const int my_leader = live ? (my_rank << logG) : lane;
  s.closest = shfl_f(s.closest, my_leader);
  s.hit = shfl_i(s.hit, my_leader);
  if (IMG) { s.u = shfl_f(s.u, my_leader); s.v = shfl_f(s.v, my_leader); }
}

// hit_world for the LDS-resident kernel, ordinary and cooperative mode in ONE instantiation of the record loops
// (two copies cost ~45 VGPRs = one to two waves of occupancy).  Ordinary mode is the degenerate case G = 1: every
// lane is its own group, its segment is the whole list, nothing is shuffled or merged.
template <bool IMG, typename P>
__device__ __forceinline__ void hit_world_lds(P blob, cst_f4p cblob, const CoopScene& cs, const Ray& my_ray, uint32_t& my_rng, bool live,
                                              bool coop_allowed, bool scene_fast_ok, int wide_logG, HitState& h) {
  RayCtx c = make_ctx(my_ray, scene_fast_ok);
  const bool fast = wave_all_regular(c, live);
  const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
  const int nlive = __builtin_popcountll(live_mask);
  const int lane = threadIdx.x & 63;
  // Two ways to be cooperative (both wave-uniform):
  //  * wide (static groups, render_kernel's split-queue phase): every aligned group of G lanes already holds the SAME
  //    pixel — ray, RNG state, everything — computed redundantly, so nothing is handed over and everybody keeps the result;
  //  * dynamic: the wave is down to <= 32 live lanes; each live ray is handed to a group of idle lanes and handed back.
  int logG = 0;
  bool handoff = false;
  if (wide_logG) {
    if (fast) logG = wide_logG; // irregular ray somewhere: every lane scans the whole list (still identical per group)
  } else if (coop_allowed && fast && nlive >= 1 && nlive <= 32) {
    logG = nlive == 1 ? 6 : 6 - (32 - __builtin_clz((unsigned)(nlive - 1))); // G = 64 >> ceil(log2 nlive)
    handoff = true;
  }
  const int G = 1 << logG;
  const int group = lane >> logG, j = lane & (G - 1);
  uint32_t rng = my_rng;
  if (handoff) { // hand every live ray (+ its context and RNG state) to a group of G lanes
    const int owner = nth_set_bit(live_mask, group < nlive ? group : 0); // surplus groups shadow ray 0 (outcome unused)
    c.r.o = mk(shfl_f(c.r.o.x, owner), shfl_f(c.r.o.y, owner), shfl_f(c.r.o.z, owner));
    c.r.d = mk(shfl_f(c.r.d.x, owner), shfl_f(c.r.d.y, owner), shfl_f(c.r.d.z, owner));
    c.r.tm = shfl_f(c.r.tm, owner);
    c.a = shfl_f(c.a, owner);
    c.yx = shfl_f(c.yx, owner); c.yy = shfl_f(c.yy, owner); c.yz = shfl_f(c.yz, owner);
    c.reg = true; // fast == every live ray is regular
    rng = (uint32_t)shfl_i((int)rng, owner);
  }
  c.live = handoff ? (group < nlive) : live;
  HitState s;
  hit_begin(s);
  bool merged = logG == 0;
  // One pass over the runs.  Up to the first constant_medium lane j of a group tests the hittables whose list index is
  // == j (mod G) (strided: every run, however short, is spread over the group); at that point (or after the last run)
  // the G partial winners are merged and from then on every lane of the group scans whole runs with the merged state and
  // the ray's RNG state.
  for (int ri = 0; ri <= cs.n_runs; ++ri) {
    int kind = DK_MEDIUM, off = 0, cnt = 0, first = 0;
    if (ri < cs.n_runs) {
      f4 runf = blob[ri];
      kind = as_i(runf.x); off = as_i(runf.y); cnt = as_i(runf.z); first = as_i(runf.w);
      if (kind & DK_ABSORBED) continue; // (an absorbed sphere run: the absorbing run's lists — which the strided scan splits too — hold its spheres)
    }
    if (!merged && kind == DK_MEDIUM) { // butterfly: afterwards all G lanes hold the group's winner
      merge_stage<IMG, 1>(s);
      if (logG > 1) merge_stage<IMG, 2>(s);
      if (logG > 2) merge_stage<IMG, 4>(s);
      if (logG > 3) merge_stage<IMG, 8>(s);
      if (logG > 4) merge_stage<IMG, 16>(s);
      if (logG > 5) merge_stage<IMG, 32>(s);
      merged = true;
    }
    if (ri == cs.n_runs) break;
    if (cnt <= 0) continue;
    // (idle lanes scan too and their outcome is dropped: a per-lane skip would put the whole scan under exec-mask
    // branches — the ordinary kernels do the same)
    if (merged) hit_records<IMG, 2, 2, true, false, false>(blob + off, cblob, kind, cnt, off, c, fast, rng, s); // (no grid walk here: lists + groups)
    else hit_records_strided<IMG>(blob + off, cblob, kind, cnt, first, off, j, logG, c, s);
  }
  if (handoff) { // hand each owner its result: the r-th live lane reads from (a lane of) group r
    const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(live_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)live_mask, 0u));
    const int my_leader = live ? (my_rank << logG) : lane;
    s.closest = shfl_f(s.closest, my_leader);
    s.hit = shfl_i(s.hit, my_leader);
    if (IMG) { s.u = shfl_f(s.u, my_leader); s.v = shfl_f(s.v, my_leader); }
    rng = (uint32_t)shfl_i((int)rng, my_leader);
  }
  h = s;
  if (live) my_rng = rng;
}


// ---- hit_record of the final nearest hit (hitable.hpp:8-24) ---------------------------------------------------
struct Rec {
  V3 p, normal;
  bool front_face;
  int mat;
  int hittable;
};

__device__ __forceinline__ void set_face_normal(Rec& rec, const Ray& r, V3 n) {
  rec.front_face = dot(r.d, n) < 0;
  rec.normal = rec.front_face ? n : mk(0.0f, 0.0f, 0.0f) - n;
}

// RECTBOX (compile-time; MATS bit 16): the scene's hittables are rects and boxes only — the headline Cornell-style family — so a hit is
// one of those two kinds and the sphere / triangle / medium branches (three exec-mask branches a wave walks past on every hit) are not
// compiled in.
template <bool RECTBOX = false, typename P>
__device__ __forceinline__ Rec resolve_hit(P blob, int hit, const Ray& r, float t) {
  Rec rec;
  rec.p = r.o + t * r.d; // ray::at ray.hpp:21
  const int off = hit_off(hit);
  const int kind = hit_kind(hit);
  f4 R0 = blob[off], R1 = blob[off + 1];
  if constexpr (RECTBOX) {
    int axis;
    if (kind == DK_RECT) { axis = as_i(R1.z); rec.mat = as_i(R1.y); rec.hittable = as_i(R1.w); }
    else { axis = hit_side(hit) >> 1; rec.mat = as_i(R0.w); rec.hittable = as_i(R1.w); }
    V3 n = axis == 0 ? mk(0, 0, 1) : axis == 1 ? mk(0, 1, 0) : mk(1, 0, 0);
    set_face_normal(rec, r, n);
    return rec;
  }
  if (kind == DK_SPHERE) {
    f4 R2 = blob[off + 2];
    V3 n = (rec.p - sphere_center(R0, R1, R2, r.tm)) / R1.x;
    set_face_normal(rec, r, n);
    rec.mat = as_i(R1.y);
    rec.hittable = as_i(R2.w);
  } else if (kind == DK_RECT || kind == DK_BOX) {
    int axis;
    if (kind == DK_RECT) { axis = as_i(R1.z); rec.mat = as_i(R1.y); rec.hittable = as_i(R1.w); }
    else { axis = hit_side(hit) >> 1; rec.mat = as_i(R0.w); rec.hittable = as_i(R1.w); }
    V3 n = axis == 0 ? mk(0, 0, 1) : axis == 1 ? mk(0, 1, 0) : mk(1, 0, 0);
    set_face_normal(rec, r, n);
  } else if (kind == DK_TRI || kind == DK_TRI_B) {
    f4 R2 = blob[off + 2];
    set_face_normal(rec, r, cross(xyz(R1), xyz(R2))); // not normalised, triangle.hpp:96 (Badouel: the same cross(u, v), :21,52)
    rec.mat = as_i(R0.w);
    rec.hittable = as_i(R1.w);
  } else {
    rec.normal = mk(1, 0, 0); // constant_medium.hpp:75-76
    rec.front_face = true;
    rec.mat = as_i(R0.z);
    rec.hittable = as_i(R0.w);
  }
  return rec;
}

// u,v of the final hit, from the hit itself (sphere.hpp:88-89 mercator of the face normal; rectangle.hpp:41-42 and its
// xz/yz twins; box = its hit side).  The reference fills them on EVERY accepted candidate; for sphere, rect and box hits
// they are a pure function of (ray, primitive, t), so the value of the final hit is the same computed once here.
// Triangles and constant_media never write u,v (theirs are whatever an earlier accepted candidate left behind): scenes
// where such a stale value could reach an image texture use the kernels that track u,v through the scan (IMG = true).
template <typename P>
__device__ __forceinline__ void winner_uv(P blob, int hit, const Ray& r, float t, const Rec& rec, float& u, float& v) {
  const int off = hit_off(hit), kind = hit_kind(hit);
  u = 0.0f; v = 0.0f;
  if (kind == DK_SPHERE) {
    mercator(rec.normal, u, v);
  } else if (kind == DK_RECT || kind == DK_BOX) {
    const f4 R0 = blob[off], R1 = blob[off + 1];
    int axis;
    float a0, a1, b0, b1;
    if (kind == DK_RECT) { axis = as_i(R1.z); a0 = R0.x; a1 = R0.y; b0 = R0.z; b1 = R0.w; }
    else {
      axis = hit_side(hit) >> 1; // sides 0,1: xy  2,3: xz  4,5: yz (PT_BOX_SIDES)
      a0 = axis == 2 ? R0.y : R0.x; a1 = axis == 2 ? R1.y : R1.x;
      b0 = axis == 0 ? R0.y : R0.z; b1 = axis == 0 ? R1.y : R1.z;
    }
    const float oa = axis == 2 ? r.o.y : r.o.x, da = axis == 2 ? r.d.y : r.d.x;
    const float ob = axis == 0 ? r.o.y : r.o.z, db = axis == 0 ? r.d.y : r.d.z;
    const float a = oa + t * da, b = ob + t * db;
    u = (a - a0) / (a1 - a0);
    v = (b - b0) / (b1 - b0);
  }
}

// ---- textures: texture.hpp:25, 42-49, 135-151 -------------------------------------------------------------------
// material record (global memory, 4 f4): M0 = (mat kind, tex kind, param, freq)  M1 = (color0.xyz, width)
//                                        M2 = (color1.xyz, height)  M3 = (atlas offset, 0, 0, 0)
__device__ __forceinline__ uint32_t texel_index(float f, uint32_t maxv) {
  if (!(f > 0.0f)) return 0;
  if (f >= (float)maxv) return maxv;
  return (uint32_t)f;
}

// MATS (compile-time, kernels specialised on what a scene contains): bit k = material kind k may occur (material.hpp:133-135
// order: lambertian 0, metal 1, dielectric 2, lightsource 3, isotropic 4); bit 8 = a texture other than solid_texture may occur.
// The generic kernels pass MATS_ALL; a scene of lambertian + lightsource materials over solid textures (the Cornell-style
// headline scene) runs kernels compiled with MATS_LAMB_LIGHT_SOLID, which carry none of the other branches.
enum { MATS_ALL = 0x11f, MATS_LAMB_LIGHT_SOLID = 0x009, MATS_RECTBOX_ONLY = 0x10000 /* + every hittable is a rect or a box (resolve_hit) */ };

// texture.hpp:43-45: `sin(a) sin(b) sin(c) < 0` (a, b, c = 10 p).  Only the SIGN of the product is looked at, and for regular arguments
// (2^-30 <= |.| < 2^30: every factor is then non-zero and at least 2^-30, so the binary32 product of three cannot underflow and its sign is
// the product of the signs) the sign of a factor follows from its range reduction alone (pt_math.hpp: sin_negative_regular) — 6 binary64
// operations per sine instead of two degree-6 polynomials (38).  A lane with an argument outside that range (0, denormal-small, huge, NaN:
// the product may then be a zero or a NaN, never "< 0" by sign alone) evaluates the product itself.  Measured: profiles/r04_ab_checker.txt.
__device__ __forceinline__ bool checker_sines_negative(float a, float b, float c) {
#ifndef PT_NO_CHECKER_SHORTCUT
  const float lo = 9.31322574615478515625e-10f, hi = 1073741824.0f; // 2^-30, 2^30
  const float mn = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
  const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
  const bool nan = (a != a) | (b != b) | (c != c); // (fmin / fmax drop a NaN operand)
  if (!nan && mn >= lo && mx < hi) {
#ifndef PT_NO_CHECKER_F32
    // First in binary32: u = |x| fl(1/pi) is within |x| 2^-23 / pi of |x| / pi (the constant's and the product's rounding), so where u lies
    // at least d = u 2^-21 from both neighbouring integers, floor(|x| / pi) = floor(u) and the sine's sign is (-1)^floor(u) sign(x) — the sign
    // of the true sine and, that far from a zero of it, of sinf_'s (tests/cpp/checker_sign_exhaustive.c checks every regular binary32
    // argument for which this form decides).  Arguments past ~2^21 never decide here (d >= 1/2) and take the binary64 reduction below.
    bool neg = false, decided = true;
    const float xs[3] = {a, b, c};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float u = __builtin_fabsf(xs[k]) * 0.318309886183790671538f;
      const float fl = __builtin_floorf(u), fr = u - fl, d = u * 4.76837158203125e-07f; // 2^-21
      decided = decided & (fr >= d) & (fr <= 1.0f - d);
      neg = neg != ((((int)fl & 1) != 0) != (xs[k] < 0.0f));
    }
    if (decided) return neg;
#endif
    return (ptm::sin_negative_regular(a) != ptm::sin_negative_regular(b)) != ptm::sin_negative_regular(c);
  }
#endif
  const float sines = ptm::sinf_(a) * ptm::sinf_(b) * ptm::sinf_(c);
  return sines < 0;
}

template <int MATS = MATS_ALL, typename UV>
__device__ __forceinline__ V3 texture_value(f4 M0, f4 M1, f4 M2, f4 M3, V3 p, UV uv, const uint8_t* __restrict__ atlas) {
  if constexpr (!(MATS & 0x100)) return xyz(M1); // every texture of the scene is a solid_texture (texture.hpp:25)
  const int tk = as_i(M0.y);
  if (tk == 1) return xyz(M1); // solid
  if (tk == 0) return checker_sines_negative(10.0f * p.x, 10.0f * p.y, 10.0f * p.z) ? xyz(M1) : xyz(M2); // checker
  const uint32_t w = (uint32_t)as_i(M1.w), h = (uint32_t)as_i(M2.w), offset = (uint32_t)as_i(M3.x);
  uint32_t i = 0, j = 0;
  bool have = false;
  if constexpr (requires(V3& nn) { uv.sphere_normal(nn); }) { // a sphere's hit: the texel straight from the normal where that is unambiguous
    V3 nn;
    if (uv.sphere_normal(nn)) have = sphere_texel_fast(nn, M0.w, w, h, i, j);
  }
  if (!have) {
    float u, v;
    uv(u, v); // only image textures look at u,v
    i = texel_index(ptm::fmod1f_(u * M0.w) * (float)(w - 1), w - 1);
    j = texel_index((1.0f - ptm::fmod1f_(v * M0.w)) * (float)(h - 1), h - 1);
  }
  uint64_t pix = (uint64_t)j * w + i + offset;
  const float scale = 1.0f / 255;
  return mk((float)atlas[pix * 3] * scale, (float)atlas[pix * 3 + 1] * scale, (float)atlas[pix * 3 + 2] * scale);
}

// vec.hpp:26
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
// vec.hpp:29-35
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float etai_over_etat) {
  float cos_theta = __builtin_fminf(-dot(uv, n), 1.0f);
  V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
  V3 r_out_parallel = (-sqrt_rn(__builtin_fabsf(1.0f - length_squared(r_out_perp)))) * n;
  return r_out_perp + r_out_parallel;
}
// material.hpp:62-66
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 *= r0;
  return r0 + (1.0f - r0) * ptm::pow5f_(1.0f - cosine);
}

// render.hpp:83-87
// regular (wave-uniform): every live lane's ray is regular (RayCtx: 2^-40 <= |d_c| <= 2^40), so a = d.d lies in [3 * 2^-80, 3 * 2^80] and
// only unit_vector(d).y = d.y / sqrt(a) is needed (render.hpp:84-85 reads .y() only): the square root through the hardware estimate +
// neighbour test (sqrt_rn_unit's form: scale-invariant under x -> 4x, so what is proved for EVERY float of [2^-60, 4] holds for every
// normal x whose residuals stay normal), the quotient through ONE correctly rounded reciprocal + div_exact's correction (no intermediate
// leaves the normal range: |q| in [2^-81, 1]).  17 + 12 -> 9 + 8 issue slots, once per iteration (some lane of 64 nearly always misses
// everything).  Same bits as the IEEE forms: tests/test_gpu_parity.py::test_sky_unit_direction_shortcut_is_exact.
__device__ __forceinline__ float sky_unit_y(float dy, float a, bool regular) {
#ifdef PT_NO_SKY_SHORTCUT /* A/B build */
  regular = false;
#endif
  if (regular) {
    const float len = sqrt_rn_unit(a);
    const float y = rcp_rn_guarded(len);
    return div_exact(dy, len, y, dy * y);
  }
  asm volatile("" : "+v"(a)); // (opaque: the general square root is not to be hoisted out of this rarely taken path)
  return dy / sqrt_rn(a);
}
// unit_vector(d) (vec.hpp: v / v.length()) for metal and dielectric scatter (material.hpp:41,74): all three quotients through sky_unit_y's
// form when every live ray of the wave is regular — one range-proved square root and one correctly rounded reciprocal shared by three
// corrected quotients (9 + 4 + 15 issue slots) instead of the general square root and three IEEE divisions (17 + 39); the same function of
// (d_c, d.d) as the sky's, whose test covers every component by symmetry (test_sky_unit_direction_shortcut_is_exact).
__device__ __forceinline__ V3 unit_direction(V3 d, bool regular) {
  float a = dot(d, d);
#ifndef PT_NO_UD_SHORTCUT
  if (regular) {
    const float len = sqrt_rn_unit(a);
    const float y = rcp_rn_guarded(len);
    return mk(div_exact(d.x, len, y, d.x * y), div_exact(d.y, len, y, d.y * y), div_exact(d.z, len, y, d.z * y));
  }
#endif
  asm volatile("" : "+v"(a)); // (as in sky_unit_y: keep the general square root inside the rarely taken path)
  return d / sqrt_rn(a);
}
__device__ __forceinline__ V3 sky_color(const Ray& r, V3 att, bool regular = false) {
  V3 ud = mk(0.0f, sky_unit_y(r.d.y, dot(r.d, r.d), regular), 0.0f); // (x and z of the unit vector are never read)
  float hit_pt = 0.5f * (ud.y + 1.0f);
  V3 c = (1.0f - hit_pt) * mk(1.0f, 1.0f, 1.0f) + hit_pt * mk(0.5f, 0.7f, 1.0f);
  return att * c;
}

// One iteration of the bounce loop render.hpp:58-89 after hit_world: emitted + scatter.
// Returns true if the path continues (ray/att updated); false if it ended with `out`.
// `uv(u, v)` yields the hit's texture coordinates; it is called before the ray is overwritten.
template <int MATS = MATS_ALL, typename PM, typename UV>
__device__ __forceinline__ bool shade(PM mats, const uint8_t* __restrict__ atlas, const Rec& rec,
                                      UV uv, Ray& ray, V3& att, uint32_t& rng, V3& out, bool regular = false) {
  PM M = mats + rec.mat * SZ_MATERIAL;
  f4 M0 = M[0], M1 = M[1];
  const int mk_ = as_i(M0.x);
  constexpr bool LAMB = MATS & 1, METAL = (MATS >> 1) & 1, GLASS = (MATS >> 2) & 1, LIGHT = (MATS >> 3) & 1, ISO = (MATS >> 4) & 1;
  // A wave holds lanes of every material, and each `if` below runs once for the lanes that take it.  What several materials
  // need — the unit direction of the incoming ray (metal, glass), a point in the unit ball (metal, isotropic: three draws and
  // four transcendentals), the texture's value (lambertian, light, isotropic) — is therefore computed ONCE, for the union of
  // the lanes that need it, ahead of the switch: one pass through that code per wave-iteration instead of one per
  // material.  Per lane nothing moves: each lane takes exactly one material, its draws keep their order (the ball is the
  // first thing metal and isotropic draw; texture values draw nothing).
  V3 ud = mk(0.0f, 0.0f, 0.0f), ball = ud, tv = ud;
  if constexpr (METAL || GLASS) { if (mk_ == 1 || mk_ == 2) ud = unit_direction(ray.d, regular); } // (regular: wave-uniform)
  if constexpr (METAL || ISO) { if (mk_ == 1 || mk_ >= 4) ball = rng_in_unit_ball(rng); }
  if constexpr (!(MATS & 0x100)) tv = xyz(M1); // solid textures only: every material's colour slot IS its texture value
  else if (mk_ == 0 || mk_ >= 3) tv = texture_value<MATS>(M0, M1, M[2], M[3], rec.p, uv, atlas);
  if (LAMB && (mk_ == 0 || !(METAL || GLASS || LIGHT || ISO))) { // lambertian material.hpp:18-28
    V3 dir = rec.normal + rng_unit_vec(rng);
    ray.o = rec.p; ray.d = dir;
    att = att * tv;
    return true;
  }
  if constexpr (METAL) {
    if (mk_ == 1) { // metal material.hpp:39-48
      V3 reflected = reflect(ud, rec.normal);
      V3 dir = reflected + M0.z * ball;
      ray.o = rec.p; ray.d = dir;
      att = att * xyz(M1);
      if (dot(dir, rec.normal) > 0) return true;
      out = mk(0.0f, 0.0f, 0.0f); // emitted of a non-light (material.hpp:50)
      return false;
    }
  }
  if constexpr (GLASS) {
    if (mk_ == 2) { // dielectric material.hpp:68-88
      att = att * xyz(M1);
      float ref_idx = M0.z;
      float ratio = rec.front_face ? (1.0f / ref_idx) : ref_idx;
      float cos_theta = __builtin_fminf(-dot(ud, rec.normal), 1.0f);
      float sin_theta = sqrt_rn(1.0f - cos_theta * cos_theta);
      bool cannot_refract = ratio * sin_theta > 1.0f;
      V3 dir;
      if (cannot_refract || reflectance(cos_theta, ratio) > rng_float(rng)) dir = reflect(ud, rec.normal);
      else dir = refract(ud, rec.normal, ratio);
      ray.o = rec.p; ray.d = dir;
      return true;
    }
  }
  if (LIGHT && (mk_ == 3 || !ISO)) { // lightsource material.hpp:104-108; returned un-attenuated (render.hpp:73)
    out = tv;
    return false;
  }
  // isotropic material.hpp:119-126
  ray.o = rec.p; ray.d = ball;
  att = att * tv;
  return true;
}

} // namespace ptd
