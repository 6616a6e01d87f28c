// pt_math.hpp — device transcendental set of the render kernel (gfx950).
//
// The reference calls sycl::sin/cos/log/pow/atan2/asin/fmod
// (rtweekend.hpp:75-79, texture.hpp:43-44,140-143, material.hpp:65,
// sphere.hpp:15-17, constant_medium.hpp:65).  A path tracer is chaotic in its
// RNG stream: one differing ulp flips a branch and re-rolls the rest of that
// pixel, so "whatever libm the platform has" cannot give
// reproducible pixels across hosts, let alone CPU vs GPU.  These functions are
// therefore DEFINED by this project: evaluated in binary64 with only
// IEEE-exact operations (+ - * / sqrt fma rint, bit ops), rounded once to
// binary32.  Same published algorithms (Sun fdlibm lineage: k_sin, k_cos,
// medium-range rem_pio2, s_atan, e_atan2, atanh-series log) and therefore the
// same bits on any IEEE machine; they land within 1 ulp of glibc's float
// functions (tests/test_math_cpu.py states the measured bound; tests/test_gpu_parity.py::test_math_bit_exact checks GPU == oracle copy).
//
// Rare in the instruction mix (shading only, never in the primitive loop), so
// fp64 throughput is not a concern.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptm {

__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ double dabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ bool disinf(double x) { return dabs(x) == __builtin_inf(); }

__device__ __forceinline__ double ksin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  return r + r * (z * p);
}

__device__ __forceinline__ double kcos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  return (1.0 - 0.5 * z) + (z * z) * p;
}

// r = x - n*pi/2 ; returns n mod 4.  Domain |x| < 2^30.
__device__ __forceinline__ int rem_pio2(double x, double& r) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;
  const double PIO2_1T = 6.07710050650619224932e-11;
  double fn = __builtin_rint(x * INV_PIO2);
  double t = __builtin_fma(-fn, PIO2_1, x);
  r = __builtin_fma(-fn, PIO2_1T, t);
  return (int)((int64_t)fn & 3);
}

__device__ __forceinline__ float sinf_(float xf) {
  double x = (double)xf;
  if (!(dabs(x) < 1073741824.0)) {
    if (x != x || disinf(x)) return (float)(x - x);
    return 0.0f;
  }
  double r;
  int n = rem_pio2(x, r);
  double s = ksin(r), c = kcos(r);
  double v = (n & 1) ? c : s;
  return (float)((n & 2) ? -v : v);
}

// The SIGN BIT of sinf_(xf) for a regular argument, 2^-30 <= |xf| < 2^30, from the range reduction alone: sinf_ rounds +-ksin(r) (n even:
// the sign of r, r != 0 there) or +-kcos(r) (n odd: positive) to binary32, negated when n & 2.  For every such binary32 argument the
// result is non-zero, at least 2^-30 in magnitude, and has this sign: tests/cpp/checker_sign_exhaustive.c visits all 1.0 x 10^9 of them
// against the oracle's copy of sinf_.  (What the checker texture needs: texture.hpp:43-45 looks at the sign of a product of three sines.)
__device__ __forceinline__ bool sin_negative_regular(float xf) {
  double r;
  const int n = rem_pio2((double)xf, r);
  return ((n & 1) ? false : (r < 0.0)) != ((n & 2) != 0);
}

__device__ __forceinline__ float cosf_(float xf) {
  double x = (double)xf;
  if (!(dabs(x) < 1073741824.0)) {
    if (x != x || disinf(x)) return (float)(x - x);
    return 1.0f;
  }
  double r;
  int n = rem_pio2(x, r);
  double s = ksin(r), c = kcos(r);
  double v = (n & 1) ? s : c;
  // n: 0 -> c, 1 -> -s, 2 -> -c, 3 -> s
  return (float)(((n + 1) & 2) ? -v : v);
}

// sinf_(xf) and cosf_(xf) from ONE range reduction and one evaluation of each kernel polynomial: the same operations in the same order
// as the two functions above, hence the same bits (tests/test_gpu_parity.py::test_math_bit_exact, ops 14 / 15).  The two calls side by side
// do not fuse by themselves — each carries its own out-of-range branch — and rng_in_unit_ball asks for both of two angles: 154 -> 72
// binary64 instructions per point in the unit ball (metal fuzz, isotropic scatter).
__device__ __forceinline__ void sincosf_(float xf, float& sn, float& cs) {
  double x = (double)xf;
  if (!(dabs(x) < 1073741824.0)) {
    if (x != x || disinf(x)) { sn = (float)(x - x); cs = sn; return; }
    sn = 0.0f; cs = 1.0f;
    return;
  }
  double r;
  int n = rem_pio2(x, r);
  double s = ksin(r), c = kcos(r);
  double vs = (n & 1) ? c : s;
  double vc = (n & 1) ? s : c;
  sn = (float)((n & 2) ? -vs : vs);
  cs = (float)(((n + 1) & 2) ? -vc : vc);
}

__device__ __forceinline__ float logf_(float xf) {
  double x = (double)xf;
  if (x != x) return xf;
  if (x < 0.0) return __builtin_nanf("");
  if (x == 0.0) return -__builtin_inff();
  if (disinf(x)) return __builtin_inff();
  uint64_t b = d2u(x);
  int e = (int)((b >> 52) & 0x7ff) - 1023;
  double m = u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double p = 4.76190476190476164086e-02;
  p = 5.26315789473684181083e-02 + z * p;
  p = 5.88235294117647050874e-02 + z * p;
  p = 6.66666666666666657415e-02 + z * p;
  p = 7.69230769230769273497e-02 + z * p;
  p = 9.09090909090909116141e-02 + z * p;
  p = 1.11111111111111104943e-01 + z * p;
  p = 1.42857142857142849213e-01 + z * p;
  p = 2.00000000000000011102e-01 + z * p;
  p = 3.33333333333333314830e-01 + z * p;
  p = 1.0 + z * p;
  double lm = 2.0 * s * p;
  return (float)((double)e * 6.93147180559945286227e-01 + lm);
}

// pow(x, 5.0f) — the only pow call site is material.hpp:65
__device__ __forceinline__ float pow5f_(float xf) {
  double x = (double)xf;
  double x2 = x * x;
  double x4 = x2 * x2;
  return (float)(x4 * x);
}

__device__ __forceinline__ double atan_pos(double ax) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
               hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
               lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  if (ax >= 7.3786976294838206464e19) return hi3 + lo3;
  int id;
  double hi = 0.0, lo = 0.0, t;
  if (ax < 0.4375) {
    if (ax < 7.450580596923828125e-09) return ax;
    id = -1; t = ax;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) { id = 0; t = (2.0 * ax - 1.0) / (2.0 + ax); hi = hi0; lo = lo0; }
    else             { id = 1; t = (ax - 1.0) / (ax + 1.0);       hi = hi1; lo = lo1; }
  } else {
    if (ax < 2.4375) { id = 2; t = (ax - 1.5) / (1.0 + 1.5 * ax); hi = hi2; lo = lo2; }
    else             { id = 3; t = -1.0 / ax;                     hi = hi3; lo = lo3; }
  }
  double z = t * t, w = z * z;
  double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) return t - t * (s1 + s2);
  return hi - ((t * (s1 + s2) - lo) - t);
}

__device__ __forceinline__ double atan2d(double y, double x) {
  const double PI = 3.14159265358979311600e+00, PI_LO = 1.22464679914735317720e-16;
  const double PIO2 = 1.57079632679489655800e+00, PIO4 = 7.85398163397448278999e-01;
  if (x != x || y != y) return x + y;
  int sy = (int)(d2u(y) >> 63), sx = (int)(d2u(x) >> 63);
  int m = sy + 2 * sx;
  if (y == 0.0) {
    if (m < 2) return y;
    return m == 2 ? PI : -PI;
  }
  if (x == 0.0) return sy ? -PIO2 : PIO2;
  double ax = dabs(x), ay = dabs(y);
  if (disinf(ax)) {
    if (disinf(ay)) {
      double q = (m & 2) ? 3.0 * PIO4 : PIO4;
      return (m & 1) ? -q : q;
    }
    double q = (m & 2) ? PI : 0.0;
    return (m & 1) ? -q : q;
  }
  if (disinf(ay)) return sy ? -PIO2 : PIO2;
  double z = atan_pos(ay / ax);
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return PI - (z - PI_LO);
    default: return (z - PI_LO) - PI;
  }
}

__device__ __forceinline__ float atan2f_(float y, float x) { return (float)atan2d((double)y, (double)x); }

__device__ __forceinline__ float asinf_(float xf) {
  double x = (double)xf;
  if (x != x) return xf;
  double c = ::sqrt((1.0 - x) * (1.0 + x));
  return (float)atan2d(x, c);
}

// fmod(x, 1.0f) — texture.hpp:140,143
__device__ __forceinline__ float fmod1f_(float x) {
  if (x != x || __builtin_fabsf(x) == __builtin_inff()) return x - x;
  float r = x - __builtin_truncf(x);
  return __builtin_copysignf(r, x);
}

} // namespace ptm
