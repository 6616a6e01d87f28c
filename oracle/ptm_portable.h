/*
 * ptm_portable.h — ORACLE-SIDE copy of the project's portable transcendental set.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Nothing under
 * path_tracer_amd/ includes this file; the HIP kernels carry their own
 * restatement of the same published algorithms in csrc/pt_math.hpp, and
 * tests/test_gpu_parity.py::test_math_bit_exact checks the two bit-for-bit on the GPU.
 *
 * Why it exists: the reference calls sycl::sin/cos/log/pow/atan2/asin/fmod
 * (rtweekend.hpp:75-79, texture.hpp:43-44,140-143, material.hpp:65,
 * sphere.hpp:15-17, constant_medium.hpp:65) which on triSYCL's host device
 * resolve to the platform libm.  libm results differ by ulps between glibc
 * builds (glibc selects FMA/non-FMA sinf/cosf/logf/powf variants per CPU) and
 * from any GPU library, and one flipped ulp re-rolls a pixel's whole RNG
 * stream (SURVEY.md §7 "chaotic sensitivity").  So the project pins ONE
 * definition of these seven functions, built only from IEEE-exact double
 * operations (+ - * / sqrt fma rint, integer bit ops), which gives identical
 * bits on x86-64 and on gfx950.  The oracle can run with either this set
 * (orc_set_math(1), bit-comparable with the GPU) or glibc (orc_set_math(0),
 * the reference's own semantics on this host).
 *
 * Algorithms restated (public, Sun fdlibm 5.3 lineage — k_sin.c, k_cos.c,
 * e_rem_pio2.c medium case, s_atan.c, e_atan2.c, e_log.c's atanh series):
 * evaluate in binary64, round once to binary32.
 */
#ifndef PTM_PORTABLE_H
#define PTM_PORTABLE_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t ptm_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double ptm_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* ---- sin / cos -------------------------------------------------------------- */

static inline double ptm_ksin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
  return r + r * (z * p);
}

static inline double ptm_kcos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
  return (1.0 - 0.5 * z) + (z * z) * p;
}

/* r = x - n*pi/2, |r| <= pi/4 (+rounding); returns n mod 4.  Domain |x| < 2^30. */
static inline int ptm_rem_pio2(double x, double* r) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
  double fn = rint(x * INV_PIO2);
  double t = fma(-fn, PIO2_1, x);
  *r = fma(-fn, PIO2_1T, t);
  return (int)((int64_t)fn & 3);
}

static inline float ptm_sinf(float xf) {
  double x = (double)xf;
  if (!(fabs(x) < 1073741824.0)) {       /* NaN, inf, or outside the pinned domain */
    if (x != x || fabs(x) == INFINITY) return (float)(x - x);
    return 0.0f;
  }
  double r;
  int n = ptm_rem_pio2(x, &r);
  double v;
  switch (n) {
    case 0: v = ptm_ksin(r); break;
    case 1: v = ptm_kcos(r); break;
    case 2: v = -ptm_ksin(r); break;
    default: v = -ptm_kcos(r); break;
  }
  return (float)v;
}

static inline float ptm_cosf(float xf) {
  double x = (double)xf;
  if (!(fabs(x) < 1073741824.0)) {
    if (x != x || fabs(x) == INFINITY) return (float)(x - x);
    return 1.0f;
  }
  double r;
  int n = ptm_rem_pio2(x, &r);
  double v;
  switch (n) {
    case 0: v = ptm_kcos(r); break;
    case 1: v = -ptm_ksin(r); break;
    case 2: v = -ptm_kcos(r); break;
    default: v = ptm_ksin(r); break;
  }
  return (float)v;
}

/* ---- log ---------------------------------------------------------------------- */

static inline float ptm_logf(float xf) {
  double x = (double)xf;
  if (x != x) return xf;
  if (x < 0.0) return (float)((x - x) / 0.0); /* NaN */
  if (x == 0.0) return -INFINITY;
  if (x == INFINITY) return INFINITY;
  uint64_t b = ptm_d2u(x); /* every finite positive float is a normal double */
  int e = (int)((b >> 52) & 0x7ff) - 1023;
  double m = ptm_u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
  if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  /* atanh series 2*s*(1 + z/3 + z^2/5 + ... + z^10/21), |s| <= 0.1716 */
  double p = 4.76190476190476164086e-02;        /* 1/21 */
  p = 5.26315789473684181083e-02 + z * p;       /* 1/19 */
  p = 5.88235294117647050874e-02 + z * p;       /* 1/17 */
  p = 6.66666666666666657415e-02 + z * p;       /* 1/15 */
  p = 7.69230769230769273497e-02 + z * p;       /* 1/13 */
  p = 9.09090909090909116141e-02 + z * p;       /* 1/11 */
  p = 1.11111111111111104943e-01 + z * p;       /* 1/9  */
  p = 1.42857142857142849213e-01 + z * p;       /* 1/7  */
  p = 2.00000000000000011102e-01 + z * p;       /* 1/5  */
  p = 3.33333333333333314830e-01 + z * p;       /* 1/3  */
  p = 1.0 + z * p;
  double lm = 2.0 * s * p;
  return (float)((double)e * 6.93147180559945286227e-01 + lm);
}

/* ---- pow(x, 5.0f): the only pow call site is material.hpp:65 ------------------- */

static inline float ptm_pow5f(float xf) {
  double x = (double)xf;
  double x2 = x * x;
  double x4 = x2 * x2;
  return (float)(x4 * x);
}

/* ---- atan / atan2 / asin ------------------------------------------------------- */

static inline double ptm_atan_pos(double ax) { /* ax >= 0, finite or inf, not NaN */
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
  if (ax >= 7.3786976294838206464e19) return hi3 + lo3; /* 2^66 */
  int id;
  double hi = 0.0, lo = 0.0, t;
  if (ax < 0.4375) {
    if (ax < 7.450580596923828125e-09) return ax; /* 2^-27 */
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

static inline double ptm_atan2d(double y, double x) {
  const double PI = 3.14159265358979311600e+00, PI_LO = 1.22464679914735317720e-16;
  const double PIO2 = 1.57079632679489655800e+00, PIO4 = 7.85398163397448278999e-01;
  if (x != x || y != y) return x + y;
  int sy = (int)(ptm_d2u(y) >> 63), sx = (int)(ptm_d2u(x) >> 63);
  int m = sy + 2 * sx;
  if (y == 0.0) {
    switch (m) { case 0: case 1: return y; case 2: return PI; default: return -PI; }
  }
  if (x == 0.0) return sy ? -PIO2 : PIO2;
  double ax = fabs(x), ay = fabs(y);
  if (ax == INFINITY) {
    if (ay == INFINITY) {
      switch (m) { case 0: return PIO4; case 1: return -PIO4;
                   case 2: return 3.0 * PIO4; default: return -3.0 * PIO4; }
    }
    switch (m) { case 0: return 0.0; case 1: return -0.0; case 2: return PI; default: return -PI; }
  }
  if (ay == INFINITY) return sy ? -PIO2 : PIO2;
  double z = ptm_atan_pos(ay / ax); /* inputs are floats widened: no over/underflow of the quotient to worry about beyond inf->pi/2 */
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return PI - (z - PI_LO);
    default: return (z - PI_LO) - PI;
  }
}

static inline float ptm_atan2f(float y, float x) { return (float)ptm_atan2d((double)y, (double)x); }

static inline float ptm_asinf(float xf) {
  double x = (double)xf;
  if (x != x) return xf;
  double c = sqrt((1.0 - x) * (1.0 + x)); /* NaN for |x| > 1 */
  return (float)ptm_atan2d(x, c);
}

/* ---- fmod(x, 1.0f): the only fmod call sites are texture.hpp:140,143 ----------- */

static inline float ptm_fmod1f(float x) {
  if (x != x || fabsf(x) == INFINITY) return x - x; /* NaN */
  float r = x - truncf(x);                            /* exact */
  return copysignf(r, x);
}

#endif /* PTM_PORTABLE_H */
