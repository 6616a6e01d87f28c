/* Exhaustive proof obligation of the checker texture's sign-only form (path_tracer_amd/csrc/pt_math.hpp: sin_negative_regular,
 * pt_device.hpp: checker_sines_negative).  Test infrastructure: it includes the ORACLE's math (oracle/ptm_portable.h) as the definition.
 *
 * texture.hpp:43-45 decides on `sin(10 x) sin(10 y) sin(10 z) < 0`.  For a REGULAR argument, 2^-30 <= |a| < 2^30, the device
 * takes the sign of sinf_(a) from the range reduction alone (n = quadrant, r = reduced argument: negative iff (n even and r < 0)
 * xor (n & 2)) and never evaluates the polynomials.  This program checks, for EVERY regular binary32 argument of both signs:
 *   (1) the sign-only form equals the sign bit of ptm_sinf(a), and ptm_sinf(a) is neither zero nor NaN;
 *   (1b) wherever the binary32 first stage decides (u = |a| fl(1/pi) at least u 2^-21 from an integer), its sign equals that sign bit too;
 *   (2) |ptm_sinf(a)| >= 2^-40, so that a product of three such factors (>= 2^-120) never underflows in binary32 and its sign is
 *       the product of the signs.
 * Prints the smallest |sin| seen and "ok", or the first counter-example.   gcc -O2 -fopenmp checker_sign_exhaustive.c -lm
 * argv[1] (optional): stride over the significands (1 = every float; the CPU suite runs a stride that still hits every binade). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/ptm_portable.h"

static int sign_only(float a) { /* the device's form, restated */
  double r;
  const int n = ptm_rem_pio2((double)a, &r);
  const int neg_v = (n & 1) ? 0 : (r < 0.0);
  return neg_v != ((n & 2) != 0);
}

/* the binary32 first stage (pt_device.hpp: checker_sines_negative): 1 = decided, *neg = its answer */
static int sign_f32(float a, int* neg) {
  const float u = fabsf(a) * 0.318309886183790671538f;
  const float fl = floorf(u), fr = u - fl, d = u * 4.76837158203125e-07f;
  *neg = ((((int)fl) & 1) != 0) != (a < 0.0f);
  return (fr >= d) && (fr <= 1.0f - d);
}

int main(int argc, char** argv) {
  const unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  const unsigned lo = (127u - 30u) << 23, hi = (127u + 30u) << 23; /* [2^-30, 2^30) */
  double min_abs = 1.0;
  unsigned bad = 0, first_bad = 0;
  unsigned long long checked = 0, decided32 = 0;
#pragma omp parallel for schedule(static) reduction(min : min_abs) reduction(+ : bad, checked, decided32)
  for (unsigned e = lo >> 23; e < (hi >> 23); e++) {
    for (unsigned m = 0; m < (1u << 23); m += stride) {
      /* the stride walks the significands; the last 64 and first 64 of every binade are always visited */
      for (int pass = 0; pass < (stride > 1 ? 3 : 1); pass++) {
        unsigned mm = m;
        if (pass == 1) { if (m >= 64u * stride) break; mm = m / stride; }
        if (pass == 2) { if (m >= 64u * stride) break; mm = (1u << 23) - 1u - m / stride; }
        for (unsigned s = 0; s < 2; s++) {
          const unsigned bits = (s << 31) | (e << 23) | mm;
          float a;
          memcpy(&a, &bits, 4);
          const float f = ptm_sinf(a);
          unsigned fb;
          memcpy(&fb, &f, 4);
          int neg32 = 0;
          const int dec32 = sign_f32(a, &neg32);
          if (dec32) decided32++;
          const int ok = (f == f) && f != 0.0f && ((int)(fb >> 31) == sign_only(a)) && fabs((double)f) >= 0x1p-40 &&
                         (!dec32 || neg32 == (int)(fb >> 31));
          if (!ok) {
            bad++;
#pragma omp critical
            if (!first_bad) first_bad = bits;
          }
          if (fabs((double)f) < min_abs) min_abs = fabs((double)f);
          checked++;
        }
      }
    }
  }
  printf("checked %llu arguments, smallest |sinf_| = %.6e (2^%.2f)\n", checked, min_abs, log2(min_abs));
  printf("the binary32 first stage decided %llu of them\n", decided32);
  if (bad) { printf("FAILED: %u counter-examples, first bits 0x%08x\n", bad, first_bad); return 1; }
  printf("ok\n");
  return 0;
}
