// trig.hpp — fp64 sin/cos/tan for the angle ranges of the bicycle models.
//
// Heading and steering angles are wrapped to [-pi, pi) before they reach the
// trigonometric functions, so the argument reduction needs no large-argument
// (Payne-Hanek) path: one rounding to the nearest multiple of pi/2 with a
// two-part pi/2 through fma, then the classical minimax kernels on
// |r| <= pi/4 (coefficients of the fdlibm kernels, < 1 ulp).  Arguments
// outside |x| <= 4 (only met on diverging solves) take the device library.
#pragma once
#include <hip/hip_runtime.h>

namespace alilqr {

__device__ __forceinline__ void sincos_kernels(double r, double* s, double* c) {
  const double z = r * r;
  // sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = fma(z, S6, S5);
  ps = fma(z, ps, S4);
  ps = fma(z, ps, S3);
  ps = fma(z, ps, S2);
  const double v = z * r;
  *s = fma(v, fma(z, ps, S1), r);
  // cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = fma(z, C6, C5);
  pc = fma(z, pc, C4);
  pc = fma(z, pc, C3);
  pc = fma(z, pc, C2);
  pc = fma(z, pc, C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  *c = w + (((1.0 - w) - hz) + z * (z * pc));
}

__device__ __forceinline__ void sincos_w(double x, double* s, double* c) {
  if (__builtin_expect(!(fabs(x) <= 4.0), 0)) {
    sincos(x, s, c);
    return;
  }
  const double q = rint(x * 6.36619772367581382433e-01);          // nearest multiple of pi/2
  double r = fma(-q, 1.57079632679489655800e+00, x);               // pi/2 high part
  r = fma(-q, 6.12323399573676603587e-17, r);                      // pi/2 low part
  double sr, cr;
  sincos_kernels(r, &sr, &cr);
  const int n = (int)q;
  const double ss = (n & 1) ? cr : sr;
  const double cc = (n & 1) ? sr : cr;
  *s = (n & 2) ? -ss : ss;
  *c = ((n + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ double tan_w(double x) {
  if (__builtin_expect(!(fabs(x) <= 4.0), 0)) return tan(x);
  double s, c;
  sincos_w(x, &s, &c);
  return s / c;
}

}  // namespace alilqr
