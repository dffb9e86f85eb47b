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

#ifndef ALILQR_HORNER_PI
#define ALILQR_HORNER_PI 1  // Horner form of the wrapped-angle kernels
#endif
#ifndef ALILQR_HORNER_W
#define ALILQR_HORNER_W 0   // Horner form of the unwrapped-angle kernels (0 compiler, 1 three-address, 2 scalar coefficient)
#endif

namespace alilqr {

// The device library's large-argument paths (Payne-Hanek reduction, the fmod loop) are reached only by diverging
// solves.  They are calls, not inlined code: inlined at every use they multiplied the code of the roll-out loops and
// their registers were allocated on the paths that never run them.
struct SinCos { double s, c; };
__device__ __noinline__ SinCos sincos_far(double x) {
  SinCos r;
  sincos(x, &r.s, &r.c);
  return r;
}
__device__ __noinline__ double tan_far(double x) { return tan(x); }
__device__ __noinline__ double fmod_far(double x, double y) { return fmod(x, y); }

// One Horner step  a * b + c  with a coefficient c that lives on.  Written as the three-address instruction: left to
// the compiler a step becomes a copy of the coefficient plus the two-address v_fmac_f64 that overwrites the copy -
// one move per coefficient, 45 per roll-out step of the lateral model.  Same operation, same bits.
// Measured (ALILQR_HORNER_PI / ALILQR_HORNER_W pick the form of the wrapped- / unwrapped-angle kernels: 0 the compiler's,
// 1 three-address with the coefficient in a vector register, 2 with the coefficient in a scalar register pair):
// form 1 is +4 % on the headline kernel, whose roll-outs use the wrapped-angle kernels; in the unwrapped-angle kernels
// (C++ generation, Jacobians) forms 1 and 2 change nothing or cost registers (4-state NEW_AL at 32 lanes -20 %), and
// form 2 in the headline kernel trades the moves for twice as many scalar moves (-2 %).
template <int THREE>
__device__ __forceinline__ double horner(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (THREE == 1) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  }
  if constexpr (THREE == 2) {                         // the coefficient from a scalar register pair
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
  }
#endif
  return fma(a, b, c);
}

template <int THREE = ALILQR_HORNER_W>
__device__ __forceinline__ void sincos_kernels(double r, double* s, double* c) {
  const double z = r * r;
  // sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = horner<THREE == 2 ? 0 : THREE>(z, S6, S5);
  ps = horner<THREE>(z, ps, S4);
  ps = horner<THREE>(z, ps, S3);
  ps = horner<THREE>(z, ps, S2);
  const double v = z * r;
  *s = fma(v, horner<THREE>(z, ps, S1), r);
  // cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = horner<THREE == 2 ? 0 : THREE>(z, C6, C5);
  pc = horner<THREE>(z, pc, C4);
  pc = horner<THREE>(z, pc, C3);
  pc = horner<THREE>(z, pc, C2);
  pc = horner<THREE>(z, pc, C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  *c = w + (((1.0 - w) - hz) + z * (z * pc));
}

__device__ __forceinline__ void sincos_w(double x, double* s, double* c) {
  if (__builtin_expect(!(fabs(x) <= 4.0), 0)) {
    const SinCos r = sincos_far(x);
    *s = r.s; *c = r.c;
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

// |x| <= 4 assumed (wrapped angles); NaN in, NaN out.  No large-argument path.
__device__ __forceinline__ void sincos_pi(double x, double* s, double* c) {
  const double q = rint(x * 6.36619772367581382433e-01);
  double r = fma(-q, 1.57079632679489655800e+00, x);
  r = fma(-q, 6.12323399573676603587e-17, r);
  double sr, cr;
  sincos_kernels<ALILQR_HORNER_PI>(r, &sr, &cr);
  const int n = (int)q;
  const double ss = (n & 1) ? cr : sr;
  const double cc = (n & 1) ? sr : cr;
  *s = (n & 2) ? -ss : ss;
  *c = ((n + 1) & 2) ? -cc : cc;
}

// tan on |x| < 0.67434 by the odd minimax polynomial of the fdlibm tangent kernel (< 1 ulp)
template <int THREE = ALILQR_HORNER_W>
__device__ __forceinline__ double tan_poly(double x) {
  const double T0 = 3.33333333333334091986e-01, T1 = 1.33333333333201242699e-01, T2 = 5.39682539762260521377e-02,
               T3 = 2.18694882948595424599e-02, T4 = 8.86323982359930005737e-03, T5 = 3.59207910759131235356e-03,
               T6 = 1.45620945432529025516e-03, T7 = 5.88041240820264096874e-04, T8 = 2.46463134818469906812e-04,
               T9 = 7.81794442939557092300e-05, T10 = 7.14072491382608190305e-05, T11 = -1.85586374855275456654e-05,
               T12 = 2.59073051863633712884e-05;
  const double z = x * x;
  const double w = z * z;
  double r = horner<THREE == 2 ? 0 : THREE>(w, T11, T9);
  r = horner<THREE>(w, r, T7);
  r = horner<THREE>(w, r, T5);
  r = horner<THREE>(w, r, T3);
  r = horner<THREE>(w, r, T1);
  double v = horner<THREE == 2 ? 0 : THREE>(w, T12, T10);
  v = horner<THREE>(w, v, T8);
  v = horner<THREE>(w, v, T6);
  v = horner<THREE>(w, v, T4);
  v = horner<THREE>(w, v, T2);
  v = z * v;
  const double s3 = z * x;
  r = z * (s3 * (r + v));
  r = fma(T0, s3, r);
  return x + r;
}

// tan for wrapped angles (|x| <= 4 or NaN): polynomial on small angles, sin/cos beyond
__device__ __forceinline__ double tan_pi(double x) {
  if (__builtin_expect(fabs(x) < 0.67434, 1)) return tan_poly<ALILQR_HORNER_PI>(x);
  double s, c;
  sincos_pi(x, &s, &c);
  return s / c;
}

__device__ __forceinline__ double tan_w(double x) {
  if (__builtin_expect(fabs(x) < 0.67434, 1)) return tan_poly(x);
  if (__builtin_expect(!(fabs(x) <= 4.0), 0)) return tan_far(x);
  double s, c;
  sincos_w(x, &s, &c);
  return s / c;
}

}  // namespace alilqr
