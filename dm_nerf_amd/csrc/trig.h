// Accurate sin/cos for positional encodings on gfx950 (shared by the fused field kernel and the
// standalone encoder).  The encoder multiplies coordinates by exact powers of two
// (networks/dm_nerf.py:25,31), so arguments reach ~7.7e3 rad; the hardware v_sin_f32 path is not
// accurate there and a 1e-4 end-to-end tolerance needs ~1 ulp encodings.
#pragma once
#include <hip/hip_runtime.h>

namespace dmn {

// h ? cos(a) : sin(a) for the exact fp32 argument a (|a| up to ~1e5), correctly rounded in
// practice: Cody-Waite reduction and Taylor kernels in fp64, one rounding to fp32 at the end.
__device__ __forceinline__ float trig_sel(float a, int h) {
  const double ad = (double)a;
  const double kd = __builtin_rint(ad * 0.63661977236758134308);
  double r = __builtin_fma(-kd, 1.57079632679489655800e+00, ad);
  r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
  const int q = (int)kd + h;
  const double r2 = r * r;
  double s = 1.6059043836821613e-10;                       // 1/13!
  s = __builtin_fma(s, r2, -2.5052108385441720e-08);       // -1/11!
  s = __builtin_fma(s, r2, 2.7557319223985893e-06);        // 1/9!
  s = __builtin_fma(s, r2, -1.9841269841269841e-04);       // -1/7!
  s = __builtin_fma(s, r2, 8.3333333333333332e-03);        // 1/5!
  s = __builtin_fma(s, r2, -1.6666666666666666e-01);       // -1/3!
  s = __builtin_fma(s * r2, r, r);
  double c = -1.1470745597729725e-11;                      // -1/14!
  c = __builtin_fma(c, r2, 2.0876756987868100e-09);        // 1/12!
  c = __builtin_fma(c, r2, -2.7557319223985888e-07);       // -1/10!
  c = __builtin_fma(c, r2, 2.4801587301587302e-05);        // 1/8!
  c = __builtin_fma(c, r2, -1.3888888888888889e-03);       // -1/6!
  c = __builtin_fma(c, r2, 4.1666666666666664e-02);        // 1/4!
  c = __builtin_fma(c, r2, -0.5);
  c = __builtin_fma(c, r2, 1.0);
  double v = (q & 1) ? c : s;
  v = (q & 2) ? -v : v;
  return (float)v;
}

// sin and cos of the exact fp32 argument a in fp64 (the same reduction and kernels as trig_sel, both results kept)
__device__ __forceinline__ void sincos_f64(float a, double& sn, double& cs) {
  const double ad = (double)a;
  const double kd = __builtin_rint(ad * 0.63661977236758134308);
  double r = __builtin_fma(-kd, 1.57079632679489655800e+00, ad);
  r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
  const int q = (int)kd;
  const double r2 = r * r;
  double s = 1.6059043836821613e-10;
  s = __builtin_fma(s, r2, -2.5052108385441720e-08);
  s = __builtin_fma(s, r2, 2.7557319223985893e-06);
  s = __builtin_fma(s, r2, -1.9841269841269841e-04);
  s = __builtin_fma(s, r2, 8.3333333333333332e-03);
  s = __builtin_fma(s, r2, -1.6666666666666666e-01);
  s = __builtin_fma(s * r2, r, r);
  double c = -1.1470745597729725e-11;
  c = __builtin_fma(c, r2, 2.0876756987868100e-09);
  c = __builtin_fma(c, r2, -2.7557319223985888e-07);
  c = __builtin_fma(c, r2, 2.4801587301587302e-05);
  c = __builtin_fma(c, r2, -1.3888888888888889e-03);
  c = __builtin_fma(c, r2, 4.1666666666666664e-02);
  c = __builtin_fma(c, r2, -0.5);
  c = __builtin_fma(c, r2, 1.0);
  const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? -s : c;      // rotate by q quarter turns
  sn = (q & 2) ? -s1 : s1;
  cs = (q & 2) ? -c1 : c1;
}

// The encoder's bands are exact powers of two (networks/dm_nerf.py:25,31), so band f + 1 is the double angle of band f:
//   sin 2a = 2 sin a cos a,   cos 2a = 1 - 2 sin^2 a
// in fp64 from one sincos_f64 of the coordinate.  The error at most quadruples per band: 1.5e-11 absolute at 2^9
// (measured against fp64 libm on 200 000 coordinates up to +-15), four orders below an fp32 ulp; the fp32 roundings
// differ from the correctly rounded value in 1e-5 of the cases, by one ulp.  Six times fewer fp64 operations than a
// reduction + two polynomials per band and function (the prologue of every field kernel: ~4 us per 128 points before).
__device__ __forceinline__ void double_angle(double& sn, double& cs) {
  const double t = sn * cs;
  cs = __builtin_fma(-2.0 * sn, sn, 1.0);
  sn = t + t;
}

}  // namespace dmn
