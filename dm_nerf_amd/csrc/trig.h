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


}  // namespace dmn
