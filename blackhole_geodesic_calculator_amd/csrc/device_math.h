// Device math shared by the trace kernels and the frame kernels (gfx950): the reciprocal and the atan2 both use.
#pragma once
#include <hip/hip_runtime.h>

namespace bhg {

// 1/x: v_rcp_f64 + one cubic Newton step (about an ulp), the trace kernels' rcp_nr is this function
__device__ __forceinline__ double rcp_newton(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    double t = __builtin_fma(e, e, e);
    return __builtin_fma(y, t, y);
}

// atan2(y, x) for finite arguments, about an ulp, without libm's special-case ladder (the sky lookup calls it twice
// per ray and was most of the shade kernel's instructions): octant reduction to q = min/max in [0, 1], then the
// classic argument reduction at 7/16 and 11/16 -- t = q, (2q - 1)/(2 + q) or (q - 1)/(q + 1) -- and the odd
// degree-21 minimax polynomial on |t| < 7/16.  atan2(0, 0) = 0.
//
// Attribution: breakpoints, atan(1/2) / atan(1) hi + lo parts and the coefficients aT[0..10] are those of FreeBSD msun
// / fdlibm's s_atan.c: "Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a Sun
// Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this software is freely granted,
// provided that this notice is preserved."
__device__ __forceinline__ double atan2_fast(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double q = (mx > 0.0) ? mn * rcp_newton(mx) : 0.0;
    const bool r0 = q < 0.4375, r1 = q < 0.6875;
    const double num = r0 ? q : (r1 ? __builtin_fma(2.0, q, -1.0) : q - 1.0);
    const double den = r0 ? 1.0 : (r1 ? 2.0 + q : q + 1.0);
    const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : 7.85398163397448278999e-01);
    const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : 3.06161699786838301793e-17);
    const double t = num * rcp_newton(den);
    const double z = t * t, w = z * z;
    double s1 = __builtin_fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    s1 = __builtin_fma(w, s1, 6.66107313738753120669e-02);
    s1 = __builtin_fma(w, s1, 9.09088713343650656196e-02);
    s1 = __builtin_fma(w, s1, 1.42857142725034663711e-01);
    s1 = __builtin_fma(w, s1, 3.33333333333329318027e-01);
    s1 *= z;
    double s2 = __builtin_fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s2 = __builtin_fma(w, s2, -7.69187620504482999495e-02);
    s2 = __builtin_fma(w, s2, -1.11111104054623557880e-01);
    s2 = __builtin_fma(w, s2, -1.99999999998764832476e-01);
    s2 *= w;
    double r = hi - ((t * (s1 + s2) - lo) - t);      // atan(q) in [0, pi/4]
    r = (ay > ax) ? 1.5707963267948966 - r : r;      // first quadrant
    r = (x < 0.0) ? 3.141592653589793 - r : r;
    return (y < 0.0) ? -r : r;
}

}  // namespace bhg
