// sr_pivot_dev.h -- scalar helpers shared by sr_factor.hip and sr_append.hip: the pivot chains (diagonal-block kernel of the
// Cholesky, one-point row append) and the radial part of the general kernel family.
#pragma once
#include "sr_common.h"

__device__ __forceinline__ double sr_readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sd = sqrt(d), inv = 1 / sqrt(d) for a positive, normal d: v_rsq_f64 seed, one coupled Goldschmidt step
// (g -> sqrt, h -> 1/(2 sqrt), both to ~2^-52) and one residual correction each.  The library sqrt() + 1.0 / x
// pair costs ~32 dependent instructions (range scaling, v_div_scale / v_div_fmas / v_div_fixup); this chain is
// 9, and it sits 128 times on the critical path of every diagonal block.  |sd^2 - d| <= 1 ulp(d), |inv sd - 1| <= 2^-52.
__device__ __forceinline__ void sr_sqrt_rsqrt(double d, double& sd, double& inv) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, d);
    g = fma(e, h, g);
    const double r2 = fma(-h, g, 0.5);
    h = fma(h, r2, h);
    sd = g;
    inv = h + h;
}


// radial part kappa(r) of the general kernel family (sr_common.h): 0 RBF, else Matern-5/2
__device__ __forceinline__ double sr_kappa(int kind, double r2) {
    if (kind == 0) return exp(-0.5 * r2);
    const double r = sqrt(r2);
    return (1.0 + 2.23606797749978969641 * r + (5.0 / 3.0) * r2) * exp(-2.23606797749978969641 * r);
}
