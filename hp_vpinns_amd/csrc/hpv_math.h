// fp64 device math for the activation hot loop.
//
// ocml's tanh(double) is a double-double evaluation (~135 f64 VALU instructions, it was 2/3 of the
// forward kernel); this one is ~32 instructions with one v_rcp_f64 (one Newton step + a residual correction of the quotient):
//   tanh|x| = -t / (2 + t),  t = expm1(-2|x|) = 2^k (e^r - 1) + (2^k - 1),  -2|x| = k ln2 + r, |r| <= ln2/2,
//   e^r - 1 = r + r^2 (1/2! + r/3! + ... + r^11/13!)      (truncation < 4e-18)
// max abs error 2.3e-16, max relative error 3.6e-16 over [-32, 32] incl. |x| -> 0 (checked against
// mpmath on the host prototype and against ocml on the device in tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double hpv_tanh(double x) {
    const double ax = fmin(fabs(x), 32.0);       // (fmin drops a NaN argument: it is put back on the result below)
    const double y = -2.0 * ax;
    const double k = rint(y * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, y);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;           // 1/13!
    p = fma(p, r, 2.08767569878681e-09);         // 1/12!
    p = fma(p, r, 2.505210838544172e-08);        // 1/11!
    p = fma(p, r, 2.755731922398589e-07);        // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);       // 1/9!
    p = fma(p, r, 2.48015873015873e-05);         // 1/8!
    p = fma(p, r, 0.0001984126984126984);        // 1/7!
    p = fma(p, r, 0.001388888888888889);         // 1/6!
    p = fma(p, r, 0.008333333333333333);         // 1/5!
    p = fma(p, r, 0.041666666666666664);         // 1/4!
    p = fma(p, r, 0.16666666666666666);          // 1/3!
    p = fma(p, r, 0.5);                          // 1/2!
    p = fma(r * r, p, r);                        // e^r - 1
    const double s = __builtin_amdgcn_ldexp(1.0, (int)k);
    const double t = fma(s, p, s - 1.0);         // expm1(-2|x|) in (-1, 0]
    const double d = 2.0 + t;                    // in (1, 2]
    double rc = __builtin_amdgcn_rcp(d);         // ~2^-26 relative
    double e = fma(-d, rc, 1.0);
    rc = fma(rc, e, rc);                         // one Newton step: ~2^-52
    double q = -t * rc;
    const double rem = fma(-d, q, -t);           // exact residual of the quotient
    q = fma(rem, rc, q);                         // correction: the quotient is good to ~1 ulp
    q = copysign(q, x);
    return x != x ? x : q;                       // NaN in -> NaN out, like ocml / tf.tanh (one compare + select)
}
