// fp64 device math for the activation hot loop.
//
// ocml's tanh(double) is a double-double evaluation (~135 f64 VALU instructions, it was 2/3 of the
// forward kernel); the round-2 version (kept under -DHPV_TANH_R4) is ~32 instructions with one v_rcp_f64 (one Newton step + a
// residual correction of the quotient):
//   tanh|x| = -t / (2 + t),  t = expm1(-2|x|) = 2^k (e^r - 1) + (2^k - 1),  -2|x| = k ln2 + r, |r| <= ln2/2,
//   e^r - 1 = r + r^2 (1/2! + r/3! + ... + r^11/13!)      (truncation < 4e-18)
// max abs error 2.3e-16, max relative error 3.6e-16 over [-32, 32] incl. |x| -> 0 (checked against
// mpmath on the host prototype and against ocml on the device in tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>

#ifndef HPV_TANH_R4
// Round 5: 25 fp64 operations instead of 32 (the forward phase of the whole-iteration kernels is fp64-VALU-bound: 15 tanh per lane
// and tile, profiles/r05_notes.md section 5).  Same identity, same division; what changed:
//   * k and 2^k without v_rndne / v_cvt / v_ldexp: km = fma(|x|, -2/ln2, 1.5 * 2^52) holds k in its low mantissa bits (one rounding
//     of the exact product), kf = km - 1.5 * 2^52, and 2^k is assembled from the low dword with one integer instruction;
//   * the reduced argument is w = |x| + k ln2/2 = -r/2 (no separate y = -2|x|), |w| <= ln2/4, in ONE fma (see below);
//   * e^(-2w) - 1 = w (w Q(w) - 2) with Q a degree-9 interpolant of (e^(-2w) - 1 + 2w) / w^2 at the Chebyshev nodes of the interval
//     (|w| |Q - exact| / 2 < 3.7e-17; the degree-11 Taylor polynomial of the round-2 version needed two more steps).
// max relative error 3.8e-16 over [-32, 32] incl. |x| -> 0 and the reduction boundaries (host prototype in exact rational
// arithmetic against mpmath, scripts/tanh_proto.py; on the device tests/test_gpu_parity.py).  -DHPV_TANH_R4: the round-2 version.
__device__ __forceinline__ double hpv_tanh(double x) {
    const double ax = fmin(fabs(x), 32.0);       // (fmin drops a NaN argument: it is put back on the result below)
    const double km = fma(ax, -2.8853900817779268, 6755399441055744.0);
    const double kf = km - 6755399441055744.0;   // k = rint(-2|x| / ln2), in [-93, 0]
    const double w = fma(kf, 0.34657359027997264, ax);      // ln2/2 to 53 bits: the product is exact inside the fma, the constant's own
                                                            // rounding (2^-55 |k|) only matters where 2^k has made the exponential small
    double p = -5.1405589494805136e-05;
    p = fma(p, w, 0.0002828297056809958);
    p = fma(p, w, -0.0014109321451518497);
    p = fma(p, w, 0.00634918945176432);
    p = fma(p, w, -0.02539682542470863);
    p = fma(p, w, 0.08888888907016779);
    p = fma(p, w, -0.26666666666656197);
    p = fma(p, w, 0.6666666666659861);
    p = fma(p, w, -1.3333333333333335);
    p = fma(p, w, 2.0000000000000004);
    p = w * fma(w, p, -2.0);                     // e^r - 1, r = -2w
    const double s = __hiloint2double((__double2loint(km) << 20) + 0x3ff00000, 0);     // 2^k
    const double t = fma(s, p, s - 1.0);         // expm1(-2|x|) in (-1, 0]
    const double d = 2.0 + t;                    // in (1, 2]
    double rc = __builtin_amdgcn_rcp(d);         // ~2^-26 relative
    double e = fma(-d, rc, 1.0);
    rc = fma(rc, e, rc);                         // one Newton step: ~2^-52
    double q = -t * rc;
#ifndef HPV_TANH_NOCORR
    const double rem = fma(-d, q, -t);           // exact residual of the quotient
    q = fma(rem, rc, q);                         // correction: the quotient is good to ~1 ulp
#endif
    // sign and NaN on the HIGH dword only (the instructions beside the fp64 pipe count too: the forward phase issues one VALU instruction
    // at a time): sign of x onto q (v_bfi), and the high dword of the canonical quiet NaN when x is a NaN -- any low dword makes that a
    // NaN again, like ocml / tf.tanh (a diverged hidden state must not turn into a finite loss)
    q = copysign(q, x);
#ifdef HPV_TANH_TAIL_A       // (A/B: the integer test + two selects of the first round-5 version)
    return (unsigned)(__double2hiint(x) & 0x7fffffff) > 0x7ff00000u ? x : q;
#else
    const int rh = __builtin_amdgcn_class(x, 0x3) ? 0x7ff80000 : __double2hiint(q);    // v_cmp_class_f64: signalling or quiet NaN
    return __hiloint2double(rh, __double2loint(q));
#endif
}
#else
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

#endif

// (The stage-major evaluation of N independent tanh -- `hpv_tanh_n`, round 3: +0.4 us, the compiler's serial chains were never the
//  problem -- is in the history: profiles/r03_fused_kernel_history.md section 2.)

// sin and cos together for the 1-D drivers' activation (P1:134; the derivative channels need the cosine).  ocml's sincos is
// two argument reductions with a Payne-Hanek branch and ~190 instructions; this one is 4 fma of Cody-Waite reduction against
// pi/2 split into 33 + 33 + 33 + 53 bits (k pi/2 is exact in the first product for |k| < 2^20), the two fdlibm kernel
// polynomials on [-pi/4, pi/4] and a branch-free quadrant fix-up: ~40 instructions.  Max error 2.2 ulp over |x| <= 1e6
// (host prototype against mpmath, 4e4 points incl. the doubles nearest to multiples of pi/2; on the device against ocml in
// tests/test_gpu_parity.py).  |x| > 1e6, NaN and inf take ocml's path (wave-divergent only there).
// hpv_sincos_fast: the branch-free part, valid for |x| <= HPV_SINCOS_MAX (anything, but finite or NaN, beyond); the MFMA
// kernels evaluate a whole layer with it and redo the layer through hpv_sincos when any lane of the wave saw a larger
// argument -- a branch per value would cut the five independent chains of a lane into separate basic blocks.
#define HPV_SINCOS_MAX 1.0e6
#ifndef HPV_SINCOS_R4
// Round 5 (same lens as the tanh: every VALU instruction of the activation is paid per lane, tile and layer): the quadrant number from the
// bits of km = fma(x, 2/pi, 1.5 * 2^52) (no v_rndne / v_cvt), the reduction as fma(k, -C, .) so that x = -0 stays -0 through it (k = +0:
// (+0)(-C) + (-0) = -0) and sin = r (1 + z p) keeps that sign -- no compare-and-select for sin(-0) = -0 --, the cosine as
// 1 - z/2 + z^2 q in three operations instead of fdlibm's six (compensated) ones.  Max error 1.9 ulp (sin) / 1.8 ulp (cos) over the
// same sample as before (scripts/sincos_proto.py: exact-arithmetic prototype against mpmath; round-4 form 1.9 / 1.75 -- the worst cases are
// the reduction's; away from them sin = r (1 + z p) is up to 0.4 ulp behind fma(z r, p, r)).  -DHPV_SINCOS_R4: the round-4 form.
__device__ __forceinline__ void hpv_sincos_fast(double x, double* so, double* co) {
    const double km = fma(x, 6.36619772367581382433e-01, 6755399441055744.0);
    const double k = km - 6755399441055744.0;
    double r = fma(k, -1.57079632673412561417e+00, x);
    r = fma(k, -6.07710050630396597660e-11, r);
    r = fma(k, -2.02226624871116645580e-21, r);
    r = fma(k, -8.47842766036889956997e-32, r);
    const double z = r * r;
    double p = 1.58969099521155010221e-10;
    p = fma(p, z, -2.50507602534068634195e-08);
    p = fma(p, z, 2.75573137070700676789e-06);
    p = fma(p, z, -1.98412698298579493134e-04);
    p = fma(p, z, 8.33333333332248946124e-03);
    p = fma(p, z, -1.66666666666666324348e-01);
    const double s = r * fma(z, p, 1.0);
    double q = -1.13596475577881948265e-11;
    q = fma(q, z, 2.08757232129817482790e-09);
    q = fma(q, z, -2.75573143513906633035e-07);
    q = fma(q, z, 2.48015872894767294178e-05);
    q = fma(q, z, -1.38888888888741095749e-03);
    q = fma(q, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, q, fma(-0.5, z, 1.0));
    const int n = __double2loint(km);            // k mod 2^32 (|x| <= 1e6: |k| < 2^20)
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
    // quadrant signs: sin flips for n = 2, 3 (mod 4), cos for n = 1, 2
    *so = __longlong_as_double(__double_as_longlong(a) ^ ((long long)(n & 2) << 62));
    *co = __longlong_as_double(__double_as_longlong(b) ^ ((long long)((n + 1) & 2) << 62));
}
#else
__device__ __forceinline__ void hpv_sincos_fast(double x, double* so, double* co) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624871116645580e-21, r);
    r = fma(-k, 8.47842766036889956997e-32, r);
    const double z = r * r;
    double p = 1.58969099521155010221e-10;
    p = fma(p, z, -2.50507602534068634195e-08);
    p = fma(p, z, 2.75573137070700676789e-06);
    p = fma(p, z, -1.98412698298579493134e-04);
    p = fma(p, z, 8.33333333332248946124e-03);
    p = fma(p, z, -1.66666666666666324348e-01);
    const double s = fma(z * r, p, r);
    double q = -1.13596475577881948265e-11;
    q = fma(q, z, 2.08757232129817482790e-09);
    q = fma(q, z, -2.75573143513906633035e-07);
    q = fma(q, z, 2.48015872894767294178e-05);
    q = fma(q, z, -1.38888888888741095749e-03);
    q = fma(q, z, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * z * q);
    const int n = (int)k;
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
    // quadrant signs: sin flips for n = 2, 3 (mod 4), cos for n = 1, 2
    const double sv = __longlong_as_double(__double_as_longlong(a) ^ ((long long)(n & 2) << 62));
    *so = x == 0.0 ? x : sv;                     // sin(-0) = -0 (the reduction's fma turns it into +0)
    *co = __longlong_as_double(__double_as_longlong(b) ^ ((long long)((n + 1) & 2) << 62));
}
#endif
__device__ __forceinline__ void hpv_sincos(double x, double* so, double* co) {
    if (__builtin_expect(!(fabs(x) <= HPV_SINCOS_MAX), 0)) sincos(x, so, co);
    else hpv_sincos_fast(x, so, co);
}
