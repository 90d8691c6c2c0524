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

// The same tanh for N independent arguments, written STAGE-MAJOR: every step of the algorithm is applied to all N values before
// the next step.  hpv_tanh is one dependent chain of ~28 fp64 operations; with one wave per SIMD (k_iter_fused) nothing else
// issues while a dependent v_fma_f64 waits for its predecessor, and the compiler keeps inlined copies of the scalar routine
// one after the other.  Interleaved, the N chains cover each other's latency (same arithmetic per value: bit-identical results).
// Left alone, instruction selection and the machine scheduler re-cluster the chains (they minimise register pressure), and a
// scheduling fence (sched_barrier) only binds the machine scheduler, which then finds the chains already clustered.  An empty
// `asm volatile` that takes the N values of a stage as read-write operands pins the stage-major order at every level: volatile
// asm statements keep their program order, stage k feeds pin k, pin k feeds stage k + 1.  It emits no instruction, and MFMA /
// LDS / memory instructions that do not touch the pinned values still move freely.
template <int N>
__device__ __forceinline__ void hpv_pin(double (&a)[N]) {
    static_assert(N == 5 || N == 10, "pin lists are written out for 5 and 10 values");
    if constexpr (N == 10)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]));
    else
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
}
template <int N>
__device__ __forceinline__ void hpv_tanh_n(const double (&x)[N], double (&out)[N]) {
    double y[N], k[N], r[N], p[N], s[N], t[N], d[N], rc[N], q[N], e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = fmin(fabs(x[i]), 32.0);
    hpv_pin(y);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = -2.0 * y[i];
    hpv_pin(y);
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = y[i] * 1.4426950408889634;
    hpv_pin(k);
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = rint(k[i]);
    hpv_pin(k);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(-k[i], 6.93147180369123816490e-01, y[i]);
    hpv_pin(r);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(-k[i], 1.90821492927058770002e-10, r[i]);
    hpv_pin(r);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma(1.6059043836821613e-10, r[i], 2.08767569878681e-09);
    hpv_pin(p);
#define HPV_TANH_STEP(C)                  \
    _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = fma(p[i], r[i], C); \
    hpv_pin(p);
    HPV_TANH_STEP(2.505210838544172e-08)
    HPV_TANH_STEP(2.755731922398589e-07)
    HPV_TANH_STEP(2.7557319223985893e-06)
    HPV_TANH_STEP(2.48015873015873e-05)
    HPV_TANH_STEP(0.0001984126984126984)
    HPV_TANH_STEP(0.001388888888888889)
    HPV_TANH_STEP(0.008333333333333333)
    HPV_TANH_STEP(0.041666666666666664)
    HPV_TANH_STEP(0.16666666666666666)
    HPV_TANH_STEP(0.5)
#undef HPV_TANH_STEP
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = r[i] * r[i];
    hpv_pin(e);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma(e[i], p[i], r[i]);
    hpv_pin(p);
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = __builtin_amdgcn_ldexp(1.0, (int)k[i]);
    hpv_pin(s);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = s[i] - 1.0;
    hpv_pin(e);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = fma(s[i], p[i], e[i]);
    hpv_pin(t);
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = 2.0 + t[i];
    hpv_pin(d);
#pragma unroll
    for (int i = 0; i < N; ++i) rc[i] = __builtin_amdgcn_rcp(d[i]);
    hpv_pin(rc);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = fma(-d[i], rc[i], 1.0);
    hpv_pin(e);
#pragma unroll
    for (int i = 0; i < N; ++i) rc[i] = fma(rc[i], e[i], rc[i]);
    hpv_pin(rc);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = -t[i] * rc[i];
    hpv_pin(q);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = fma(-d[i], q[i], -t[i]);
    hpv_pin(e);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = fma(e[i], rc[i], q[i]);
    hpv_pin(q);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = x[i] != x[i] ? x[i] : copysign(q[i], x[i]);
}

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
__device__ __forceinline__ void hpv_sincos(double x, double* so, double* co) {
    if (__builtin_expect(!(fabs(x) <= HPV_SINCOS_MAX), 0)) sincos(x, so, co);
    else hpv_sincos_fast(x, so, co);
}
