#!/usr/bin/env python3
"""Host prototype of the device sincos (csrc/hpv_math.h): the operation sequence with every fma / product rounded once (exact rational
arithmetic) against mpmath; prints the max error in ulp of sin and cos for the shipped form and for the trimmed one (k from the bits
of one fma, sin(-0) = -0 through the signs of the reduction's products, cos tail 1 - z/2 + z^2 q in three operations).
  python scripts/sincos_proto.py [n_samples]"""
import math, random, sys
from fractions import Fraction as F
import mpmath as mp
mp.mp.dps = 60

def fma(a, b, c): return float(F(a) * F(b) + F(c))
C = (1.57079632673412561417e+00, 6.07710050630396597660e-11, 2.02226624871116645580e-21, 8.47842766036889956997e-32)
SP = (1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01)
CQ = (-1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02)
MAGIC = 6755399441055744.0

def kernels(r, trimmed):
    z = r * r
    p = SP[0]
    for c in SP[1:]: p = fma(p, z, c)
    q = CQ[0]
    for c in CQ[1:]: q = fma(q, z, c)
    if trimmed:
        s = r * fma(z, p, 1.0)
        c = fma(z * z, q, fma(-0.5, z, 1.0))
    else:
        s = fma(z * r, p, r)
        hz = 0.5 * z; w = 1.0 - hz
        c = w + (((1.0 - w) - hz) + z * z * q)
    return s, c

def sincos(x, trimmed):
    if trimmed:
        km = fma(x, 6.36619772367581382433e-01, MAGIC); k = km - MAGIC
        r = fma(k, -C[0], x)
        for c in C[1:]: r = fma(k, -c, r)
    else:
        k = float(round(x * 6.36619772367581382433e-01))
        r = fma(-k, C[0], x)
        for c in C[1:]: r = fma(-k, c, r)
    s, c = kernels(r, trimmed)
    n = int(k)
    a, b = (c, s) if n & 1 else (s, c)
    if n & 2: a = -a
    if (n + 1) & 2: b = -b
    if not trimmed and x == 0.0: a = x
    return a, b

def ulp(v, ref):
    ref_f = float(ref)
    u = math.ulp(abs(ref_f)) if ref_f != 0 else 5e-324
    return float(abs(mp.mpf(v) - ref) / u)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    random.seed(3)
    xs = [random.uniform(-40, 40) for _ in range(n)] + [random.uniform(-1e6, 1e6) for _ in range(n // 2)] + [k * (math.pi / 2) for k in range(0, 1500)] \
        + [10.0 ** random.uniform(-300, 0) * random.choice((-1, 1)) for _ in range(n // 4)] + [1e6, -1e6]
    for trimmed in (False, True):
        ws = wc = 0.0
        for x in xs:
            a, b = sincos(x, trimmed)
            ws = max(ws, ulp(a, mp.sin(mp.mpf(x)))); wc = max(wc, ulp(b, mp.cos(mp.mpf(x))))
        print("trimmed" if trimmed else "shipped", "max ulp error: sin %.2f cos %.2f" % (ws, wc))
    for trimmed in (False, True):
        a, b = sincos(-0.0, trimmed)
        print("sin(-0) =", a, math.copysign(1, a), "cos(-0) =", b)
