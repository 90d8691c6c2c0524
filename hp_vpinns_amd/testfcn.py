"""Legendre-difference test functions and their derivatives (host side tables).

phi_n(x)   = P_{n+1}(x) - P_{n-1}(x),  n = 1..N     (vanish at +-1)
phi'_n(x)  = (n+2)/2 P_n^{(1,1)} - n/2 P_{n-2}^{(1,1)}          (2nd term absent for n=1)
phi''_n(x) = (n+2)(n+3)/4 P_{n-1}^{(2,2)} - n(n+1)/4 P_{n-3}^{(2,2)}  (2nd term absent for n<=2)

Same names / shapes as the reference's `VPINN.Test_fcn` / `VPINN.dTest_fcn`
(P1:157-183, P2:196-229, P3:257-284): the result keeps the shape of `x` behind a leading
N_test axis, i.e. `(N_test, len(x), 1)` for the column vectors the drivers pass.
The tables are evaluated once on the host, uploaded once and LDS-staged by the
projection kernels (they are constants of the iteration, P1:73-74).
"""
import numpy as np

from .quadrature import Jacobi


def Test_fcn(N_test, x):
    return np.asarray([Jacobi(n + 1, 0, 0, x) - Jacobi(n - 1, 0, 0, x)
                       for n in range(1, N_test + 1)])


def dTest_fcn(N_test, x):
    d1, d2 = [], []
    for n in range(1, N_test + 1):
        t1 = ((n + 2) / 2) * Jacobi(n, 1, 1, x)
        if n >= 2:
            t1 = t1 - (n / 2) * Jacobi(n - 2, 1, 1, x)
        t2 = ((n + 2) * (n + 3) / 4) * Jacobi(n - 1, 2, 2, x)
        if n >= 3:
            t2 = t2 - (n * (n + 1) / 4) * Jacobi(n - 3, 2, 2, x)
        d1.append(t1)
        d2.append(t2)
    return np.asarray(d1), np.asarray(d2)


def tables_1d(N_test, xi):
    """(3, N_test, Q) array: phi, phi', phi'' at the 1-D reference nodes `xi` (Q,)."""
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)
    t0 = Test_fcn(N_test, xi)
    t1, t2 = dTest_fcn(N_test, xi)
    return np.ascontiguousarray(np.stack([t0, t1, t2]))
