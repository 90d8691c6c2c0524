"""Jacobi polynomials and Gauss-Lobatto-Jacobi quadrature rules (host side, numpy only).

Mirrors the public surface of the reference quadrature module
(`Utilities/GaussJacobiQuadRule_V3.py:24-61`): `Jacobi`, `DJacobi`,
`GaussJacobiWeights`, `GaussLobattoJacobiWeights` -- same names, argument order and
return conventions -- but everything is evaluated with the stable three-term
recurrence (never expanded monomial coefficients) and the nodes come from a Newton
iteration on the recurrence instead of scipy's eigenvalue solver, so this file has no
scipy dependency.  Pinned against the reference by `tests/golden/quadrature.npz`.
"""
import math

import numpy as np


def Jacobi(n, a, b, x):
    """P_n^{(a,b)}(x) by the three-term recurrence.  (reference: Q:24-26)

    `x` may be any array-like; the result has the shape of `np.array(x)`.
    n < 0 returns zeros (the reference never asks for it; the test-function code
    guards those terms the same way the reference does, P1:164-183).
    """
    x = np.array(x, dtype=np.float64)
    if n < 0:
        return np.zeros_like(x)
    p0 = np.ones_like(x)
    if n == 0:
        return p0
    p1 = 0.5 * ((a - b) + (a + b + 2.0) * x)
    if n == 1:
        return p1
    for k in range(1, n):
        # 2(k+1)(k+a+b+1)(2k+a+b) P_{k+1} =
        #   (2k+a+b+1)[(2k+a+b+2)(2k+a+b) x + a^2-b^2] P_k - 2(k+a)(k+b)(2k+a+b+2) P_{k-1}
        c = 2.0 * k + a + b
        a1 = 2.0 * (k + 1.0) * (k + a + b + 1.0) * c
        a2 = (c + 1.0) * (a * a - b * b)
        a3 = c * (c + 1.0) * (c + 2.0)
        a4 = 2.0 * (k + a) * (k + b) * (c + 2.0)
        p2 = ((a2 + a3 * x) * p1 - a4 * p0) / a1
        p0, p1 = p1, p2
    return p1


def DJacobi(n, a, b, x, k: int):
    """k-th derivative of P_n^{(a,b)}.  (reference: Q:30-33)"""
    x = np.array(x, dtype=np.float64)
    if k > n:
        return np.zeros_like(x)
    ctemp = math.gamma(a + b + n + 1 + k) / (2 ** k) / math.gamma(a + b + n + 1)
    return ctemp * Jacobi(n - k, a + k, b + k, x)


def _jacobi_roots(n, a, b):
    """Roots of P_n^{(a,b)} by Newton with polynomial deflation-free updates.

    Uses the classical simultaneous (Aberth-like) correction so that every root
    converges to a distinct zero; initial guesses are Chebyshev points.
    """
    if n <= 0:
        return np.zeros(0)
    k = np.arange(n, dtype=np.float64)
    x = -np.cos((2.0 * k + 1.0) * math.pi / (2.0 * n))
    for _ in range(100):
        p = Jacobi(n, a, b, x)
        dp = DJacobi(n, a, b, x, 1)
        diff = x[:, None] - x[None, :]
        np.fill_diagonal(diff, 1.0)
        s = (1.0 / diff).sum(axis=1) - 1.0  # remove the diagonal's 1/1
        dx = p / (dp - s * p)
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    # two plain Newton polish steps (quadratic, removes the Aberth coupling error)
    for _ in range(2):
        x = x - Jacobi(n, a, b, x) / DJacobi(n, a, b, x, 1)
    x = np.sort(x)
    if a == b:  # symmetric weight: enforce exact symmetry like an eigen-solver would not
        x = 0.5 * (x - x[::-1])
    return x


def GaussJacobiWeights(Q: int, a, b):
    """Gauss-Jacobi nodes and weights.  (reference: Q:38-40; imported but never called)"""
    X = _jacobi_roots(Q, a, b)
    # w_i = Gamma(a+Q+1)Gamma(b+Q+1)/(Gamma(a+b+Q+1) Q!) * 2^{a+b+1} / ((1-x^2) P'_Q(x)^2)
    lg = (math.lgamma(a + Q + 1) + math.lgamma(b + Q + 1)
          - math.lgamma(a + b + Q + 1) - math.lgamma(Q + 1))
    c = math.exp(lg) * 2.0 ** (a + b + 1)
    W = c / ((1.0 - X * X) * DJacobi(Q, a, b, X, 1) ** 2)
    return [X, W]


def GaussLobattoJacobiWeights(Q: int, a, b):
    """Gauss-Lobatto-Jacobi nodes (endpoints included) and weights.  (reference: Q:46-61)

    Interior nodes are the roots of P_{Q-2}^{(a+1,b+1)}; the Legendre case (a=b=0, the
    only one the drivers use: P1:260, P2:355, P3:395) has weights 2/((Q-1) Q P_{Q-1}(x)^2).
    """
    X = _jacobi_roots(Q - 2, a + 1, b + 1)
    if a == 0 and b == 0:
        W = 2 / ((Q - 1) * (Q) * (Jacobi(Q - 1, 0, 0, X) ** 2))
        Wl = 2 / ((Q - 1) * (Q) * (Jacobi(Q - 1, 0, 0, -1) ** 2))
        Wr = 2 / ((Q - 1) * (Q) * (Jacobi(Q - 1, 0, 0, 1) ** 2))
    else:
        g = math.gamma
        c = 2 ** (a + b + 1) * g(a + Q) * g(b + Q) / ((Q - 1) * g(Q) * g(a + b + Q + 1))
        W = c / (Jacobi(Q - 1, a, b, X) ** 2)
        Wl = (b + 1) * c / (Jacobi(Q - 1, a, b, -1) ** 2)
        Wr = (a + 1) * c / (Jacobi(Q - 1, a, b, 1) ** 2)
    W = np.append(W, Wr)
    W = np.append(Wl, W)
    X = np.append(X, 1)
    X = np.append(-1, X)
    return [X, W]
