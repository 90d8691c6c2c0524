"""Right-hand-side assembly on the device.

The reference assembles `F_ext_total` with Python loops over elements and test functions
(P1:277-291, P2:386-411): F[e][k][r] = J_e sum_q w_x phi_r(xi_i) w_y phi_k(eta_j) f(x_q).  That is exactly the
forward half of the projection kernel applied to the values of f at the quadrature points, so the
driver evaluates f with numpy (one vectorised call) and the GPU does the projection.  The Gauss-Lobatto-Legendre
rule and the test-function tables come from the device as well (`hpv_gll_rule`, `hpv_test_tables`: Newton / three-term
recurrence kernels), so nothing of the set-up needs scipy or the host Jacobi code.
"""
import numpy as np

from . import _lib


def assemble_F_ext_1d(f_ext, grid, N_test, N_quad, device=0):
    """-> (NE, N_test, 1) like P1:293-294."""
    grid = np.asarray(grid, dtype=np.float64)
    h = _lib.Handle(_lib.PDE_POISSON1D, 1, _lib.ACT_SIN, [1, 1], device=device)
    x, w = h.gll_rule(N_quad)
    h.set_quadrature(x, w)
    h.set_tables(h.test_tables(N_test, x))
    h.set_elements(grid)
    ne = grid.size - 1
    xq = np.concatenate([grid[e] + (grid[e + 1] - grid[e]) / 2 * (x + 1) for e in range(ne)])   # P1:276
    F = h.assemble_rhs(f_ext(xq), ne * N_test)
    h.close()
    return F.reshape(ne, N_test, 1)


def assemble_F_ext_2d(f_ext, grid_x, grid_y, N_test_x, N_test_y, N_quad, device=0):
    """-> (NE_x, NE_y, N_test_y, N_test_x) like P2:414."""
    grid_x, grid_y = np.asarray(grid_x, dtype=np.float64), np.asarray(grid_y, dtype=np.float64)
    h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 1], device=device)
    x, w = h.gll_rule(N_quad)
    h.set_quadrature(x, w, x, w)
    h.set_tables(h.test_tables(N_test_x, x), h.test_tables(N_test_y, x))
    h.set_elements(grid_x, grid_y)
    nex, ney = grid_x.size - 1, grid_y.size - 1
    f = np.empty((nex, ney, N_quad, N_quad))
    for ex in range(nex):
        xq = grid_x[ex] + (grid_x[ex + 1] - grid_x[ex]) / 2 * (x + 1)                  # P2:391
        for ey in range(ney):
            yq = grid_y[ey] + (grid_y[ey + 1] - grid_y[ey]) / 2 * (x + 1)              # P2:392
            f[ex, ey] = f_ext(xq[None, :], yq[:, None])                                # [j][i], x fastest
    F = h.assemble_rhs(f, nex * ney * N_test_y * N_test_x)
    h.close()
    return F.reshape(nex, ney, N_test_y, N_test_x)
