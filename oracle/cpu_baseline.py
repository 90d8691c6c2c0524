"""ctypes binding of oracle/cpu_closed_form.c -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.

Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this module (same rule as vpinn_oracle.py).
The C file is baseline "B" of BASELINE.md section 3: one Poisson-2D var_form-1 training iteration (P2:68-132) in closed
form with OpenMP over elements; tests/test_oracle.py pins it to the autograd oracle before anything is timed with it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .vpinn_oracle import Test_fcn, dTest_fcn

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libhpvc.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None


def build():
    subprocess.run(["bash", os.path.join(_HERE, "build_cpu_baseline.sh")], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        common = [_ip, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp,
                  C.c_int, C.c_double]
        _lib.hpvc_loss_grad.argtypes = [_dp] + common + [C.c_int, _dp, _dp]
        _lib.hpvc_train.argtypes = [_dp, _dp, _dp, _dp] + common + [C.c_double, C.c_int, C.c_int, _dp]
        _lib.hpvc_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class CPoisson2D:
    """Poisson-2D var_form 1 on a tensor grid, arguments as the reference's VPINN(...) constructor takes them (P2:28-29)."""

    def __init__(self, X_u_train, u_train, X_quad, W_quad, F_exact_total, gridx, gridy, layers, init_params,
                 lossb_weight=10.0, LR=0.001, threads=0):
        self.lib = load()
        X_quad, W_quad = np.asarray(X_quad, dtype=np.float64), np.asarray(W_quad, dtype=np.float64)
        Q = int(round(np.sqrt(X_quad.shape[0])))
        self.Q = Q
        self.xi = np.ascontiguousarray(X_quad[:Q, 0])
        self.wq = np.ascontiguousarray(W_quad[:Q, 0])
        F = np.asarray(F_exact_total, dtype=np.float64)
        self.nex, self.ney, self.nty, self.ntx = F.shape
        self.F = np.ascontiguousarray(F.reshape(-1))

        def tab(n):                                   # [2][n][Q]: phi, phi' at the reference nodes (scipy, as the reference)
            t0 = Test_fcn(n, self.xi[:, None])[:, :, 0]
            t1 = dTest_fcn(n, self.xi[:, None])[0][:, :, 0]
            return np.ascontiguousarray(np.stack([t0, t1]))
        self.tabx, self.taby = tab(self.ntx), tab(self.nty)
        self.gridx = np.ascontiguousarray(gridx, dtype=np.float64)
        self.gridy = np.ascontiguousarray(gridy, dtype=np.float64)
        self.Xd = np.ascontiguousarray(X_u_train, dtype=np.float64)
        self.ud = np.ascontiguousarray(np.asarray(u_train, dtype=np.float64).reshape(-1))
        self.layers = (C.c_int * len(layers))(*[int(v) for v in layers])
        self.nl = len(layers)
        self.theta = np.array(init_params, dtype=np.float64).reshape(-1).copy()
        self.m, self.v = np.zeros_like(self.theta), np.zeros_like(self.theta)
        self.bpow = np.array([0.9, 0.999])
        self.lossb_weight, self.LR, self.threads = float(lossb_weight), float(LR), int(threads)

    def _common(self):
        return [self.layers, self.nl, _p(self.xi), _p(self.wq), self.Q, _p(self.gridx), self.nex, _p(self.gridy), self.ney,
                _p(self.tabx), self.ntx, _p(self.taby), self.nty, _p(self.F), _p(self.Xd), _p(self.ud), self.ud.size,
                self.lossb_weight]

    def loss_and_grad(self):
        l3, g = np.empty(3), np.empty_like(self.theta)
        rc = self.lib.hpvc_loss_grad(_p(self.theta), *self._common(), self.threads, _p(l3), _p(g))
        assert rc == 0, rc
        return l3, g

    def train(self, n):
        """n TF1-Adam iterations; returns the loss triples of the n forward passes (each BEFORE its update)."""
        hist = np.empty((n, 3))
        rc = self.lib.hpvc_train(_p(self.theta), _p(self.m), _p(self.v), _p(self.bpow), *self._common(), self.LR,
                                 self.threads, int(n), _p(hist))
        assert rc == 0, rc
        return hist

    def max_threads(self):
        return int(self.lib.hpvc_max_threads())
