# Phase stamps inside project_element_wg as k_iter_tile runs it (needs HPV_EXTRA_FLAGS="-DHPV_FZ_TIMING -DHPV_PJ_TIMING" bash hp_vpinns_amd/csrc/build.sh): pj_timing.py t2|t5b
import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from hp_vpinns_amd.init import xavier_init
mode = sys.argv[1]
if mode == "t2":
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.vpinn import VPINN1D
    s = poisson1d.setup(N_Element=16); L1 = [1, 20, 20, 20, 1]
    m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"], s["u_test"], L1, s["X_f_train"], s["f_train"], init_params=xavier_init(L1, 1234)); NB = 16
else:
    from hp_vpinns_amd.drivers import advdiff
    L = [2, 20, 20, 20, 1]
    s = advdiff.setup(N_el_x=8, N_quad=10, with_test_grid=False)
    m = advdiff.build_model(s, L, init_params=xavier_init(L, 1234, extra=[1.0])); NB = 8
m.h.step(50, False)
out = np.empty(NB * 16)
m.h.lib.hpv_debug_read_out.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
m.h.lib.hpv_debug_read_out(m.h._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
t = out.reshape(NB, 16)[:, :8]
d = np.diff(t, axis=1).mean(axis=0)
if mode == "t2":   # project_element_1d: five stamps, the kernel's own at [7]
    print(mode, "integrands+barrier, residual+barrier, adjoint contraction+barrier, adjoint channels, (return):", np.round(d[:4]).tolist(), "total", round(float((t[:, 7] - t[:, 0]).mean())))
else:
    print(mode, "start->tables-sync, ->G stored, ->contraction 1, ->contraction 2, ->R/loss, ->S, ->end:", np.round(d).tolist(), "total", round(float((t[:, 7] - t[:, 0]).mean())))
