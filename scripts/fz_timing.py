import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
# needs a library built with HPV_EXTRA_FLAGS=-DHPV_FZ_TIMING bash hp_vpinns_amd/csrc/build.sh
import numpy as np
from hp_vpinns_amd.drivers import poisson2d
from hp_vpinns_amd.init import xavier_init
LAYERS = [2, 20, 20, 20, 1]
mode = sys.argv[1] if len(sys.argv) > 1 else "4"
if mode == "seg":    # k_iter_fused at config 4: segments of the reverse tile body (sums over a wave's tiles / its tile count)
    os.environ["HPV_DEBUG_READ_CHANNELS"] = "1"
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    m = poisson2d.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234))
    m.h.step(50, False)
    out = np.empty(256 * 4 * 12)
    m.h.lib.hpv_debug_read_out.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
    m.h.lib.hpv_debug_read_out(m.h._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
    t = out.reshape(256, 4, 12)
    names = ["fetch+recompute", "head", "L3 zbar+transposes+hbar", "L3 dW", "L2 zbar+transposes+hbar", "L2 dW", "-", "L1 zbar+dW1+loop"]
    for w in range(4):
        nt = t[:, w, 8]
        print("wave", w, "tiles %.2f:" % nt.mean(), {n: round(float((t[:, w, i] / nt).mean()), 1) for i, n in enumerate(names) if n != "-"},
              "sum/tile %.0f" % float((t[:, w, :8].sum(axis=1) / nt).mean()))
    sys.exit(0)
if mode == "c5":     # k_iter_tall (kernels_tall.hip): BASELINE config 5, 8 elements x 80x80 points, 32 workgroups per element
    from hp_vpinns_amd.drivers import advdiff
    s = advdiff.setup(N_el_x=8, N_quad=80, with_test_grid=False)
    m = advdiff.build_model(s, LAYERS, init_params=xavier_init(LAYERS, 1234, extra=[1.0]))
    m.h.step(50, False)
    out = np.empty(256 * 4 * 12)
    m.h.lib.hpv_debug_read_out.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
    m.h.lib.hpv_debug_read_out(m.h._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
    t = out.reshape(256, 4, 12)
    names = ["staging", "forward", "wait", "partial-proj", "barrier", "residual+adjoint", "reverse", "wait", "epilogue", "total-us", "tiles"]
    for nt in (3.0, 4.0):      # (quarter-tile plan: every wave owns 3 whole tiles + a quarter)
        sel = t[:, :, 10] == nt
        if sel.any():
            print("waves with %d whole tiles (%d):" % (nt, sel.sum()), {n: round(float(t[:, :, i][sel].mean()), 1) for i, n in enumerate(names)})
    top = t[:, :, 10].max()
    bw = t[:, :, 4][t[:, :, 10] == top]
    print("exchange wait ('barrier') of the waves with %d tiles: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f cycles"
          % (top, bw.min(), np.percentile(bw, 10), np.median(bw), np.percentile(bw, 90), bw.max()))
    fw = t[:, :, 1].max(axis=1)
    print("forward phase (max over a workgroup's waves): min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f cycles; staging min %.0f median %.0f max %.0f"
          % (fw.min(), np.percentile(fw, 10), np.median(fw), np.percentile(fw, 90), fw.max(), t[:, 0, 0].min(), np.median(t[:, 0, 0]), t[:, 0, 0].max()))
    print("structure", m.h.pass_structure())
    sys.exit(0)
small = mode == "3"        # config 3 (k_iter_small: 64 workgroups of 8 waves) instead of config 4
shard = int(mode[1:]) if mode.startswith("s") else 1     # "s8": the 32-element shard one of 8 GPUs owns (split kernel)
if mode.startswith("t"):   # k_iter_tile (kernels_tile.hip): t1 / t2 = Poisson-1D with 1 / 16 elements, t5b = AdvDiff 8 elements, 10x10 rule
    os.environ["HPV_DEBUG_READ_STORE"] = "1"
    if mode in ("t1", "t2"):
        from hp_vpinns_amd.drivers import poisson1d
        from hp_vpinns_amd.vpinn import VPINN1D
        ne = 1 if mode == "t1" else 16
        s = poisson1d.setup(N_Element=ne)
        L1 = [1, 20, 20, 20, 1]
        m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"],
                    s["X_test"], s["u_test"], L1, s["X_f_train"], s["f_train"], init_params=xavier_init(L1, 1234))
        NB, NW = ne, 6
    else:
        from hp_vpinns_amd.drivers import advdiff
        s = advdiff.setup(N_el_x=8, N_quad=10, with_test_grid=False)
        m = advdiff.build_model(s, LAYERS, init_params=xavier_init(LAYERS, 1234, extra=[1.0]))
        NB, NW = 8, 8
elif mode in ("g0", "g1"):    # the general forms on the config-4 grid: g0 = Poisson-2D var_form 0 (four channels, the tight plan), g1 = AdvDiff var_form 1
    if mode == "g0":
        s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
        m = poisson2d.build_model(s, LAYERS, var_form=0, init_params=xavier_init(LAYERS, 1234))
    else:
        from hp_vpinns_amd.drivers import advdiff
        s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=10, N_test_t=10, N_quad=20, with_test_grid=False)
        m = advdiff.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234, extra=[1.0]))
    NB, NW = 256, 4
else:
    if small:
        s = poisson2d.setup(N_el_x=8, N_el_y=8, with_test_grid=False)
    else:
        s = poisson2d.setup(N_el_x=16, N_el_y=16 // shard, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    m = poisson2d.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234))
    NB, NW = (64, 8) if small else (256, 4)
# (a shard of 256 / n elements runs 256 workgroups too: n per element)
m.h.step(50, False)
print(m.h.kernel_variant())
out = np.empty(NB * NW * 10)
m.h.lib.hpv_debug_read_out.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
m.h.lib.hpv_debug_read_out(m.h._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
t = out.reshape(NB, NW, 10)
names = ["stage-sync", "fwd", "wait-after-fwd", "proj", "rev", "wait-after-rev", "epilogue", "total"]
print("shader cycles, mean over blocks, per wave (k_iter_fused: 'total' = microseconds on the 100 MHz wall clock):")
for w in range(NW):
    print("wave", w, {n: round(float(t[:, w, i].mean()), 1) for i, n in enumerate(names)})
t0 = t[:, :, 8].min()
st, en = t[:, :, 8] - t0, t[:, :, 9] - t0
print("wall clock, us after the first wave's start: starts  min %.2f  median %.2f  max %.2f | ends  min %.2f  median %.2f  max %.2f"
      % (st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max()))
print("start of workgroup b (wave 0), us:", np.round(st[:, 0][:: max(1, NB // 32)], 2).tolist())
late = np.argsort(-en.max(axis=1))[:6]
for b in late.tolist() + [int(np.argsort(en.max(axis=1))[NB // 2])]:
    w = int(np.argmax(en[b]))
    print("workgroup %3d wave %d ends %.2f us:" % (b, w, en[b, w]), {n: round(float(t[b, w, i]), 1) for i, n in enumerate(names)})
