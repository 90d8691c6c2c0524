import os, sys, time
sys.path.insert(0, os.getcwd())
from hp_vpinns_amd.drivers import advdiff, poisson2d
from hp_vpinns_amd.init import xavier_init
L = [2, 20, 20, 20, 1]
def run(prob, nex, ney, n):
    if prob == "p2vf0":
        s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
        m = poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234))
    else:
        s = advdiff.setup(N_el_x=nex, N_el_t=ney, N_test_x=10, N_test_t=10, N_quad=20, with_test_grid=False)
        m = advdiff.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234, extra=[1.0]))
    m.h.step(64, False)
    t0 = time.perf_counter(); m.h.step(n, False)
    return (time.perf_counter() - t0) / n * 1e6, m.h.pass_structure(), m.h.kernel_variant()
for prob, nex, ney in [("p2vf0", 16, 4), ("advf0", 16, 4), ("p2vf0", 16, 2), ("p2vf0", 23, 24), ("p2vf0", 40, 40)]:
    for fuse in (None, "n", "i"):
        if fuse: os.environ["HPV_FUSE"] = fuse
        else: os.environ.pop("HPV_FUSE", None)
        if fuse == "i" and nex * ney <= 256: continue
        t, ps, kv = run(prob, nex, ney, 1000)
        print("| %s | %dx%d | HPV_FUSE=%s | %.1f | %s | `%s` |" % (prob, nex, ney, fuse, t, ps, kv), flush=True)
