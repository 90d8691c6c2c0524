#!/usr/bin/env python3
"""Would padding a 17..19-point rule onto the 20x20 tight plan (four channels, three hidden layers) beat the rule's own launches?
pad_probe.py: Poisson-2D / AdvDiff var_form 0 on 16x16 elements, [2,20,20,20,1], default (padding onto 20 rejected) against forced."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hp_vpinns_amd.vpinn as V  # noqa: E402
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
orig = V._device_rule_2d


def forced(*a, **k):
    k["reject"] = (10,)
    return orig(*a, **k)


for prob in ("p2vf0", "advf0"):
    for q in (17, 18, 19):
        nt = q // 2
        for mode in ("default", "padded onto 20x20"):
            V._device_rule_2d = orig if mode == "default" else forced
            if prob == "p2vf0":
                s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
                m = poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234))
            else:
                s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=nt, N_test_t=nt, N_quad=q, with_test_grid=False)
                m = advdiff.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234, extra=[1.0]))
            m.h.step(64, False)
            t0 = time.perf_counter()
            m.h.step(1000, False)
            print("| %s | %dx%d / %dx%d | %s | %.1f | %s | `%s` |" % (prob, q, q, nt, nt, mode, (time.perf_counter() - t0) / 1000 * 1e6,
                                                                  m.h.pass_structure(), m.h.kernel_variant()), flush=True)
V._device_rule_2d = orig
