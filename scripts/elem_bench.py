#!/usr/bin/env python3
"""Iteration time of element shapes / variational forms other than the five BASELINE configs': the default dispatch (the
whole-iteration kernel k_iter_fused for the two-term forms on 16x16 / 8x8 and 12x12 / 6x6 elements since round 4, otherwise the
faster of the next two) against the generic element-resident kernel (csrc/kernels_elem.hip, HPV_FUSE=e) and the separate launches
(HPV_FUSE=n: forward -> activation store -> projection -> reverse -> finalize).  Prints a markdown table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402


def timeit(m, n=400):
    m._step(50, False)
    m.h.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m._step(n, False)
        m.h.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return 1e6 * best


def with_fuse(val, build):
    prev = os.environ.get("HPV_FUSE")
    if val is None:
        os.environ.pop("HPV_FUSE", None)
    else:
        os.environ["HPV_FUSE"] = val
    try:
        m = build()
        return timeit(m), m.h.kernel_variant(), m.h.pass_structure()
    finally:
        if prev is None:
            os.environ.pop("HPV_FUSE", None)
        else:
            os.environ["HPV_FUSE"] = prev


def both(build):
    """default dispatch, the generic element-resident kernel forced (HPV_FUSE=e), the separate launches (HPV_FUSE=n)"""
    us, v, ps = with_fuse(None, build)
    use, ve, pse = with_fuse("e", build)
    usn, vn, _ = with_fuse("n", build)
    return us, v, ps, use, (ve if pse == "whole-iteration-element" else "-"), usn


rows = []
for (q, nt, nex, ney, L, vf) in [(16, 8, 16, 16, [2, 20, 20, 20, 1], 1), (16, 8, 16, 16, [2, 20, 20, 1], 1), (16, 8, 16, 4, [2, 20, 20, 20, 1], 1),
                                (12, 6, 16, 16, [2, 20, 20, 20, 1], 1), (12, 6, 16, 16, [2, 20, 20, 1], 1), (12, 6, 16, 4, [2, 20, 20, 20, 1], 1),
                                (16, 8, 32, 32, [2, 20, 20, 20, 1], 1), (12, 6, 32, 32, [2, 20, 20, 20, 1], 1),
                                (16, 8, 16, 16, [2, 20, 20, 20, 1], 0), (20, 10, 16, 16, [2, 20, 20, 20, 1], 0), (20, 10, 16, 16, [2, 20, 20, 20, 1], 2),
                                (12, 6, 16, 16, [2, 20, 20, 20, 1], 0), (16, 8, 16, 4, [2, 20, 20, 20, 1], 0),
                                (16, 8, 16, 16, [2, 32, 32, 32, 1], 1)]:
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
    r = both(lambda: poisson2d.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234)))
    rows.append((f"Poisson-2D var_form {vf}, {nex}x{ney} elements, {q}x{q} points, {nt}x{nt} test fcns, {L}", nex * ney * q * q) + r)
L = [2, 20, 20, 20, 1]
for (q, nt, nex, net, vfs) in [(16, 8, 16, 16, (0, 1)), (12, 6, 16, 16, (0, 1)), (20, 10, 16, 16, (0, 1)), (16, 8, 16, 4, (0, 1))]:
    s = advdiff.setup(N_el_x=nex, N_el_t=net, N_test_x=nt, N_test_t=nt, N_quad=q, with_test_grid=False)
    for vf in vfs:
        r = both(lambda: advdiff.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234, extra=[1.0])))
        rows.append((f"AdvDiff var_form {vf} (trainable epsilon), {nex}x{net} elements, {q}x{q} points, {nt}x{nt} test fcns, {L}", nex * net * q * q) + r)
print("| problem | points | default: us / iteration | kernel | generic element-resident kernel (HPV_FUSE=e) | separate launches (HPV_FUSE=n) |\n|---|---|---|---|---|---|")
for name, npt, us, v, ps, use, ve, usn in rows:
    print(f"| {name} | {npt} | **{us:.1f}** ({ps}) | `{v}` | {use:.1f} `{ve}` | {usn:.1f} |")

# quadrature rules / test-function counts between the instantiated ones: the rule goes to the device padded with zero-weight
# points (vpinn._pad_rule), the counts are run-time values of the kernels -- against the same problem on the general launches
print()
print("| problem (Poisson-2D var_form 1, [2, 20, 20, 20, 1]) | default: us / iteration | kernel | rule as it is (HPV_NO_RULE_PADDING=1) | kernels |\n|---|---|---|---|---|")
for (q, nt, ne) in [(14, 7, 16), (18, 9, 16), (11, 5, 16), (7, 4, 16), (20, 7, 16), (16, 5, 16)]:
    L = [2, 20, 20, 20, 1]
    s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
    build = lambda: poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
    us, v, ps = with_fuse(None, build)
    os.environ["HPV_NO_RULE_PADDING"] = "1"
    try:
        us2, v2, _ = with_fuse(None, build)
    finally:
        del os.environ["HPV_NO_RULE_PADDING"]
    print(f"| {ne}x{ne} elements, {q}x{q} points, {nt}x{nt} test fcns | **{us:.1f}** | `{v}` | {us2:.1f} | `{v2}` |")
