#!/usr/bin/env python3
"""Long training runs against the reference's published figures: loss and error levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hp_vpinns_amd.drivers import advdiff, poisson1d, poisson2d

t0 = time.time()
r = poisson1d.run(Opt_Niter=40000 + 1, N_Element=3, verbose=False)        # reference defaults otherwise (P1:231-240)
rec = np.array(r["total_record"])
err = np.abs(r["setup"]["u_test"] - r["u_pred"]).max()
print("Poisson-1D, 3 elements, [1,20,20,20,20,1] sin, 40 001 Adam its: loss %.3e -> %.3e (min %.3e), max|u-u_NN| = %.2e, rel L2 = %.2e, %.1f s"
      % (rec[0, 1], rec[-1, 1], rec[:, 1].min(), err, r["rel_l2"], time.time() - t0))
print("   reference figures: loss ~4e2 plateau -> ~5e-5 at 40k its (Results/loss.pdf); pointwise error <= ~1.3e-3 (Results/error.pdf)")
t0 = time.time()
r = poisson2d.run(n_iter=10000 + 1, record_every=100, verbose=False)       # reference defaults: 4x4 elements, [2,5,5,5,1]
err = np.abs(r["setup"]["u_test"] - r["u_pred"]).max()
print("Poisson-2D reference defaults (4x4 el, [2,5,5,5,1]), 10 001 its: loss %.3e -> %.3e, max|u-u_NN| = %.2e, rel L2 = %.2e, %.1f s"
      % (r["loss_his"][0], r["loss_his"][-1], err, r["rel_l2"], time.time() - t0))
print("   reference figure: max point-wise error ~0.29 (Results/Poisson2D_VPINNs_PntErr.png)")
t0 = time.time()
r = poisson2d.run(n_iter=50000 + 1, N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, Net_layer=[2, 20, 20, 20, 1],
                  record_every=1000, verbose=False)
err = np.abs(r["setup"]["u_test"] - r["u_pred"]).max()
print("Poisson-2D BASELINE config 4 (16x16 el, 20x20 quad, 10x10 test, [2,20,20,20,1]), 50 001 its: loss %.3e -> %.3e, max err %.2e, rel L2 = %.2e, %.1f s"
      % (r["loss_his"][0], r["loss_his"][-1], err, r["rel_l2"], time.time() - t0))
t0 = time.time()
r = advdiff.run(Opt_Niter=150000 + 1, verbose=False)                       # reference defaults (P3:31-54), figure run length
print("AdvDiff identification, reference defaults, 150 001 its: loss -> %.3e, identified epsilon = %.5f (exact %.5f), rel L2 = %.2e, %.1f s"
      % (r["total_record"][-1][1], r["epsilon"], advdiff.epsilon, r["rel_l2"], time.time() - t0))
print("   reference figures: loss ~2 -> ~2.2e-4; epsilon 1.0 -> ~0.032 (exact 0.031831)")
