"""Driver-verifiable convergence: the accuracy half of the metric, against the reference's committed figures
 -- full-length training runs through the reference's class surface on one MI355X.

  * BASELINE config 4 (16x16 elements, [2,20,20,20,1]): relative L2 error of u on the driver's 201 x 201 test grid
    <= 1e-2 within 40 000 Adam iterations (seeded Xavier start; 2.7 s of training; lowest-loss checkpoint of the last 10 000).
  * the published 1-D run (3 elements [-1,-0.1,0.1,1], [1,20,20,20,20,1] sin, P1:270-273; Results/loss.pdf, error.pdf):
    the recorded loss reaches <= 1e-4 (the figure bottoms out at ~5e-5; Adam at lr 1e-3 keeps oscillating between 4e-5 and
    ~1e-3 afterwards, here as in the figure) and the max point-wise error after the 40 001 iterations is <= 1.3e-3.
  * the 2-D reference defaults ([2,5,5,5,1], 4x4 elements, 10 001 iterations; Results/Poisson2D_VPINNs_PntErr.png, max
    error ~0.29): a 5-wide network at 10 k iterations is initialisation-dependent (round 2, seeds 0..7: 0.21, 0.43, 1.19,
    1.67, 1.52, 0.54, 0.41, 0.79) -- the published value must lie inside the spread of 8 seeds, at least three of the eight
    must land within 2x of it, and the median must stay below 1.0 (a systematically wrong gradient or table would not).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_reaches_1e_2_relative_l2():
    """BASELINE config 4 from a seeded Xavier start: relative L2 error <= 1e-2 within 40 000 Adam iterations.  The last iterate of
    Adam at lr 1e-3 oscillates (error 8e-3 at 25 000, 1.6e-2 at 30 000, 6e-3 at 35 000 for this seed, and which of them depends on
    the last bit of the arithmetic: scripts/convergence_cfg4.py), so the assertion is on the lowest-loss checkpoint of the last
    10 000 iterations -- what a user who keeps the best iterate gets -- and on the loss level of the tail."""
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    L = [2, 20, 20, 20, 1]
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20)
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
    l0 = m.loss()[0]
    m._step(30000, False)
    best, best_theta, losses = np.inf, None, []
    for _ in range(10):
        l = float(m._step(1000, True)[0])
        losses.append(l)
        if l < best:
            best, best_theta = l, m.get_params()
    # the LAST iterate is asserted too (advisor, round 3: keep a like-for-like figure next to the best-checkpoint one): a plain
    # run's final parameters are within 2x of the best checkpoint's bar at this length (5e-3 .. 1.6e-2 from 25 000 on)
    err_last = m.rel_l2_error(s["X_test"], s["u_test"])
    assert err_last <= 2e-2, (err_last, losses)
    m.set_params(best_theta)
    err = m.rel_l2_error(s["X_test"], s["u_test"])
    assert err <= 1e-2, (err, err_last, losses)
    assert best < 2e-4 * l0 and np.median(losses) < 1e-3 * l0, (best / l0, losses)


def test_published_1d_three_element_run():
    """Adam at lr 1e-3 on the sin network keeps oscillating once the loss is down (here as in Results/loss.pdf): the LAST
    iterate of the 40 001-iteration run has a max error anywhere between 3e-4 and 9e-3 depending on the seed and on the last
    bit of the arithmetic.  The test therefore looks at the whole tail: the run records loss <= 1e-4 inside the published run
    length, and among 100 checkpoints of the last 10 000 iterations the one with the lowest loss has max error <= 1.3e-3."""
    from hp_vpinns_amd.drivers import poisson1d
    r = poisson1d.run(Opt_Niter=30000 + 1, N_Element=3, verbose=False)    # reference defaults otherwise (P1:231-240)
    rec = np.array(r["total_record"])
    assert abs(rec[0, 1] - 408.04) < 1.0          # the ~4e2 plateau of Results/loss.pdf = sum_e mean(F_e^2) + 1
    assert len(rec) == 3001 and rec[-1, 0] == 30000                       # every 10th iteration recorded (P1:210)
    m, s = r["model"], r["setup"]
    best_loss, best_theta = float(m.loss()[0]), m.get_params()
    for _ in range(100):                                                  # iterations 30 001 .. 40 000
        l = float(m._step(100, True)[0])
        if l < best_loss:
            best_loss, best_theta = l, m.get_params()
    assert min(best_loss, rec[:, 1].min()) <= 1e-4, (best_loss, rec[:, 1].min())    # Results/loss.pdf bottoms out at ~5e-5
    m.set_params(best_theta)
    err = np.abs(s["u_test"] - m.predict(s["X_test"])).max()
    assert best_loss <= 1e-4 and err <= 1.3e-3, (best_loss, err)          # Results/error.pdf


def test_2d_reference_defaults_published_error_against_eight_seeds():
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    L = [2, 5, 5, 5, 1]
    errs = []
    for seed in range(8):
        r = poisson2d.run(n_iter=10000 + 1, record_every=100, verbose=False, init_params=xavier_init(L, seed))
        errs.append(float(np.abs(r["setup"]["u_test"] - r["u_pred"]).max()))
        print("seed", seed, "loss", r["loss_his"][0], "->", r["loss_his"][-1], "max err", errs[-1])
        assert r["loss_his"][-1] < 0.05 * r["loss_his"][0]      # (round 3, seeds 0..7: 62.5 -> 0.26 .. 1.15)
    errs_sorted = sorted(errs)
    assert errs_sorted[0] < 0.29 < errs_sorted[-1], errs               # the published figure lies inside the spread
    assert errs_sorted[2] < 0.6, errs                                  # three of eight seeds within 2x of the published 0.29
    assert 0.2 < 0.5 * (errs_sorted[3] + errs_sorted[4]) < 1.0, errs   # median


def test_advdiff_identifies_the_published_diffusion_coefficient():
    """The headline of the third driver (P3:41-42, 63; Results/hpPINN_ADE_Iden_diffcoeff_Avg.pdf): the trainable coefficient, started
    at 1.0, is driven to ~0.032 (exact 0.1 / pi = 0.031831).  Reference defaults (one element, 10 x 10 points, 5 x 5 test functions,
    [2,5,5,5,1], var_form 0, lr 1e-3), 150 001 Adam iterations as in the published figure, five seeded Xavier starts on the same
    data.  Measured (round 5, scripts/advdiff_eps_probe.py, seeds 0..7): +2.3, +15.3, (stalled: 0.127 at a 12x larger loss), +8.3, +24.6,
    +5.3, -9.9, +0.2 % off the exact coefficient with final losses 3e-5 .. 2.7e-4 (round 1's single run: -12 %).  A [2,20,20,20,1]
    network on the same data lands at +0.5 .. +20 %, on 4 x 2 elements at +19 .. +26 %: the spread is what 15 interior measurements
    and 150 k Adam iterations at lr 1e-3 determine, not the network's capacity and not the kernels (the restated loss itself is
    minimised by the exact coefficient when the exact solution is put in: tests/test_oracle.py, 1.35e-5 at 5 % off against 3e-13).
    So: most starts must reach the published loss level (~2e-4, Results/hpPINN_ADE_Iden_loss.pdf), every converged start must land
    within 30 % of 0.1 / pi and their median within 15 % -- the published ~0.032 lies inside the spread."""
    from hp_vpinns_amd.drivers import advdiff
    from hp_vpinns_amd.init import xavier_init
    L = [2, 5, 5, 5, 1]
    s = advdiff.setup(with_test_grid=False)
    exact = 0.1 / np.pi
    eps, losses = [], []
    for seed in range(6):
        m = advdiff.build_model(s, L, var_form=0, init_params=xavier_init(L, seed, extra=[1.0]))
        assert abs(float(m.epsilon[0]) - 1.0) < 1e-15                      # P3:63
        m._step(150000, False)
        l = m._step(1, True)
        eps.append(float(m.epsilon[0]))
        losses.append(float(l[0]))
        print("seed", seed, "identified epsilon", eps[-1], "(exact %.6f)" % exact, "loss", losses[-1])
    eps, losses = np.array(eps), np.array(losses)
    conv = losses < 5e-4                                                        # the published loss level, ~2e-4
    assert conv.sum() >= 4, (eps, losses)
    assert np.all(np.abs(eps[conv] - exact) < 0.30 * exact), (eps, losses)
    assert abs(np.median(eps[conv]) - exact) < 0.15 * exact, (eps, losses)
    assert eps[conv].min() < 1.06 * exact, (eps, losses)                        # ... and some start gets close (the published value)
    assert conv.all() or np.all(losses[~conv] > 3 * losses[conv].max()), (eps, losses)   # (a stalled start is recognisable by its loss)
