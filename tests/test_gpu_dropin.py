"""The drop-in boundary on the real library, in the reference drivers' statement order.

The reference's `__main__` blocks build the model FIRST and create the history list `train` appends to AFTERWARDS
(P1:333 `model = VPINN(...)`, P1:335 `total_record = []`, P1:336 `model.train(...)`; P2:430 / 433 / 434), and never pass
`var_form` / `LR` / `lossb_weight` / `scheme` / `V`: the class reads them as globals of the module it lives in.  The
mini-drivers below are this repo's own text, written in that order and exec'd as a module of their own; the oracle
(test infrastructure) checks the numbers that end up in the lists.  `tests/test_reference_dropin.py` does the same with
the reference's own scripts in the build container.
"""
import numpy as np
import pytest

from cases import gold, p1_args, p2_args, p3_args, rel, theta0

pytestmark = pytest.mark.gpu

DRIVER_1D = """
from hp_vpinns_amd.vpinn import VPINN1D as VPINN        # the one-line binding of INTEGRATION.md 1
LR = 0.002
var_form = 2
lossb_weight = 3
model = VPINN(*args, init_params=theta)                 # P1:333-334: no list, no hyper-parameters handed in
total_record = []                                       # P1:335: created AFTER the constructor
model.train(Opt_Niter, 2e-32)                           # P1:336
iteration = [total_record[i][0] for i in range(len(total_record))]     # P1:392-393
loss_his = [total_record[i][1] for i in range(len(total_record))]
"""

DRIVER_2D = """
from hp_vpinns_amd.vpinn import VPINN2D as VPINN
scheme = 'VPINNs'
var_form = 2
model = VPINN(*args, init_params=theta)                 # P2:430-431
u_pred_his, loss_his = [], []                           # P2:433: created AFTER the constructor
model.train(n_iter)                                     # P2:434
u_pred = model.predict()                                # P2:435
"""

DRIVER_ADVDIFF = """
from hp_vpinns_amd.vpinn import VPINNAdvDiff as VPINN
LR = 0.002
var_form = 1
V = 0.5
model = VPINN(*args, init_params=theta)                 # P3:488-489
error_record, total_record, u_record, u_records_iterhis, total_time_train = model.train(Opt_Niter, 2e-32)   # P3:493-494
"""


def _oracle_trajectory(o, n):
    out = []
    for _ in range(n):
        o.adam_step()
        out.append(float(o.loss_and_grad()[0][0]))
    return np.array(out)


def test_poisson1d_driver_order_fills_the_list_created_after_the_constructor():
    from oracle.vpinn_oracle import OracleVPINN1D
    g = gold("poisson1d_small")
    a = p1_args(g)
    th = theta0(a[8], 11)
    w = int(a[8][1])
    th[w:2 * w] = 0.05 * np.arange(w)    # non-zero first bias: an odd sin network makes d loss / d b_out pure round-off, which Adam amplifies
    ns = {"args": a, "theta": th, "Opt_Niter": 31, "__name__": "mini_driver_1d"}
    exec(compile(DRIVER_1D, "<mini driver 1-D>", "exec"), ns)
    rec = ns["total_record"]
    assert [int(r[0]) for r in rec] == [0, 10, 20, 30] and rec is ns["model"].total_record
    o = OracleVPINN1D(*a, init_params=th, var_form=2, LR=0.002, lossb_weight=3)      # the mini-driver's module globals
    traj = _oracle_trajectory(o, 31)
    assert rel(ns["loss_his"], traj[[0, 10, 20, 30]]) < 1e-7
    assert rel(ns["model"].get_params(), o.get_params()) < 1e-7
    # a second model in a namespace that never creates the list keeps a private one (no NameError, nothing shared)
    ns2 = {"args": a, "theta": th, "__name__": "mini_driver_1d_b"}
    exec("from hp_vpinns_amd.vpinn import VPINN1D as VPINN\nmodel = VPINN(*args, init_params=theta)\nmodel.train(11, 0.0)\n", ns2)
    assert len(ns2["model"].total_record) == 2 and "total_record" not in ns2 and len(rec) == 4


def test_poisson2d_driver_order_fills_the_list_created_after_the_constructor():
    from oracle.vpinn_oracle import OracleVPINN2D
    g = gold("poisson2d_small")
    a = p2_args(g, layers=[2, 20, 20, 20, 1])
    th = theta0(a[13], 5)
    ns = {"args": a, "theta": th, "n_iter": 12, "__name__": "mini_driver_2d"}
    exec(compile(DRIVER_2D, "<mini driver 2-D>", "exec"), ns)
    his = ns["loss_his"]
    assert len(his) == 12 and his is ns["model"].loss_his
    o = OracleVPINN2D(*a, init_params=th, var_form=2)
    o.vectorized = True        # (the batched restatement: same arithmetic, tests/test_oracle.py compares the two)
    assert rel(his, _oracle_trajectory(o, 12)) < 1e-7
    assert ns["u_pred"].shape == (a[11].shape[0], 1)
    # an explicit list (what hp_vpinns_amd/drivers do) still wins over the module's
    ns3 = dict(ns, mine=[], __name__="mini_driver_2d_c")
    exec("model = VPINN(*args, init_params=theta, loss_his=mine)\nloss_his = []\nmodel.train(3)\n", ns3)
    assert len(ns3["mine"]) == 3 and ns3["loss_his"] == []


def test_advdiff_driver_module_globals_reach_the_library():
    from oracle.vpinn_oracle import OracleVPINNAdvDiff
    g = gold("advdiff_default")
    a = p3_args(g)
    th = theta0(a[12], 3, extra=[1.0])
    ns = {"args": a, "theta": th, "Opt_Niter": 21, "__name__": "mini_driver_ad"}
    exec(compile(DRIVER_ADVDIFF, "<mini driver AdvDiff>", "exec"), ns)
    rec = ns["total_record"]
    assert [int(r[0]) for r in rec] == [0, 10, 20]
    o = OracleVPINNAdvDiff(*a, init_params=th, var_form=1, LR=0.002, V=0.5)
    o.vectorized = True
    traj = _oracle_trajectory(o, 21)
    assert rel([r[1] for r in rec], traj[[0, 10, 20]]) < 1e-7
    assert abs(float(rec[-1][2][0]) - float(o.get_params()[-1])) < 1e-9
    assert ns["error_record"][0] == rec[-1][1]
