"""The MFMA path for hidden widths other than 20 (csrc/kernels_wide.hip; verdict round 3, missing 3: "remove the 75x cliff").

The reference takes any `Net_layer` (P1:236, P2:280-286, P3:46-51).  Every hidden width up to 64 -- uniform or not, a multiple
of 4 or not (the classes zero-pad to the next instantiated width: exact) -- must run on MFMA kernels and agree with the oracle:
loss triple, gradient (incl. d/d epsilon), a TF1-Adam trajectory and the final parameters, for every channel set of the three
problems, 1..4 hidden layers, point counts that are not a multiple of the 16-point tile."""
import numpy as np
import pytest

from cases import gold, p1_args, p2_args, p3_args, rel, theta0
from test_gpu_parity import TOL, TRAJ_TOL, _check_loss_grad, _check_traj, _pair_1d, _pair_2d, _pair_adv

pytestmark = pytest.mark.gpu


def _is_wide(m, H):
    v = m.h.kernel_variant()
    assert m.backend() == "mfma" and "k_fwd_wide" in v and "k_bwd_wide" in v and f"H={H}>" in v, v


@pytest.mark.parametrize("H", [24, 32, 40, 48, 64])
@pytest.mark.parametrize("kind,vf", [("1d", 1), ("1d", 2), ("1d", 3), ("2d", 0), ("2d", 1), ("2d", 2), ("adv", 0), ("adv", 1)])
def test_wide_networks_all_channel_sets(kind, vf, H):
    if kind == "1d":
        o, m = _pair_1d("poisson1d_small", vf, layers=[1, H, H, H, 1], backend="mfma")
    elif kind == "2d":
        o, m = _pair_2d("poisson2d_small", vf, layers=[2, H, H, H, 1], backend="mfma")
    else:
        o, m = _pair_adv("advdiff_small", vf, layers=[2, H, H, H, 1], backend="mfma")
    _check_loss_grad(o, m)
    _is_wide(m, H)
    _check_traj(o, m, n=6)


@pytest.mark.parametrize("H,nhid", [(24, 1), (32, 1), (32, 2), (32, 4), (40, 2), (40, 4), (64, 1), (64, 2)])
def test_wide_networks_depths(H, nhid):
    o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [H] * nhid + [1], backend="mfma")
    _check_loss_grad(o, m)
    _is_wide(m, H)
    o, m = _pair_1d("poisson1d_small", 1, layers=[1] + [H] * nhid + [1], backend="mfma")
    _check_loss_grad(o, m)
    _is_wide(m, H)
    _check_traj(o, m, n=4)


@pytest.mark.parametrize("layers,H", [([2, 30, 30, 1], 32), ([2, 20, 40, 20, 1], 40), ([2, 21, 7, 1], 24), ([2, 33, 50, 12, 1], 64)])
def test_widths_that_are_padded_to_the_next_instantiated_one(layers, H):
    """Non-uniform widths and widths that are no multiple of 4: zero-padded (exact), parameters / gradients in the USER's layout."""
    o, m = _pair_adv("advdiff_small", 0, layers=layers)
    assert m.get_params().size == o.get_params().size
    _check_loss_grad(o, m)
    _is_wide(m, H)
    _check_traj(o, m, n=8)
    assert m.h.num_params() > m.get_params().size      # the device holds the padded network


@pytest.mark.parametrize("H", [32, 40])
def test_config4_grid_with_wider_networks_against_the_oracle(H):
    """The verdict's named cases: [2,32,32,32,1] and [2,40,40,40,1] on the config-4 grid (102 400 points), full size."""
    from hp_vpinns_amd.vpinn import VPINN2D
    from oracle.vpinn_oracle import OracleVPINN2D
    L = [2, H, H, H, 1]
    a = p2_args(gold("poisson2d_cfg4"), layers=L)
    th = theta0(L, 77)
    o, m = OracleVPINN2D(*a, init_params=th), VPINN2D(*a, init_params=th)
    o.vectorized = True
    l3o, go = o.loss_and_grad()
    l3m, gm = m.loss_and_grad()
    _is_wide(m, H)
    assert "k_project_wg<20x20/10x10>" in m.h.kernel_variant()     # 256 elements: one workgroup per element
    assert rel(l3m, l3o) < TOL and rel(gm, go) < TOL, (l3m, l3o, rel(gm, go))
    assert rel(m.h.residuals(25600), o.last["R"].reshape(-1)) < TOL
    _check_traj(o, m, n=5)


def test_one_dimensional_four_hidden_layers_reference_default_depth():
    """P1:236's own depth (`Net_layer = [1] + [20] * 4 + [1]`) at the reference width -- the hand-tuned path -- and at 32."""
    for H in (20, 32):
        o, m = _pair_1d("poisson1d_cfg2", 1, layers=[1] + [H] * 4 + [1])
        _check_loss_grad(o, m)
        assert m.backend() == "mfma"
        _check_traj(o, m, n=6)


def test_generic_fallback_warns_once_for_networks_beyond_every_mfma_kernel():
    with pytest.warns(UserWarning, match="generic kernels"):
        o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [20] * 7 + [1])     # seven hidden layers: deeper than any MFMA kernel (1..6)
    assert m.backend() == "generic"
    _check_loss_grad(o, m)
    with pytest.warns(UserWarning, match="generic kernels"):
        o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [40] * 5 + [1])     # five hidden layers at a width beyond 32 (1..4 there)
    assert m.backend() == "generic"


@pytest.mark.parametrize("depth", [5, 6])
def test_five_and_six_hidden_layers_on_the_mfma_kernels(depth):
    """PINN-style deeper networks: 5 and 6 hidden layers of width <= 20 run on k_fwd_mfma / k_bwd_mfma (kernels_mfma.hip
    instantiates L = 1..6), narrower layers zero-padded as always -- 1-D sin (P1) and 2-D tanh (P2, config-4 element shape with the
    projection fused into the reverse kernel; a 10x10 grid on the separate launches) against the oracle."""
    o, m = _pair_1d("poisson1d_cfg2", 1, layers=[1] + [20] * depth + [1])
    _check_loss_grad(o, m)
    assert m.backend() == "mfma" and f"L={depth}" in m.h.kernel_variant(), m.h.kernel_variant()
    _check_traj(o, m, n=5)
    o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [12] * depth + [1])
    _check_loss_grad(o, m)
    assert m.backend() == "mfma" and f"L={depth}" in m.h.kernel_variant(), m.h.kernel_variant()
    _check_traj(o, m, n=5)
    o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [30] * depth + [1])          # (padded to 32: the width-generic kernels)
    _check_loss_grad(o, m)
    assert m.backend() == "mfma" and f"L={depth},H=32" in m.h.kernel_variant(), m.h.kernel_variant()
    _check_traj(o, m, n=5)
    from hp_vpinns_amd.vpinn import VPINN2D
    from oracle.vpinn_oracle import OracleVPINN2D
    L = [2] + [20] * depth + [1]
    a = p2_args(gold("poisson2d_cfg4"), layers=L)
    th = theta0(L, 78)
    o, m = OracleVPINN2D(*a, init_params=th), VPINN2D(*a, init_params=th)
    o.vectorized = True
    l3o, go = o.loss_and_grad()
    l3m, gm = m.loss_and_grad()
    assert f"L={depth}" in m.h.kernel_variant() and "proj=20x20/10x10" in m.h.kernel_variant(), m.h.kernel_variant()
    assert rel(l3m, l3o) < TOL and rel(gm, go) < TOL, (l3m, l3o, rel(gm, go))
