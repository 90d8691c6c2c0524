"""hp_vpinns_amd -- MI355X-native hp-VPINN training path.

The directory is spelled with an underscore because `hp-vpinns_amd` is not a legal
Python identifier.  Host side = numpy (quadrature, test-function tables, drivers);
the per-iteration hot path = hand-written HIP kernels for gfx950 behind the C-ABI in
`include/hpvpinn.h`, loaded by `hp_vpinns_amd._lib`.
"""
from .quadrature import Jacobi, DJacobi, GaussJacobiWeights, GaussLobattoJacobiWeights  # noqa: F401
from .testfcn import Test_fcn, dTest_fcn  # noqa: F401

__all__ = ["Jacobi", "DJacobi", "GaussJacobiWeights", "GaussLobattoJacobiWeights",
           "Test_fcn", "dTest_fcn"]


def __getattr__(name):
    """`from hp_vpinns_amd import VPINN2D` without importing torch / the HIP library at package import."""
    if name in ("VPINN1D", "VPINN2D", "VPINNAdvDiff"):
        from . import vpinn
        return getattr(vpinn, name)
    raise AttributeError(f"module 'hp_vpinns_amd' has no attribute {name!r}")
