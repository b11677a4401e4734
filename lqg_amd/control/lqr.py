"""lqg.control.lqr — mirrors lqg/control/lqr.py:8-42 (Gains, backward) on the HIP library."""
from typing import NamedTuple

import torch

from lqg_amd import _hip
from lqg_amd.spec import LQGSpec


class Gains(NamedTuple):
    """LQR control gains"""

    L: torch.Tensor
    l: torch.Tensor
    H: torch.Tensor = None


def backward(spec: LQGSpec, eps: float = 1e-8) -> Gains:
    """Finite-horizon Riccati recursion (lqg/control/lqr.py:16-42).

    Returns Gains(L[T,u,b], l[T,u], H[T,u,u]) in forward time order, H being the regularised Ht
    (with a leading [B] axis when the spec carries one)."""
    L, l, H = _hip.riccati_backward(spec, eps)
    return Gains(L=L, l=l, H=H)
