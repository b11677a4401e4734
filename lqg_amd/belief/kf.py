"""lqg.belief.kf — mirrors lqg/belief/kf.py:6-21 (forward) on the HIP library."""
import torch

from lqg_amd import _hip
from lqg_amd.spec import LQGSpec


def forward(spec: LQGSpec, Sigma0: torch.Tensor) -> torch.Tensor:
    """Kalman gain recursion K[T,b,y] from the initial belief covariance Sigma0[b,b] (lqg/belief/kf.py:6-21)."""
    return _hip.kalman_forward(spec, Sigma0)
