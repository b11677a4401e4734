"""LQGSpec — the 12-field (generalised) LQG specification of the reference, holding torch tensors.

Mirrors lqg/spec.py:5-19 field for field (same names, same order).  Every field carries time on
axis 0 — `Q[T,b,b] q[T,b] P[T,u,b] R[T,u,u] r[T,u] A[T,b,b] B[T,b,u] V[T,b,nv] F[T,y,b] W[T,y,nw]`
(`Qf[b,b]`, `qf[b]` have no time axis) — and MAY carry one extra leading axis of B systems
(parameter candidates), which is how this package expresses what the reference writes as
`jax.vmap` over candidates (notebooks/Tutorial.ipynb cell 38).  Fields without the extra axis are shared
by all systems.  Time-invariant fields should be stride-0 `expand` views (what `time_stack_spec` builds):
the HIP library then loads them once per system instead of once per step.
"""
from typing import NamedTuple

import torch


class LQGSpec(NamedTuple):
    """ (generalized) LQG specification """

    Q: torch.Tensor
    q: torch.Tensor
    Qf: torch.Tensor
    qf: torch.Tensor
    P: torch.Tensor
    R: torch.Tensor
    r: torch.Tensor
    A: torch.Tensor
    B: torch.Tensor
    V: torch.Tensor
    F: torch.Tensor
    W: torch.Tensor
