"""Temporal-delay augmentation — mirrors lqg/tracking/delay.py:9-51.

`delay_system(spec, delay)` appends a shift register of `delay` past states to the state vector (state dimension
n -> n*(1+delay)); the observation matrix reads the OLDEST copy, the cost and the control act on the current one,
and the noise matrix gains n*delay zero columns.  As in the reference the affine terms are dropped (q = P = r = 0,
Qf = Q[-1]) and R, W pass through.  Time-invariant fields stay stride-0 views (built once, re-expanded).

Kernel coverage: the HIP kernels keep one system's matrices in registers, so joint dimensions x+b up to 20 are
served (shapes outside lqg_dims.def are compiled on first use, lqg_amd/build.py:build_dims_library):
BoundedActor with delay <= 4, SubjectiveActor(dim=1) with delay <= 3.  The reference's DelayedSubjectiveActor
(delay = 12, x+b = 65) constructs here but its likelihood is rejected with LQG_ERR_DIMS — see DESIGN.md §9.
"""
import torch

from lqg_amd.spec import LQGSpec
from lqg_amd.system import System
from lqg_amd.tracking.subjective import SubjectiveActor
from lqg_amd.utils import mark_zero


def _per_step(M, fn):
    """Apply fn to the [..., r, c] matrices of a [..., T, r, c] stack; a stride-0 time axis stays stride-0."""
    T = M.shape[-3]
    if T > 1 and M.stride(-3) == 0:
        out = fn(M.select(-3, 0))
        return out.unsqueeze(-3).expand(*out.shape[:-2], T, *out.shape[-2:])
    return fn(M)


def delay_system(spec: LQGSpec, delay: int) -> LQGSpec:
    """lqg/tracking/delay.py:9-33."""
    delay = int(delay)
    if delay < 0:
        raise ValueError("delay must be >= 0")
    T, n = spec.A.shape[-3], spec.A.shape[-1]
    u, y = spec.R.shape[-1], spec.F.shape[-2]
    N, pad = n * (1 + delay), n * delay
    kw = dict(dtype=spec.A.dtype, device=spec.A.device)

    def grow(M, rows, cols, r0=0, c0=0):
        out = torch.zeros(*M.shape[:-2], rows, cols, **kw)
        out[..., r0:r0 + M.shape[-2], c0:c0 + M.shape[-1]] = M
        return out

    def aug_A(A):                                               # blockdiag(A, 0) + ones on the n-th subdiagonal
        out = grow(A, N, N)
        i = torch.arange(pad, device=A.device)
        out[..., i + n, i] += 1.0
        return out

    A = _per_step(spec.A, aug_A)
    B = _per_step(spec.B, lambda M: grow(M, N, u))
    F = _per_step(spec.F, lambda M: grow(M, y, N, 0, pad))      # observe the oldest copy
    V = _per_step(spec.V, lambda M: grow(M, N, M.shape[-1] + pad))
    Q = _per_step(spec.Q, lambda M: grow(M, N, N))
    z = torch.zeros((), **kw)
    lead = Q.shape[:-3]
    return LQGSpec(A=A, B=B, F=F, V=V, W=spec.W, Q=Q, R=spec.R, q=mark_zero(z.expand(*lead, T, N)),
                   Qf=Q.select(-3, T - 1), qf=mark_zero(z.expand(*lead, N)), P=mark_zero(z.expand(*lead, T, u, N)),
                   r=mark_zero(z.expand(*lead, T, u)))


class TemporalDelayModel(System):
    """lqg/tracking/delay.py:36-41."""

    def __init__(self, system, delay):
        super().__init__(actor=delay_system(system.actor, delay=delay),
                         dynamics=delay_system(system.dynamics, delay=delay))


class DelayedSubjectiveActor(TemporalDelayModel):
    """lqg/tracking/delay.py:44-51 (delay fixed at 12 steps, as in the reference)."""

    def __init__(self, process_noise=1., c=0.5, action_variability=0.5, subj_noise=1., subj_vel_noise=10.,
                 sigma_target=6., sigma_cursor=3., dt=1. / 60, **kw):
        system = SubjectiveActor(process_noise=process_noise, action_cost=c, action_variability=action_variability,
                                 subj_noise=subj_noise, subj_vel_noise=subj_vel_noise, sigma_target=sigma_target,
                                 sigma_cursor=sigma_cursor, dt=dt, **kw)
        super().__init__(system=system, delay=12)
