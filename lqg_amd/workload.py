"""Synthetic workloads of BASELINE.json / SURVEY.md §8(d): seeded parameter candidates drawn log-uniformly
around the reference's defaults, one LQG system per candidate, trajectories simulated FROM the model itself
(so log-likelihoods are O(-T d) and well conditioned — not white noise).  Used by bench.py, smoke() and the
full-size property tests; all arithmetic runs through the HIP library."""
import math

import torch

from lqg_amd.spec import LQGSpec
from lqg_amd.system import System
from lqg_amd.tracking import BoundedActor, SubjectiveActor

# ranges: inside the tutorial slider ranges (notebooks/Tutorial.ipynb:715-724) and the prior scales of
# lqg/infer/prior.py:7-15
RANGES = dict(action_variability=(0.1, 2.0), sigma_target=(1.0, 50.0), sigma_cursor=(1.0, 15.0),
              action_cost=(0.01, 10.0), subj_noise=(0.5, 2.0), subj_vel_noise=(0.1, 2.0))


def log_uniform(n, lo, hi, gen, device, dtype):
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    return torch.exp(math.log(lo) + u * (math.log(hi) - math.log(lo))).to(dtype)


def sample_params(names, n, seed, device, dtype):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return {k: log_uniform(n, *RANGES[k], gen, device, dtype) for k in names}


def headline_system(B, T, seed, device, dtype):
    """BASELINE headline / config 5: SubjectiveActor(dim=2) — x=4, b=6 (n=6), u=2, y=4, d=4."""
    p = sample_params(("action_variability", "sigma_target", "sigma_cursor", "action_cost", "subj_noise",
                       "subj_vel_noise"), B, seed, device, dtype)
    return SubjectiveActor(dim=2, T=T, process_noise=1.0, dt=1.0 / 60, device=device, dtype=dtype, **p), p


def bounded_system(B, T, seed, device, dtype):
    """configs 1 / 3: BoundedActor — x=b=2, u=1, y=2, d=2."""
    p = sample_params(("action_variability", "sigma_target", "sigma_cursor", "action_cost"), B, seed, device, dtype)
    return BoundedActor(T=T, process_noise=1.0, dt=1.0 / 60, device=device, dtype=dtype, **p), p


def slice_system(system, lo, hi):
    """Systems lo:hi of a batched System (views, no copies)."""
    def cut(spec):
        return LQGSpec(**{f: (getattr(spec, f)[lo:hi] if getattr(spec, f).dim() == _batched_ndim(f) else getattr(spec, f))
                          for f in LQGSpec._fields})
    a = cut(system.actor)
    d = a if system.actor is system.dynamics else cut(system.dynamics)
    return System(actor=a, dynamics=d)


def _batched_ndim(f):
    base = 1 if f in ("q", "qf", "r") else 2
    return base + (0 if f in ("Qf", "qf") else 1) + 1


def simulate_one_trial_each(system, seed, d=None, chunk=1 << 15):
    """One simulated trajectory per system: x[B, 1, T+1, d] in the reference's layout."""
    B = system.n_systems
    d = system.xdim if d is None else d
    out = torch.empty((B, 1, system.T + 1, d), dtype=system.actor.A.dtype, device=system.actor.A.device)
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        xs = slice_system(system, lo, hi).simulate(seed + lo, n=1)
        out[lo:hi] = xs[..., :d]
        del xs
    return out


def pack_trials(x):
    """Re-lay trajectories x[(B,) n, T+1, d] so that the batch index is the fastest-varying one in HBM
    ([T+1][d][B*n] storage) while keeping the logical shape: a wave's 64 lanes (one system or trial each)
    then read 64 consecutive elements per (t, component).  Returns a strided VIEW with the original shape —
    every API of this package accepts it unchanged (the C ABI takes explicit strides)."""
    if x.dim() == 3:
        return x.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    return x.permute(2, 3, 0, 1).contiguous().permute(2, 3, 0, 1)


def pack_systems(t):
    """Re-lay a batched time-varying spec field t[B, T, r, c] so that the SYSTEM index is the fastest-varying one in HBM
    ([T][r][c][B] storage) while keeping the logical shape — the layout the lane-per-system sweeps want for arrays they
    read at every step (one wave-load = 256 contiguous bytes instead of 64 cache lines; mode M2 runs 1.6x faster on it,
    DESIGN.md 6b).  Note that `a.permute(...) * b` does NOT produce it: an elementwise result takes the memory order of its
    permuted operand.  Returns a strided VIEW; every API of this package accepts it unchanged."""
    if t.dim() != 4:
        raise ValueError(f"pack_systems expects [B, T, r, c], got {tuple(t.shape)}")
    return t.permute(1, 2, 3, 0).contiguous().permute(3, 0, 1, 2)
