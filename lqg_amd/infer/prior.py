"""Default priors of lqg/infer/prior.py:7-22 as plain (family, parameters) records plus a torch sampler
(NumPyro is not available; only prior sampling for parameter-recovery studies is provided)."""
import math

import torch

default_prior = {
    "action_cost": ("lognormal", -2.0, 1.0),
    "sigma_target": ("halfnormal", 50.0),
    "action_variability": ("halfnormal", 1.0),
    "signal_dep_noise": ("halfnormal", 1.0),
    "sigma_cursor": ("halfnormal", 12.5),
    "sigma": ("halfnormal", 50.0),
    "subj_noise": ("halfnormal", 1.0),
    "subj_vel_noise": ("halfnormal", 2.0),
    **{f"sigma_target_{k}": ("halfnormal", 50.0) for k in range(6)},       # per-condition noise levels, prior.py:16-21
}


def prior():
    return default_prior


def lognormal_from_quantiles(x1, x2, p1=0.05, p2=0.95):
    """(mu, sigma) of the log-normal with F(x1) = p1 and F(x2) = p2 (lqg/infer/prior.py:33-49)."""
    nd = torch.distributions.Normal(0.0, 1.0)
    z1, z2 = float(nd.icdf(torch.tensor(p1))), float(nd.icdf(torch.tensor(p2)))
    sigma = (math.log(x2) - math.log(x1)) / (z2 - z1)
    mu = (math.log(x2) * z2 - math.log(x1) * z1) / (z2 - z1)
    return mu, sigma


def sample_params(prior_dict, seed=0, n=None):
    g = torch.Generator()
    g.manual_seed(int(seed))
    shape = () if n is None else (n,)
    out = {}
    for name, rec in prior_dict.items():
        z = torch.randn(shape, generator=g, dtype=torch.float64)
        out[name] = torch.exp(rec[1] + rec[2] * z) if rec[0] == "lognormal" else rec[1] * z.abs()
    return out
