"""Maximum likelihood — the role of lqg/infer/mle.py:14-25 (NumPyro SVI with an empty guide = Adam on the
log-likelihood of positive-constrained parameters).

The reference differentiates the likelihood with JAX reverse mode.  Two gradient methods are provided:
method="fd" (default): CENTRAL FINITE DIFFERENCES in the unconstrained (log) space evaluated as ONE batched candidate
sweep of 2P+1 systems through the fused HIP path — for ONE parameter vector this is the faster of the two on MI355X
(all 2P+1 candidates x trials run in parallel; the whole evaluation is one hipGraph replay, lqg_amd/infer/graphed.py:
0.24 ms per evaluation at T=500, 50 trials, P=4, and the Adam loop never synchronises with the host);
method="adjoint": the reverse-mode HIP sweep behind torch.autograd (lqg_amd/grad.py; 2.8 ms per evaluation on the same
workload — one lane walks one trial's 500 steps four times — but exact, and P-independent / 3x cheaper per gradient when
many parameter vectors are differentiated at once).  fp64 is used throughout.  `candidate_search` is the
derivative-free population evaluation of BASELINE config 3 (thousands of candidates x shared trials).
"""
import torch

from lqg_amd.infer.models import get_model_params, log_likelihood_objective
from lqg_amd.tracking import BoundedActor

GUARD_EVERY = 64        # steps between two read-backs of the graphed loop's NaN guard (max_likelihood)


def candidate_search(x, model=BoundedActor, candidates=None, process_noise=1.0, dt=1.0 / 60, group=None, **fixed):
    """Objective (summed log-likelihood, fp64) of every candidate: `candidates` maps parameter -> [C] tensor.
    Returns (objective[C], index of the best candidate)."""
    obj = log_likelihood_objective(x, model, candidates, process_noise=process_noise, dt=dt, group=group, **fixed)
    return obj, int(torch.argmax(obj))


def max_likelihood(x, model=BoundedActor, numpyro_fn=None, process_noise=1.0, dt=1.0 / 60, steps=2_000, step_size=0.01,
                   fd_step=1e-4, group=None, method="fd", **fixed):
    """Adam on the negative log-likelihood of the positive parameters of `model` (defaults as in the reference:
    2000 steps, step size 0.01, initial values = constructor defaults).  Returns (params, losses) like
    `svi.run` in lqg/infer/mle.py:23-25: params = dict name -> fitted value, losses[steps] = -log p(x | params).
    Positional order as lqg/infer/mle.py:14 (x, model, numpyro_fn, process_noise, dt, steps, step_size).  numpyro_fn: the
    reference's NumPyro model function (default `lqg_model`); there is no NumPyro here — None or this package's `lqg_model`
    select the built-in objective, anything else raises (the rule of `infer`)."""
    if numpyro_fn is not None:
        from lqg_amd.infer import models as _models
        if numpyro_fn is not getattr(_models, "lqg_model", None):
            raise NotImplementedError("max_likelihood(numpyro_fn=...): custom NumPyro model functions are not supported "
                                      "(NumPyro is not part of lqg_amd); pass None for the lqg_model objective")
    x = x.to(torch.float64)
    names = [k for k in get_model_params(model) if k not in fixed]
    P = len(names)
    z = torch.log(torch.tensor([float(get_model_params(model)[k]) for k in names], dtype=torch.float64, device=x.device))
    m1, m2 = torch.zeros_like(z), torch.zeros_like(z)
    b1, b2, eps = 0.9, 0.999, 1e-8
    eye = torch.eye(P, dtype=torch.float64, device=x.device)
    losses = torch.empty(steps, dtype=torch.float64, device=x.device)     # (kept on the device: no synchronisation per step)
    if method not in ("fd", "adjoint"):
        raise ValueError(f"method must be 'fd' or 'adjoint', got {method!r}")
    ev = None
    if method == "fd":            # the whole evaluation as one hipGraph replay when it can be captured (infer/graphed.py)
        from lqg_amd.infer.gradient import _graphed_fd
        ev = _graphed_fd(x, model, names, fixed, process_noise, dt, fd_step, group)
    # A precondition of the frozen graph (infer/graphed.py: eigenvalue floor inactive, conditioning) may fail somewhere along
    # the path: the replay then returns NaN.  The loop never synchronises per step; every GUARD_EVERY steps (one read-back) it
    # checks the losses since the last verified state and, on a NaN, resumes from THAT state on the eager path, which
    # re-decides per evaluation (round 3 refitted all `steps` iterations from scratch).
    it, verified = 0, (0, z, m1, m2)
    while it < steps:
        if method == "adjoint":
            from lqg_amd.infer.gradient import value_and_grad
            theta = {k: float(torch.exp(z[i])) for i, k in enumerate(names)}
            val, g = value_and_grad(x, model, theta, process_noise=process_noise, dt=dt, group=group, **fixed)
            loss = torch.tensor(-val, dtype=torch.float64, device=x.device)
            grad = -torch.tensor([g[k] * theta[k] for k in names], dtype=torch.float64, device=x.device)   # d/d log
        elif ev is not None:
            out = ev(z[None])                                 # [1, 1 + P]: objective, d objective / d z
            loss, grad = -out[0, 0].clone(), -out[0, 1:].clone()
        else:
            # one sweep over 2P+1 candidates: z, z + h e_i, z - h e_i
            Z = torch.cat([z[None], z[None] + fd_step * eye, z[None] - fd_step * eye])
            cand = {k: torch.exp(Z[:, i]) for i, k in enumerate(names)}
            obj = log_likelihood_objective(x, model, cand, process_noise=process_noise, dt=dt, group=group, **fixed)
            loss = -obj[0]
            grad = -(obj[1:P + 1] - obj[P + 1:]) / (2 * fd_step)
        losses[it] = loss
        m1 = b1 * m1 + (1 - b1) * grad
        m2 = b2 * m2 + (1 - b2) * grad * grad
        z = z - step_size * (m1 / (1 - b1 ** (it + 1))) / (torch.sqrt(m2 / (1 - b2 ** (it + 1))) + eps)
        it += 1
        if ev is not None and (it % GUARD_EVERY == 0 or it == steps):
            if bool(torch.isnan(losses[verified[0]:it]).any()):
                it, z, m1, m2 = verified                      # back to the last state no replay had poisoned
                ev = None                                     # ... and on from there on the eager path
            else:
                verified = (it, z, m1, m2)
    params = {k: float(v) for k, v in zip(names, torch.exp(z).cpu())}
    return params, losses.cpu()
