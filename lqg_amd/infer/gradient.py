"""value_and_grad of the summed log-likelihood with respect to the (positive) model parameters, by central finite
differences in log-space evaluated as ONE batched candidate sweep (2P+1 systems) — the stand-in for
`jax.value_and_grad(ll)` (notebooks/Tutorial.ipynb cell 42, lqg/optim.py:142-147) until an adjoint sweep exists."""
import torch

from lqg_amd.infer.models import get_model_params, log_likelihood_objective


def value_and_grad(x, model_type, params, process_noise=1.0, dt=1.0 / 60, fd_step=1e-4, group=None, **fixed):
    """params: dict name -> positive float.  Returns (objective, {name: d objective / d param}) in fp64."""
    x = x.to(torch.float64)
    names = list(params)
    P = len(names)
    z = torch.log(torch.tensor([float(params[k]) for k in names], dtype=torch.float64, device=x.device))
    eye = torch.eye(P, dtype=torch.float64, device=x.device)
    Z = torch.cat([z[None], z[None] + fd_step * eye, z[None] - fd_step * eye])
    cand = {k: torch.exp(Z[:, i]) for i, k in enumerate(names)}
    obj = log_likelihood_objective(x, model_type, cand, process_noise=process_noise, dt=dt, group=group, **fixed)
    dlog = (obj[1:P + 1] - obj[P + 1:]) / (2 * fd_step)               # d obj / d log(param)
    grad = {k: float(dlog[i]) / float(params[k]) for i, k in enumerate(names)}
    return float(obj[0]), grad
