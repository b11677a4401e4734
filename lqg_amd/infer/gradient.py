"""value_and_grad of the summed log-likelihood with respect to the (positive) model parameters — the role of
`jax.value_and_grad(ll)` (notebooks/Tutorial.ipynb cell 42, lqg/optim.py:142-147).

method="adjoint" (default): reverse mode — the HIP adjoint sweep behind torch.autograd (lqg_amd/grad.py), chained
through the model constructor; cost independent of the number of parameters.
method="fd": central finite differences in log-space evaluated as ONE batched candidate sweep (2P+1 systems) of the
forward path; the cross-check, and the fallback for model shapes without adjoint kernels."""
import torch

from lqg_amd.infer.models import get_model_params, log_likelihood_objective


def value_and_grad(x, model_type, params, process_noise=1.0, dt=1.0 / 60, fd_step=1e-4, group=None, method="adjoint",
                   **fixed):
    """params: dict name -> positive float.  Returns (objective, {name: d objective / d param}) in fp64.
    x[n, T, d] follows lqg_model's convention (T rows = T-1 steps, lqg/infer/models.py:32)."""
    x = x.to(torch.float64)
    names = list(params)
    if method == "adjoint":
        from lqg_amd import dist as ld
        theta = {k: torch.tensor(float(params[k]), dtype=torch.float64, device=x.device, requires_grad=True) for k in names}
        kw = dict(get_model_params(model_type))
        kw.update(fixed)
        kw.update(theta)
        model = model_type(process_noise=process_noise, dt=dt, T=x.shape[-2] - 1, device=x.device, dtype=x.dtype, **kw)
        obj = model.log_likelihood(x).sum()
        obj.backward()
        vec = torch.stack([obj.detach()] + [theta[k].grad for k in names])
        vec = ld.all_reduce_sum(vec, group=group)                      # trials sharded over ranks: one all-reduce
        return float(vec[0]), {k: float(vec[1 + i]) for i, k in enumerate(names)}
    if method != "fd":
        raise ValueError(f"method must be 'adjoint' or 'fd', got {method!r}")
    P = len(names)
    z = torch.log(torch.tensor([float(params[k]) for k in names], dtype=torch.float64, device=x.device))
    eye = torch.eye(P, dtype=torch.float64, device=x.device)
    Z = torch.cat([z[None], z[None] + fd_step * eye, z[None] - fd_step * eye])
    cand = {k: torch.exp(Z[:, i]) for i, k in enumerate(names)}
    obj = log_likelihood_objective(x, model_type, cand, process_noise=process_noise, dt=dt, group=group, **fixed)
    dlog = (obj[1:P + 1] - obj[P + 1:]) / (2 * fd_step)               # d obj / d log(param)
    grad = {k: float(dlog[i]) / float(params[k]) for i, k in enumerate(names)}
    return float(obj[0]), grad
