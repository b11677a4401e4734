"""value_and_grad of the summed log-likelihood with respect to the (positive) model parameters — the role of
`jax.value_and_grad(ll)` (notebooks/Tutorial.ipynb cell 42, lqg/optim.py:142-147).

method="adjoint" (default): reverse mode — the HIP adjoint sweep behind torch.autograd (lqg_amd/grad.py), chained
through the model constructor; cost independent of the number of parameters.
method="fd": central finite differences in log-space evaluated as ONE batched candidate sweep (2P+1 systems) of the
forward path; the cross-check, and the fallback for model shapes without adjoint kernels."""
import torch

from lqg_amd import options

from lqg_amd.infer.models import get_model_params, log_likelihood_objective


_graphed = {}


def _graphed_fd(x, model_type, names, fixed, process_noise, dt, fd_step, group, n_points=1):
    """Captured evaluator for (data, model class, parameter names, fixed values), built on first use; None when the
    evaluation cannot be captured (then, and with LQG_GRAPH=0, the eager path below runs)."""
    import os
    if group is not None or not x.is_cuda or not options.flag("GRAPH"):
        return None
    from lqg_amd.infer import graphed
    if graphed._sharded():
        return None
    key = (x.data_ptr(), tuple(x.shape), x.dtype, x._version, model_type, tuple(names),
           tuple(sorted((k, float(v)) for k, v in fixed.items())), float(process_noise), float(dt), float(fd_step), n_points)
    if key not in _graphed:
        if len(_graphed) >= 8:
            old = _graphed.pop(next(iter(_graphed)))[0]
            if old is not None:
                old.release()               # (a graph must not be destroyed while a replay of it is still in flight)
        x64 = x.to(torch.float64)                # (the objective is evaluated in fp64, as the eager path does)
        ev = graphed.make(graphed.GraphedFiniteDifference, x64, model_type, list(names), n_points, group=group, h=fd_step,
                          fixed=fixed, process_noise=process_noise, dt=dt)
        _graphed[key] = (ev, x, x64)             # (the data kept alive: the graph holds its address)
    return _graphed[key][0]


def value_and_grad(x, model_type, params, process_noise=1.0, dt=1.0 / 60, fd_step=1e-4, group=None, method="adjoint",
                   **fixed):
    """params: dict name -> positive float.  Returns (objective, {name: d objective / d param}) in fp64.
    x[n, T, d] follows lqg_model's convention (T rows = T-1 steps, lqg/infer/models.py:32)."""
    x_in = x
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
    ev = _graphed_fd(x_in, model_type, names, fixed, process_noise, dt, fd_step, group)
    if ev is not None:            # the whole evaluation replayed as one hipGraph (lqg_amd/infer/graphed.py)
        import math
        out = ev(torch.tensor([[math.log(float(params[k])) for k in names]], dtype=torch.float64)).cpu()
        if not bool(torch.isnan(out).any()):      # (NaN: a precondition of the frozen graph failed at these values — eager path)
            return float(out[0, 0]), {k: float(out[0, 1 + i]) / float(params[k]) for i, k in enumerate(names)}
    z = torch.log(torch.tensor([float(params[k]) for k in names], dtype=torch.float64, device=x.device))
    eye = torch.eye(P, dtype=torch.float64, device=x.device)
    Z = torch.cat([z[None], z[None] + fd_step * eye, z[None] - fd_step * eye])
    cand = {k: torch.exp(Z[:, i]) for i, k in enumerate(names)}
    obj = log_likelihood_objective(x, model_type, cand, process_noise=process_noise, dt=dt, group=group, **fixed)
    dlog = (obj[1:P + 1] - obj[P + 1:]) / (2 * fd_step)               # d obj / d log(param)
    grad = {k: float(dlog[i]) / float(params[k]) for i, k in enumerate(names)}
    return float(obj[0]), grad
