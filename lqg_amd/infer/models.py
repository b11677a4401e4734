"""Model plumbing of lqg/infer/models.py without NumPyro: which constructor arguments are inferred
(`get_model_params`, models.py:9-17) and the objective `lqg_model` hands to NumPyro (models.py:20-34):
the summed log-likelihood of data x[n, T, d] under `model_type(T=T-1, **params)`."""
import inspect


_NOT_INFERRED = ("self", "dim", "dt", "T", "process_noise", "delay", "covar", "device", "dtype")


_model_params_cache = {}


def get_model_params(model_class):
    """Constructor arguments that are model parameters, with their defaults (lqg/infer/models.py:9-17).
    (Cached per class: the signature walk costs ~40 us, once per objective evaluation in an optimiser loop.)"""
    if model_class not in _model_params_cache:
        init_signature = inspect.signature(model_class.__init__)
        # (*args / **kw catch-alls are not parameters: the reference's loop would list `kw` with an empty default and then
        # fail to sample it; DelayedSubjectiveActor forwards device / dtype / T through **kw here)
        _model_params_cache[model_class] = {name: param.default for name, param in init_signature.parameters.items()
                                            if name not in _NOT_INFERRED
                                            and param.kind not in (param.VAR_POSITIONAL, param.VAR_KEYWORD)}
    return dict(_model_params_cache[model_class])


def log_likelihood_objective(x, model_type, params, process_noise=1.0, dt=1.0 / 60, group=None, **fixed_params):
    """sum_n log p(x_n | theta_c) for every candidate c (fp64 [C], or a scalar for scalar params).

    x[n, T, d] follows the reference's convention: T rows = T-1 steps (`T=T - 1`, models.py:32).
    `params` maps parameter name -> scalar or [C] tensor of candidates; `fixed_params` pins the others
    (models.py:25-26); anything left takes the constructor default.  With an initialised process group the trials in
    x are this rank's shard and the partial sums are all-reduced (lqg_amd.dist)."""
    from lqg_amd import dist as ld

    n, T, d = x.shape
    kw = dict(get_model_params(model_type))
    kw.update(fixed_params)
    kw.update(params)
    model = model_type(process_noise=process_noise, dt=dt, T=T - 1, device=x.device, dtype=x.dtype, **kw)
    return ld.log_likelihood_sum(model, x, group=group)


def lqg_model(x, model_type, process_noise=1.0, dt=1.0 / 60, **fixed_params):
    """The name and signature of lqg/infer/models.py:20-34.  In the reference this is a NumPyro model function (it
    registers `numpyro.param`s and one `numpyro.sample("x", ..., obs=x)` site) handed to `max_likelihood` / `infer` as their
    third / fifth positional argument.  There is no NumPyro param store here: called directly it returns what that sample
    site contributes at the initial point — sum_n log p(x_n | constructor defaults, fixed_params), fp64 — and the drivers of
    this package recognise the function itself as "the built-in objective" (`max_likelihood(x, Model, lqg_model, 1.0)`)."""
    return log_likelihood_objective(x, model_type, {}, process_noise=process_noise, dt=dt, **fixed_params)


def lifted_model(x, model_type, process_noise=1.0, dt=1.0 / 60, **fixed_params):
    """What lqg/infer/utils.py:9 imports as the default model of `infer` (parameters lifted to their priors,
    lqg/infer/prior.py): the log joint at the initial point — `lqg_model` plus the log prior of the non-fixed parameters at
    their constructor defaults.  `infer` samples this posterior (lqg_amd/infer/mcmc.py: Potential)."""
    import torch
    from lqg_amd.infer import prior as _prior
    from lqg_amd.infer.mcmc import log_prior
    lp = sum(float(log_prior(k, torch.tensor(float(v), dtype=torch.float64), _prior.default_prior))
             for k, v in get_model_params(model_type).items() if k not in fixed_params and k in _prior.default_prior)
    return lqg_model(x, model_type, process_noise, dt, **fixed_params) + lp


# ---------------------------------------------------------------------------------------------------------------------
# (Nc, N, T, d) data: several CONDITIONS, each with its own trials, some parameters shared across conditions and some
# per condition — lqg/infer/models.py:37-61 (`common_lqg_model`: everything shared except sigma_target) and :67-130
# (`shared_params_lqg_model`: `shared_params` names the shared ones), what cpp_data_fit.py:15-55 fits to the
# 6 x 20 x 1068 x 2 tracking data of lqg/io.py.  Those two reference models construct `model_type(T=T)` for data with T
# rows (the scan then has mismatched lengths); this follows lqg_model's convention, T rows = T-1 steps.

def split_params(model_type, shared_params=None, **fixed_params):
    """(shared names, per-condition names) among the inferred parameters of `model_type` (models.py:98-107)."""
    names = [k for k in get_model_params(model_type) if k not in fixed_params]
    shared = [k for k in names if k in set(shared_params or ())]
    return shared, [k for k in names if k not in shared]


def shared_params_objective(x, model_type, params, shared_params=None, process_noise=1.0, dt=1.0 / 60, dim=1, group=None,
                            per_condition=False, _log_likelihood=None, **fixed_params):
    """sum_k sum_n log p(x[k, n] | theta_shared, theta_k) — the likelihood part of shared_params_lqg_model's potential.

    x[Nc, N, T, d]: conditions x trials x rows x observed dims (this rank's shard of the trial axis when `group` is an
    initialised process group; the partial sums are all-reduced once).  `params`: name -> value; a SHARED parameter is a
    scalar or a [C] tensor of candidates, a PER-CONDITION parameter is [Nc] or [C, Nc]; missing ones take
    `fixed_params`, then the constructor default.  Returns fp64 [C] (or a scalar when nothing carries a candidate
    axis); per_condition=True returns the [C, Nc] (or [Nc]) table instead of its sum.

    One parameter vector: the Nc conditions are the SYSTEM axis of ONE launch, condition k scoring its own trials x[k]
    (x is handed over as per-system data, no copy).  C candidates: one launch of C systems per condition, the
    condition's trials shared by all of them (system stride 0, no copy).  Differentiable: parameters that require grad
    get their gradient through the HIP adjoint sweep (lqg_amd/grad.py)."""
    import inspect

    import torch

    from lqg_amd import _hip
    from lqg_amd import dist as ld

    Nc, N, T, d = x.shape
    shared, _ = split_params(model_type, shared_params)
    base = dict(get_model_params(model_type))
    base.update(fixed_params)
    C, conv = None, {}
    for name, v in params.items():
        v = torch.as_tensor(v, dtype=x.dtype, device=x.device)
        if name in shared:
            if v.dim() > 1:
                raise ValueError(f"shared parameter {name}: scalar or [C], got {tuple(v.shape)}")
            cand = v.dim() == 1
        else:
            if v.dim() == 0:
                v = v.expand(Nc)
            if v.dim() > 2 or v.shape[-1] != Nc:
                raise ValueError(f"per-condition parameter {name}: [Nc={Nc}] or [C, Nc], got {tuple(v.shape)}")
            cand = v.dim() == 2
        if cand:
            if C is not None and v.shape[0] != C:
                raise ValueError(f"parameter {name} has {v.shape[0]} candidates, others have {C}")
            C = v.shape[0]
        conv[name] = v
    ctor = dict(process_noise=process_noise, dt=dt, T=T - 1, device=x.device, dtype=x.dtype)
    if "dim" in inspect.signature(model_type.__init__).parameters:
        ctor["dim"] = dim

    loglik = _log_likelihood or (lambda model, data: model.log_likelihood(data))   # (CPU tests inject the oracle here)

    def trial_sum(ll):                                    # [B, N] -> fp64 [B]
        if ll.requires_grad or not ll.is_cuda:
            return ll.double().sum(-1)
        return _hip.sum_trials(ll)

    if C is None:                                         # conditions on the system axis
        kw = dict(base)
        for name, v in conv.items():
            kw[name] = v.expand(Nc) if name in shared else v
        table = trial_sum(loglik(model_type(**ctor, **kw), x))                              # [Nc]
    else:                                                 # candidates on the system axis, one launch per condition
        cols = []
        for k in range(Nc):
            kw = dict(base)
            for name, v in conv.items():
                if name in shared:
                    kw[name] = v if v.dim() == 1 else v.expand(C)
                else:
                    kw[name] = v[:, k] if v.dim() == 2 else v[k].expand(C)
            cols.append(trial_sum(loglik(model_type(**ctor, **kw), x[k])))                  # [C]
        table = torch.stack(cols, dim=-1)                                                   # [C, Nc]
    table = _all_reduce_autograd(table, group) if table.requires_grad else ld.all_reduce_sum(table, group=group)
    return table if per_condition else table.sum(-1)


def _all_reduce_autograd(t, group):
    """All-reduce of a differentiable per-rank partial sum: the gradient of the total w.r.t. this rank's term is 1."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t

    class _AR(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v):
            v = v.clone()
            dist.all_reduce(v, group=group)
            return v

        @staticmethod
        def backward(ctx, g):
            return g
    return _AR.apply(t)


def common_objective(x, model_type, params, **kw):
    """common_lqg_model (models.py:37-61): every parameter shared across conditions except sigma_target."""
    shared = [k for k in get_model_params(model_type) if k != "sigma_target"]
    return shared_params_objective(x, model_type, params, shared_params=shared, **kw)
