"""Model plumbing of lqg/infer/models.py without NumPyro: which constructor arguments are inferred
(`get_model_params`, models.py:9-17) and the objective `lqg_model` hands to NumPyro (models.py:20-34):
the summed log-likelihood of data x[n, T, d] under `model_type(T=T-1, **params)`."""
import inspect


_NOT_INFERRED = ("self", "dim", "dt", "T", "process_noise", "delay", "covar", "device", "dtype")


def get_model_params(model_class):
    """Constructor arguments that are model parameters, with their defaults (lqg/infer/models.py:9-17)."""
    init_signature = inspect.signature(model_class.__init__)
    parameters = {}
    for name, param in init_signature.parameters.items():
        if name not in _NOT_INFERRED:
            parameters[param.name] = param.default
    return parameters


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
