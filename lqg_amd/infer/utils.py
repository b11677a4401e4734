"""lqg/infer/utils.py: `infer` (NUTS / NeuTra through NumPyro) is out of scope (NumPyro drivers) — the
gradient it needs exists (lqg_amd/grad.py: `model.log_likelihood(x)` is differentiable by torch.autograd through the HIP
adjoint sweep), NumPyro's samplers do not; it raises instead of silently doing something else.  `sample_from_prior` is
provided."""
from lqg_amd.infer import prior
from lqg_amd.infer.models import get_model_params


def infer(x, num_samples, num_warmup, model=None, method="nuts", **kwargs):
    if method not in ("nuts", "neutra"):
        raise ValueError("Please specify a valid inference method (nuts, neutra).")      # lqg/infer/utils.py:33-34
    raise NotImplementedError(
        "lqg_amd has no MCMC driver (NumPyro is not part of this build).  d log p / d theta is available — "
        "lqg_amd.infer.value_and_grad or torch.autograd through model.log_likelihood — for an external sampler; "
        "lqg_amd.infer.max_likelihood and candidate_search are the built-in drivers.")


def sample_from_prior(model_type, seed, prior_dict=prior.default_prior, n=None):
    """Prior draw restricted to the parameters of `model_type` (lqg/infer/utils.py:44-46)."""
    params = prior.sample_params(prior_dict, seed=seed, n=n)
    return {k: v for k, v in params.items() if k in get_model_params(model_type)}
