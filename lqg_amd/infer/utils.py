"""lqg/infer/utils.py: `infer` (NUTS / NeuTra through NumPyro) is out of scope for this round — it needs the
gradient of the likelihood (reverse-mode adjoint sweep) and NumPyro's samplers, neither of
which exists here; it raises instead of silently doing something else.  `sample_from_prior` is provided."""
from lqg_amd.infer import prior
from lqg_amd.infer.models import get_model_params


def infer(x, num_samples, num_warmup, model=None, method="nuts", **kwargs):
    if method not in ("nuts", "neutra"):
        raise ValueError("Please specify a valid inference method (nuts, neutra).")      # lqg/infer/utils.py:33-34
    raise NotImplementedError(
        "lqg_amd has no MCMC driver: NUTS needs d log p / d theta (adjoint sweep, planned) and NumPyro. "
        "Use lqg_amd.infer.max_likelihood (finite-difference Adam) or candidate_search.")


def sample_from_prior(model_type, seed, prior_dict=prior.default_prior, n=None):
    """Prior draw restricted to the parameters of `model_type` (lqg/infer/utils.py:44-46)."""
    params = prior.sample_params(prior_dict, seed=seed, n=n)
    return {k: v for k, v in params.items() if k in get_model_params(model_type)}
