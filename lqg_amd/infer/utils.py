"""lqg/infer/utils.py: `infer` — posterior sampling with NUTS over the priors of lqg/infer/prior.py (the reference runs
NumPyro's NUTS on `lifted_model`, utils.py:14-41) — and `sample_from_prior`.  NumPyro is not part of this build: the
sampler is lqg_amd/infer/mcmc.py (NUTS with dual averaging; all chains' gradient evaluations batched on the candidate
axis of the HIP likelihood).  method="neutra" (a BNAF normalising flow trained by SVI, utils.py:20-31) is not provided."""
from lqg_amd.infer import prior
from lqg_amd.infer.models import get_model_params


def infer(x, num_samples, num_warmup, model=None, numpyro_fn=None, process_noise=1., dt=1. / 60, method="nuts",
          progress_bar=True, num_chains=1, seed=0, grad_method="fd", **fixed):
    """lqg/infer/utils.py:14-41, same positional order.  x[n, T, d] (T rows = T-1 steps, as lqg_model); returns an object
    with get_samples(group_by_chain=False) / print_summary() / get_extra_fields() like numpyro's MCMC.
    numpyro_fn: the reference's 5th positional parameter (a NumPyro model function, default `lifted_model`).  There is no
    NumPyro here; None or this package's `lqg_model` / `lifted_model` select the built-in potential, anything else raises."""
    if numpyro_fn is not None:
        from lqg_amd.infer import models as _models
        if numpyro_fn not in (getattr(_models, "lqg_model", None), getattr(_models, "lifted_model", None)):
            raise NotImplementedError("infer(numpyro_fn=...): custom NumPyro model functions are not supported (NumPyro is not "
                                      "part of lqg_amd); pass None for the lqg_model potential")
    if method not in ("nuts", "neutra"):
        raise ValueError("Please specify a valid inference method (nuts, neutra).")      # lqg/infer/utils.py:33-34
    if method == "neutra":
        raise NotImplementedError("method='neutra' (NumPyro AutoBNAFNormal + NeuTraReparam) is not part of lqg_amd; "
                                  "method='nuts' samples the same posterior")
    from lqg_amd.infer.mcmc import infer_nuts
    from lqg_amd.tracking import BoundedActor
    del progress_bar
    return infer_nuts(x, num_samples, num_warmup, BoundedActor if model is None else model, process_noise=process_noise,
                      dt=dt, num_chains=num_chains, seed=seed, grad_method=grad_method, **fixed)


def sample_from_prior(model_type, seed, prior_dict=prior.default_prior, n=None):
    """Prior draw restricted to the parameters of `model_type` (lqg/infer/utils.py:44-46)."""
    params = prior.sample_params(prior_dict, seed=seed, n=n)
    return {k: v for k, v in params.items() if k in get_model_params(model_type)}
