"""lqg_amd.infer: host logic on CPU; likelihood-based fitting on the GPU (marked)."""
import numpy as np
import pytest
import torch

import lqg_amd
from lqg_amd.infer import candidate_search, get_model_params, infer, max_likelihood, sample_from_prior
from lqg_amd.infer.prior import lognormal_from_quantiles


def test_get_model_params_matches_reference_reflection():
    """lqg/infer/models.py:9-17: every ctor kwarg except self/dim/dt/T/process_noise/delay/covar."""
    assert get_model_params(lqg_amd.BoundedActor) == dict(action_variability=0.5, sigma_target=6.0, sigma_cursor=6.0,
                                                          action_cost=1.0)
    assert list(get_model_params(lqg_amd.SubjectiveActor)) == ["action_cost", "action_variability", "subj_noise",
                                                               "subj_vel_noise", "sigma_target", "sigma_cursor"]
    assert "sigma" in get_model_params(lqg_amd.RelativeObservationBoundedActor)
    assert set(get_model_params(lqg_amd.PointMassBoundedActor)) >= {"damping", "m", "tau", "action_cost"}


def test_infer_rejects_unknown_method_like_the_reference():
    with pytest.raises(ValueError, match="valid inference method"):      # lqg/infer/utils.py:33-34
        infer(None, 10, 10, method="hmc")
    with pytest.raises(NotImplementedError):
        infer(None, 10, 10, method="nuts")


def test_prior_helpers():
    p = sample_from_prior(lqg_amd.BoundedActor, seed=1)
    assert set(p) == {"action_cost", "sigma_target", "action_variability", "sigma_cursor"}
    assert all(float(v) > 0 for v in p.values())
    mu, sigma = lognormal_from_quantiles(1.0, 10.0)
    nd = torch.distributions.Normal(0.0, 1.0)
    assert abs(float(nd.cdf(torch.tensor((np.log(1.0) - mu) / sigma))) - 0.05) < 1e-6
    assert abs(float(nd.cdf(torch.tensor((np.log(10.0) - mu) / sigma))) - 0.95) < 1e-6


@pytest.mark.gpu
def test_candidate_search_recovers_sigma_target():
    """The reference's tutorial sweep `vmap(ll)(sigmas)` (notebooks/Tutorial.ipynb cell 38)."""
    true = dict(sigma_target=25.0, action_variability=0.5, action_cost=0.05, sigma_cursor=1.0)
    m = lqg_amd.BoundedActor(T=400, device="cuda", dtype=torch.float64, **true)
    x = m.simulate(0, n=50)                                   # [50, 401, 2]
    sigmas = torch.linspace(5.0, 50.0, 64, device="cuda", dtype=torch.float64)
    fixed = {k: v for k, v in true.items() if k != "sigma_target"}
    obj, best = candidate_search(x, lqg_amd.BoundedActor, dict(sigma_target=sigmas), **fixed)
    assert obj.shape == (64,) and obj.dtype == torch.float64
    assert abs(float(sigmas[best]) - 25.0) < 6.0
    # the objective is the per-candidate sum of System.log_likelihood
    one = lqg_amd.BoundedActor(T=400, device="cuda", dtype=torch.float64, **{**true, "sigma_target": float(sigmas[10])})
    assert abs(float(one.log_likelihood(x).sum()) - float(obj[10])) < 1e-8 * abs(float(obj[10]))


@pytest.mark.gpu
def test_max_likelihood_improves_and_approaches_truth():
    true = dict(sigma_target=12.0, action_variability=0.4, action_cost=0.3, sigma_cursor=2.0)
    m = lqg_amd.BoundedActor(T=300, device="cuda", dtype=torch.float64, **true)
    x = m.simulate(3, n=40)
    params, losses = max_likelihood(x, lqg_amd.BoundedActor, steps=150, step_size=0.05, action_cost=0.3,
                                    sigma_cursor=2.0)
    assert set(params) == {"action_variability", "sigma_target"}          # the fixed ones are not fitted
    assert losses.shape == (150,) and float(losses[-1]) < float(losses[0])
    nll_true = -float(m.log_likelihood(x).sum())
    assert float(losses[-1]) < nll_true + 5.0                              # at least as good as the truth (+slack)
    assert abs(params["sigma_target"] - 12.0) < 4.0 and abs(params["action_variability"] - 0.4) < 0.15


@pytest.mark.gpu
def test_value_and_grad_matches_oracle_finite_differences(oracle_lib):
    """The batched finite-difference gradient against an independent central difference of the fp64 CPU oracle
    (the reference's `grad(ll)(28.)`, Tutorial cell 42, differentiates the same function)."""
    import lqg_np as O
    from lqg_amd.infer import value_and_grad
    true = dict(sigma_target=25.0, action_variability=0.5, action_cost=0.05, sigma_cursor=1.0)
    m = lqg_amd.BoundedActor(T=120, device="cuda", dtype=torch.float64, **true)
    x = m.simulate(7, n=8)
    fixed = dict(action_variability=0.5, action_cost=0.05, sigma_cursor=1.0)
    val, grad = value_and_grad(x, lqg_amd.BoundedActor, dict(sigma_target=28.0), **fixed)

    def oracle_obj(sig):
        A, B = np.eye(2), (1.0 / 60) * np.array([[0.0], [1.0]])
        spec = O.time_stack_spec(A, B, np.eye(2), np.diag([1.0, 0.5]), np.diag([sig, 1.0]),
                                 np.array([[1.0, -1.0], [-1.0, 1.0]]), np.eye(1) * 0.05, 120)
        return float(oracle_lib.log_likelihood(spec, spec, x.cpu().numpy()).sum())

    h = 1e-4
    ref_grad = (oracle_obj(28.0 + h) - oracle_obj(28.0 - h)) / (2 * h)
    assert abs(val / oracle_obj(28.0) - 1) < 1e-11
    assert abs(grad["sigma_target"] / ref_grad - 1) < 1e-5
