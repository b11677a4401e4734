"""lqg_amd.optim.minimize (role of lqg/optim.py:14-169): scipy driver with torch.autograd gradients."""
import numpy as np
import pytest
import torch

from lqg_amd.optim import minimize


def test_minimize_recovers_the_minimum_of_a_tree_function():
    A = torch.tensor([[3.0, 0.5], [0.5, 2.0]], dtype=torch.float64)
    target = dict(w=torch.tensor([1.0, -2.0], dtype=torch.float64), b=torch.tensor(0.7, dtype=torch.float64))
    seen = []

    def fun(p, scale):
        d = p["w"] - target["w"]
        return scale * (d @ A @ d) + (p["b"] - target["b"]) ** 2 + (p["extra"][0] - 3.0) ** 2

    x0 = dict(w=torch.zeros(2, dtype=torch.float64), b=0.0, extra=[torch.tensor(1.0, dtype=torch.float64)])
    res = minimize(fun, x0, method="L-BFGS-B", args=(2.0,), callback=lambda p, *a: seen.append(float(p["b"])))
    assert res.success and res.fun < 1e-10
    assert torch.allclose(res.x["w"], target["w"], atol=1e-5) and abs(float(res.x["b"]) - 0.7) < 1e-5
    assert abs(float(res.x["extra"][0]) - 3.0) < 1e-5 and isinstance(res.x["extra"], list)
    assert len(seen) >= 1


def test_minimize_respects_bounds_and_plain_tensors():
    res = minimize(lambda x: ((x - 2.0) ** 2).sum(), torch.zeros(3, dtype=torch.float64), method="L-BFGS-B",
                   bounds=[(None, 1.0)] * 3)
    assert np.allclose(res.x.numpy(), 1.0)


@pytest.mark.gpu
def test_minimize_fits_a_model_through_the_adjoint_sweep():
    """The reference's use: minimise the negative log-likelihood over (log) parameters, gradients by reverse mode."""
    import lqg_amd
    true = dict(sigma_target=12.0, action_cost=0.1)
    kw = dict(T=200, sigma_cursor=2.0, action_variability=0.4, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        x = lqg_amd.BoundedActor(**true, **kw).simulate(21, n=40)

    def nll(z):
        return -lqg_amd.BoundedActor(sigma_target=torch.exp(z["s"]), action_cost=torch.exp(z["c"]), **kw).log_likelihood(x).sum()

    z0 = dict(s=torch.tensor(np.log(6.0), device="cuda"), c=torch.tensor(np.log(0.5), device="cuda"))
    res = minimize(nll, z0, method="L-BFGS-B", options=dict(maxiter=60))
    with torch.no_grad():
        at_truth = float(nll(dict(s=torch.tensor(np.log(12.0), device="cuda", dtype=torch.float64),
                                  c=torch.tensor(np.log(0.1), device="cuda", dtype=torch.float64))))
    assert res.fun <= at_truth + 1e-6                              # at least as good as the generating parameters
    assert abs(float(torch.exp(res.x["s"])) - 12.0) < 4.0 and np.linalg.norm(res.jac) < 1e-2 * abs(res.fun)
