#!/usr/bin/env python3
"""The flow of the reference's notebooks/Tutorial.ipynb on MI355X with `lqg_amd` as the drop-in:
build an LQG from matrices (cell 14) -> simulate trials -> likelihood sweep over sigma_target, the reference's
`vmap(ll)(sigmas)` (cell 38) -> its gradient `grad(ll)(28.)` (cell 42: reverse-mode HIP sweep via torch.autograd) ->
maximum likelihood (lqg/infer/mle.py) -> belief tracking (cell 52).    usage: python examples/tutorial.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import lqg_amd as lqg
from lqg_amd.infer import candidate_search, max_likelihood, value_and_grad

dev, dt_ = "cuda", torch.float64
dt = 1.0 / 60.0
true_params = dict(sigma_target=25.0, action_variability=0.5, action_cost=0.05, sigma_cursor=1.0)

# --- cell 14: an LQG from its matrices
A = torch.eye(2, device=dev, dtype=dt_)
B = torch.tensor([[0.0], [dt]], device=dev, dtype=dt_)
V = torch.diag(torch.tensor([1.0, 0.5], device=dev, dtype=dt_))
C = torch.eye(2, device=dev, dtype=dt_)
W = torch.diag(torch.tensor([25.0, 1.0], device=dev, dtype=dt_))
Q = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], device=dev, dtype=dt_)
R = torch.eye(1, device=dev, dtype=dt_) * 0.05
T = 500
model = lqg.LQG(A, B, C, V, W, Q, R, T=T)
x = model.simulate(0, n=50)
print("simulate:", tuple(x.shape), "log_likelihood:", tuple(model.log_likelihood(x).shape))

# --- cell 38: likelihood of one parameter, the others at their true values
fixed = {k: v for k, v in true_params.items() if k != "sigma_target"}
sigmas = torch.linspace(5.0, 50.0, 50, device=dev, dtype=dt_)
t0 = time.perf_counter()
obj, best = candidate_search(x, lqg.BoundedActor, dict(sigma_target=sigmas), **fixed)
torch.cuda.synchronize()
print(f"sweep over 50 sigmas x 50 trials x T={T}: {1e3 * (time.perf_counter() - t0):.1f} ms; argmax sigma = {float(sigmas[best]):.2f}")

# --- cell 42: d/d sigma at 28
val, grad = value_and_grad(x, lqg.BoundedActor, dict(sigma_target=28.0), **fixed)
print(f"ll(28) = {val:.4f}, d ll / d sigma = {grad['sigma_target']:.6f}  (adjoint sweep)")
sigma = torch.tensor(28.0, device=dev, dtype=dt_, requires_grad=True)              # the same thing, spelled like jax.grad
lqg.BoundedActor(T=T, sigma_target=sigma, device=dev, dtype=dt_, **fixed).log_likelihood(x).sum().backward()
_, grad_fd = value_and_grad(x, lqg.BoundedActor, dict(sigma_target=28.0), method="fd", **fixed)
print(f"torch.autograd: {float(sigma.grad):.6f}; finite differences: {grad_fd['sigma_target']:.6f}")

# --- lqg/infer/mle.py: Adam on the log-likelihood (two free parameters)
t0 = time.perf_counter()
params, losses = max_likelihood(x, lqg.BoundedActor, steps=200, step_size=0.05, action_cost=0.05, sigma_cursor=1.0)
print(f"max_likelihood (200 Adam steps, {1e3 * (time.perf_counter() - t0) / 200:.2f} ms/step):",
      {k: round(v, 3) for k, v in params.items()}, f"loss {float(losses[0]):.2f} -> {float(losses[-1]):.2f}")

# --- cell 52: belief tracking
bt = lqg.BoundedActor(T=T, device=dev, dtype=dt_, **true_params).belief_tracking_distribution(x)
print("belief_tracking_distribution:", bt.shape(), "loc[0, :3] =", bt.loc[0, :3].tolist())
