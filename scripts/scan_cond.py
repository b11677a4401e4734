"""Accuracy of the time-parallel sweeps against the sequential fp64 sweeps as the observed block's conditioning degrades."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.plan import LogLikelihoodPlan, _observed_noise_cond
dev = torch.device("cuda")
for T in (50, 500):
    for av in (1e-3, 0.05, 0.5):
        for d in (2, 4):
            m = lqg_amd.PointMassBoundedActor(T=T, action_variability=av, device=dev, dtype=torch.float64)
            with torch.no_grad():
                x = m.simulate(3, n=16)[..., :d].contiguous()
            os.environ["LQG_SCAN"] = "0"; ref = LogLikelihoodPlan(m, x).run().clone()
            os.environ["LQG_SCAN"] = "1"; got = LogLikelihoodPlan(m, x).run().clone()
            print(f"T={T} av={av} d={d} cond={_observed_noise_cond(m, d):.2e} rel={float((got/ref-1).abs().max()):.2e} abs={float((got-ref).abs().max()):.2e}")
