"""fp32 log-likelihood of the ill-conditioned golden case pointmass_d4_T50 (cond of the observed block ~1e12) on every
kernel mapping, against the golden fp64 value."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from gpu_common import np_, system_from_golden
g, actor, dyn = load_golden("pointmass_d4_T50")
for dtype in (torch.float32, torch.float64):
    for env in ({}, {"LQG_NO_SPECIALIZE": "1"}, {"LQG_COOP": "1"}, {"LQG_SCAN": "1"}, {"LQG_MIXED": "0"}):
        os.environ.update(env)
        try:
            s = system_from_golden(actor, dyn, dtype)
            x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
            ll = np_(s.log_likelihood(x)); ll1 = np_(s.log_likelihood(x[:1]))
            print(dtype, env, "multi-trial", np.abs(ll / g["ll"] - 1).max(), "one trial", np.abs(ll1 / g["ll"][:1] - 1).max(), "ll", g["ll"][:2])
        except Exception as e:
            print(dtype, env, "failed", repr(e)[:200])
        for k in env: os.environ.pop(k)
