import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from gpu_common import np_, system_from_golden
g, actor, dyn = load_golden("pointmass_d4_T50")
dtype = torch.float32
for env in ({}, {"LQG_TRIAL_CHUNKS": "0"}, {"LQG_TRIAL_CHUNKS": "0", "LQG_MIXED": "0"}, {"LQG_MIXED": "0"},
            {"LQG_TRIAL_CHUNKS": "0", "LQG_NO_SPECIALIZE": "1"}, {"LQG_TRIAL_CHUNKS": "0", "LQG_NO_SPECIALIZE": "1", "LQG_MIXED": "0"}):
    os.environ.update(env)
    s = system_from_golden(actor, dyn, dtype)
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    ll = np_(s.log_likelihood(x))
    print(env, "n=%d" % x.shape[0], ll, g["ll"])
    for k in env: os.environ.pop(k)
s = system_from_golden(actor, dyn, dtype)
mu, Sig = s._moments(x, None)
print("moments finite:", bool(torch.isfinite(mu).all()), bool(torch.isfinite(Sig).all()), "first bad t (Sigma):",
      int((~torch.isfinite(Sig).flatten(1).all(1)).float().argmax()) if not torch.isfinite(Sig).all() else -1)
