import sys, os, numpy as np, torch
sys.path[:0] = [os.getcwd(), os.getcwd() + "/tests", os.getcwd() + "/oracle"]
os.environ["LQG_COOP_ADJOINT"] = "1"
from conftest import load_golden
from gpu_common import system_from_golden
from lqg_amd import grad as G
import lqg_adjoint_np as ADJ
for name in ("timevarying_T30", "subjective1d_T50", "delay12_subjective1d_T30"):
    g, actor, dyn = load_golden(name)
    x = g["x"]; S0 = g.get("Sigma0")
    w = np.linspace(0.5, 1.5, x.shape[0])
    s = system_from_golden(actor, dyn, torch.float64)
    xt = torch.as_tensor(x, device="cuda"); S0t = None if S0 is None else torch.as_tensor(S0, device="cuda")
    outs = []
    for rep in range(12):
        ll, bars, _ = G.raw_grad(s.actor, s.dynamics, xt, g=torch.as_tensor(w, device="cuda"), Sigma0=S0t)
        outs.append({k: v.clone() for k, v in bars.items()})
    bad = {k: max(float((outs[r][k] - outs[0][k]).abs().max()) for r in range(1, 12)) for k in outs[0] if not k in ("aQf","aS0") or True}
    print(name, "max deviation between replays:", {k: v for k, v in bad.items() if v > 0})
