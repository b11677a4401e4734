import sys, os, numpy as np, torch
sys.path[:0] = [os.getcwd(), os.getcwd() + "/tests", os.getcwd() + "/oracle"]
os.environ["LQG_COOP_ADJOINT"] = "1"
from conftest import load_golden
from gpu_common import system_from_golden
from lqg_amd import grad as G
import lqg_adjoint_np as ADJ
g, actor, dyn = load_golden("timevarying_T30")
x, S0 = g["x"], g["Sigma0"]
w = np.linspace(0.5, 1.5, x.shape[0])
ll_ref, ga, gd, S0b = ADJ.loglik_grad(actor, dyn, x, w, S0)
s = system_from_golden(actor, dyn, torch.float64)
xt, S0t = torch.as_tensor(x, device="cuda"), torch.as_tensor(S0, device="cuda")
ll, bars, _ = G.raw_grad(s.actor, s.dynamics, xt, g=torch.as_tensor(w, device="cuda"), Sigma0=S0t)
print("ll err", np.abs(ll.cpu().numpy() - g["ll"]).max())
tot = {k: v.sum(1)[0].cpu().numpy() for k, v in bars.items()}
for k, ref in (("dA", gd["A"]), ("dB", gd["B"]), ("dF", gd["F"]), ("aA", None), ("aF", ga["F"]), ("aQ", ga["Q"]), ("aR", ga["R"])):
    got = tot[k] if k != "aA" else tot["aA"] + tot["aA2"]
    ref = ga["A"] if k == "aA" else ref
    err = np.abs(got - ref).reshape(got.shape[0], -1).max(1)
    print(k, " ".join("%.0e" % e for e in err))
