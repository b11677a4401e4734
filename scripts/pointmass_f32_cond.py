"""fp32 error of PointMassBoundedActor candidates (fp32 default routes vs the fp64 path on the same inputs) against the predictor
the WIDE route uses, cond((V V')[:d, :d]) of the dynamics — is there a threshold that separates good from bad candidates?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch, lqg_amd
from lqg_amd import workload
dev = torch.device("cuda")
os.environ["LQG_F32_WIDE"] = "0"
B, T, n = 256, 500, 8
g = torch.Generator(device=dev); g.manual_seed(5)
names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost")
kw = {k: workload.log_uniform(B, *workload.RANGES[k], g, dev, torch.float32) for k in names}
m32 = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32, **kw)
m64 = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float64, **{k: v.double() for k, v in kw.items()})
truth = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32)
x = truth.simulate(3, n=n)[..., :2].contiguous()
l32 = m32.log_likelihood(x).double()
l64 = m64.log_likelihood(x.double())
err = ((l32 - l64).abs() / l64.abs()).max(1).values.cpu().numpy()
V = m64.dynamics.V[:, 0, :2, :]
ev = torch.linalg.eigvalsh(V @ V.transpose(-1, -2)).cpu().numpy()
cond = ev[:, -1] / np.maximum(ev[:, 0], 1e-300)
order = np.argsort(cond)
for lo, hi in ((0, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e7), (1e7, 1e9), (1e9, 1e30)):
    sel = (cond >= lo) & (cond < hi)
    if sel.any():
        print("cond in [%.0e, %.0e): %3d candidates, fp32 rel err max %.1e median %.1e" % (lo, hi, sel.sum(), err[sel].max(), np.median(err[sel])))
av = kw["action_variability"].cpu().numpy()
bad = err > 1e-6
print("bad candidates:", bad.sum(), "of", B, "| action_variability of bad: min %.2f max %.2f; of good: min %.2f max %.2f" % (av[bad].min() if bad.any() else 0, av[bad].max() if bad.any() else 0, av[~bad].min(), av[~bad].max()))

# ---- the predictor the whitening suggests: max_t max diag(chol(Sigma_oo,t)^-1) from the fp64 moments
xs = x[:1].double()
_, Sig = m64._moments(xs.expand(B, *xs.shape) if False else xs, None)            # Sigma[B, T, m, m]
Soo = Sig[..., :2, :2]
Lc = torch.linalg.cholesky(Soo)
gain = (1.0 / torch.diagonal(Lc, dim1=-2, dim2=-1)).amax((-1, -2)).cpu().numpy()   # max over steps and components
for lo, hi in ((0, 10), (10, 30), (30, 100), (100, 300), (300, 1000), (1000, 3000), (3000, 1e4), (1e4, 1e5), (1e5, 1e9)):
    sel = (gain >= lo) & (gain < hi)
    if sel.any():
        print("max whitening gain in [%g, %g): %3d candidates, fp32 rel err max %.1e median %.1e" % (lo, hi, sel.sum(), err[sel].max(), np.median(err[sel])))
