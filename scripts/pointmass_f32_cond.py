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

# ---- is it the kernels or the fp32-built specs?  the fp64 path over the fp64 IMAGE of the fp32 model (same matrices, widened)
l64_img = m32.to(torch.float64).log_likelihood(x.double())
err_img = ((l32 - l64_img).abs() / l64_img.abs()).max(1).values.cpu().numpy()
print("fp32 routes against the fp64 path on the fp64 IMAGE of the SAME fp32 spec matrices: max %.1e (bad > 1e-6: %d)" % (err_img.max(), (err_img > 1e-6).sum()))
dif = {f: float((getattr(m32.dynamics, f).double() - getattr(m64.dynamics, f)).abs().max() / getattr(m64.dynamics, f).abs().max()) for f in ("A", "B", "V")}
print("spec matrices, fp32-built vs fp64-built (max abs diff / max abs):", dif)
worst = int(np.argmax(err))
print("worst candidate", worst, {k: float(v[worst]) for k, v in kw.items()})
print("V fp32-built:\n", m32.dynamics.V[worst, 0].cpu().numpy(), "\nV fp64-built:\n", m64.dynamics.V[worst, 0].cpu().numpy())

# ---- which fp32 route carries it?  the bad candidates alone, every route switch against the fp64 image
from lqg_amd import options
sel = np.nonzero(err_img > 1e-6)[0]
if len(sel):
    idx = torch.as_tensor(sel, device=dev)
    sub = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32, **{k: v[idx] for k, v in kw.items()})
    ref = sub.to(torch.float64).log_likelihood(x.double())
    for label, ov, xx in (("default (n=8)", {}, x), ("MIXED=0", dict(MIXED=0), x), ("SCAN=0", dict(SCAN="0"), x), ("SCAN=1", dict(SCAN="1"), x),
                          ("NO_SPECIALIZE", dict(NO_SPECIALIZE=1), x), ("COOP=1", dict(COOP="1"), x), ("n=1", {}, x[:1]), ("n=2", {}, x[:2])):
        with options.override(**ov):
            l = sub.log_likelihood(xx).double()
        r = ref[:, :xx.shape[0]]
        print("  %-16s max rel err %.1e" % (label, float(((l - r).abs() / r.abs()).max())))
    out = os.path.join(R, "gpurun_out", "pointmass_worst.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    w = m32
    np.savez(out, x=x.cpu().numpy(), worst=worst, err=err_img,
             **{"act_" + f: getattr(w.actor, f)[worst, 0].cpu().numpy() for f in ("A", "B", "F", "V", "W", "Q", "R")},
             **{"dyn_" + f: getattr(w.dynamics, f)[worst, 0].cpu().numpy() for f in ("A", "B", "F", "V", "W")})
    print("wrote", out)

# ---- the metric itself: |ll| of the bad candidates against the scale of the sum it is the result of.  A log-likelihood is a sum of T
# per-step terms of either sign (0.5 z'z > 0, the half log-determinant < 0 for tight predictions); where they cancel, |ll| is far below
# the magnitude the fp32 arithmetic works at, and an error RELATIVE TO |ll| says nothing about the kernel.
absll = l64_img.abs().cpu().numpy(); abserr = (l32 - l64_img).abs().cpu().numpy()
print("bad candidates: |ll| min/median %.3g / %.3g  (good candidates: %.3g / %.3g);  T*d = %d" % (
    absll[bad].min(), np.median(absll[bad]), absll[~bad].min(), np.median(absll[~bad]), T * 2))
print("absolute error: bad max %.2e, good max %.2e;  relative to max(|ll|, T*d): bad %.1e, good %.1e" % (
    abserr[bad].max(), abserr[~bad].max(), (abserr / np.maximum(absll, T * 2))[bad].max(), (abserr / np.maximum(absll, T * 2))[~bad].max()))
