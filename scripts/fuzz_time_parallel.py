"""Randomised cross-check of the time-parallel path (scans + chunked per-trial sweep) against the sequential kernels: random
zoo class, parameters over wide log-uniform ranges, horizon, trials, number of candidates, chunk count; fp64, 1e-9."""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch, lqg_amd
from lqg_amd.plan import LogLikelihoodPlan
dev = torch.device("cuda")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
MODE = sys.argv[3] if len(sys.argv) > 3 else "forced"
worst, bad, used = 0.0, [], 0
classes = [("BoundedActor", 1), ("BoundedActor", 2), ("SubjectiveActor", 1), ("SubjectiveActor", 2), ("OptimalActor", 1),
           ("RelativeObservationBoundedActor", 1), ("PointMassBoundedActor", 0)]
from lqg_amd.infer.models import get_model_params
for case in range(N):
    name, dim = rng.choice(classes)
    cls = getattr(lqg_amd, name)
    B = rng.choice([1, 1, 2, 3, 6])
    T = rng.randint(64, 700)
    n = rng.choice([3, 5, 17, 64, 130, 400])
    lu = lambda lo, hi: torch.exp(torch.rand(B, dtype=torch.float64) * (np.log(hi) - np.log(lo)) + np.log(lo)).to(dev)
    kw = {}
    for k in get_model_params(cls):
        if k in ("damping", "m", "tau"):
            continue
        kw[k] = lu(0.05, 20.0) if "sigma" in k or "noise" in k else lu(0.02, 3.0)
    if B == 1:
        kw = {k: float(v[0]) for k, v in kw.items()}
    if dim:
        kw["dim"] = dim
    m = cls(T=T, device=dev, dtype=torch.float64, **kw)
    d = m.xdim if name != "PointMassBoundedActor" else rng.choice([2, 2, 4])
    with torch.no_grad():
        x = m.simulate(case, n=n)
        x = (x[0] if B > 1 else x)[..., :d].contiguous()
    os.environ["LQG_SCAN"] = "0"; os.environ["LQG_TRIAL_CHUNKS"] = "0"
    ref = LogLikelihoodPlan(m, x).run().clone()
    if MODE == "forced":
        os.environ["LQG_SCAN"] = "1"             # wherever the scans are defined (skips the conditioning guard)
    else:
        os.environ.pop("LQG_SCAN")               # the default rule, conditioning guard included
    ch = rng.choice(["", "2", "5", "13", "31"])
    if ch:
        os.environ["LQG_TRIAL_CHUNKS"] = ch
    else:
        os.environ.pop("LQG_TRIAL_CHUNKS")
    p = LogLikelihoodPlan(m, x)
    got = p.run().clone()
    if not torch.isfinite(ref).all():
        continue
    err = float((got / ref - 1).abs().max())
    # forced scans on the fully observed point mass (cond of the observed noise block up to 1e10) lose digits by design; the
    # default rule must keep such systems on the sequential sweeps
    tol = 1e-9 if (MODE == "default" or not (name == "PointMassBoundedActor" and d == 4)) else float("inf")
    worst = max(worst, err if tol == 1e-9 else 0.0)
    if not err < tol:
        bad.append((case, name, dim, B, T, n, d, ch, err, all(w["scan"] for w in p.work)))
    used = used + int(all(w["scan"] for w in p.work))
print("mode", MODE, "cases", N, "scan path used in", used, "worst rel diff", worst, "failures", bad)
