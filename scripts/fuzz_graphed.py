"""Randomised cross-check of the graphed objective (hipGraph replay, affine constructor map, merged components) against the
eager evaluation: random zoo class / dim, data, parameter vectors over wide ranges; value 1e-11, gradient 1e-6 of its scale."""
import os, sys, random, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer.gradient import value_and_grad
from lqg_amd.infer.models import get_model_params
dev = torch.device("cuda")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
classes = [("BoundedActor", 1), ("BoundedActor", 2), ("SubjectiveActor", 1), ("SubjectiveActor", 2), ("OptimalActor", 1),
           ("RelativeObservationBoundedActor", 1), ("PointMassBoundedActor", 0)]
bad, worst_v, worst_g, graphed = [], 0.0, 0.0, 0
for case in range(N):
    name, dim = rng.choice(classes)
    cls = getattr(lqg_amd, name)
    T, n = rng.randint(40, 500), rng.choice([1, 2, 7, 50, 200])
    fixed = dict(dim=dim) if dim else {}
    truth = cls(T=T, device=dev, dtype=torch.float64, **fixed)
    d = truth.xdim if name != "PointMassBoundedActor" else 2
    with torch.no_grad():
        x = truth.simulate(case, n=n)[..., :d].contiguous()
    x = torch.cat([x, x[:, -1:]], dim=1)
    names = [k for k in get_model_params(cls) if k not in ("damping", "m", "tau")]
    for rep in range(3):                      # several parameter vectors through the SAME captured graph
        p = {k: math.exp(rng.uniform(math.log(0.1), math.log(10.0))) for k in names}
        os.environ["LQG_GRAPH"] = "0"
        v0, g0 = value_and_grad(x, cls, p, method="fd", **fixed)
        os.environ["LQG_GRAPH"] = "1"
        v1, g1 = value_and_grad(x, cls, p, method="fd", **fixed)
        if not math.isfinite(v0):
            continue
        ev = abs(v1 / v0 - 1)
        s_ = max(abs(v) for v in g0.values())
        eg = max(abs(g1[k] - g0[k]) for k in g0) / s_
        worst_v, worst_g = max(worst_v, ev), max(worst_g, eg)
        if not (ev < 1e-11 and eg < 1e-6):
            bad.append((case, name, dim, T, n, ev, eg))
print("cases", N, "x 3 vectors; worst value rel diff", worst_v, "worst gradient diff / scale", worst_g, "failures", bad)
