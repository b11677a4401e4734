"""Where is the fp32 tail of configs 3 / 4?  Worst (candidate, trial) pairs of config 3 with their parameters, the
per-candidate maximum against each parameter, and config 4 on the SAME fp32-representable inputs in both precisions."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lqg_amd
from lqg_amd import workload
from bench_configs import hand2d_system
DEV = "cuda"
m, p = workload.bounded_system(4096, 1067, seed=5, device=DEV, dtype=torch.float32)
truth = lqg_amd.BoundedActor(T=1067, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, action_variability=0.5, device=DEV, dtype=torch.float32)
x = truth.simulate(13, n=1024)
ll = m.log_likelihood(workload.pack_trials(x)).clone()
ll64 = m.to(torch.float64).log_likelihood(workload.pack_trials(x.double())).clone()
rel = (ll.double() / ll64 - 1).abs()
print(json.dumps({"case": "cfg3", "max": float(rel.max()), "median": float(rel.median())}))
pc = rel.max(dim=1).values
order = torch.argsort(pc, descending=True)
for c in order[:24].tolist():
    print(json.dumps({"cand": c, "max_rel": float(pc[c]), "median_rel": float(rel[c].median()), "ll": float(ll64[c].mean()),
                      **{k: float(v[c]) for k, v in p.items()}}))
# rank correlation of the per-candidate max with each parameter
for k, v in p.items():
    a = torch.argsort(torch.argsort(v.double())).double(); b = torch.argsort(torch.argsort(pc)).double()
    print(k, "spearman", float(torch.corrcoef(torch.stack([a, b]))[0, 1]))
# forcing the fp64 operator stream (time-parallel scans) for a slice of candidates: is the tail in the operators?
os.environ["LQG_SCAN"] = "1"
try:
    sub = order[:64]
    ms = workload.slice_system(m, 0, 4096)
    from lqg_amd.spec import LQGSpec
    def take(spec):
        return LQGSpec(**{f: (getattr(spec, f)[sub] if getattr(spec, f).dim() == workload._batched_ndim(f) else getattr(spec, f)) for f in LQGSpec._fields})
    a = take(m.actor)
    msub = lqg_amd.System(actor=a, dynamics=a)
    l2 = msub.log_likelihood(workload.pack_trials(x)).clone()
    r2 = (l2.double() / ll64[sub] - 1).abs()
    print(json.dumps({"case": "cfg3 worst-64 candidates, scan operators (fp64) + fp32 trial sweep", "max": float(r2.max()),
                      "before": float(rel[sub].max())}))
except Exception as e:
    print("scan variant failed:", repr(e))
del os.environ["LQG_SCAN"]
# config 4 on the same inputs
m64 = hand2d_system(1000, DEV, torch.float64)
x64 = m64.simulate(14, n=32768)[..., :4].contiguous()
a = m64.log_likelihood(workload.pack_trials(x64)).clone()
m32 = m64.to(torch.float32)
b = m32.log_likelihood(workload.pack_trials(x64.float())).clone()
r = (b.double() / a - 1).abs()
print(json.dumps({"case": "cfg4 fp32 path on rounded inputs vs fp64 path on unrounded inputs", "max": float(r.max()), "p99": float(torch.quantile(r, 0.99)), "median": float(r.median())}))
a2 = m32.to(torch.float64).log_likelihood(workload.pack_trials(x64.float().double())).clone()
r = (b.double() / a2 - 1).abs()
print(json.dumps({"case": "cfg4 same fp32-representable inputs", "max": float(r.max()), "p99": float(torch.quantile(r, 0.99)), "median": float(r.median()), "ll_abs_min": float(a2.abs().min()), "ll_mean": float(a2.mean())}))
r = (a2 / a - 1).abs()
print(json.dumps({"case": "cfg4 fp64 path: rounded vs unrounded inputs", "max": float(r.max()), "p99": float(torch.quantile(r, 0.99)), "median": float(r.median())}))
