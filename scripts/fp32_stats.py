"""fp32-vs-fp64 relative error of the log-likelihood at the full sizes of BASELINE configs 2, 3, 4 (quantiles)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lqg_amd
from lqg_amd import workload
from bench_configs import hand2d_system
DEV = "cuda"
def stats(name, rel):
    rel = rel.flatten()
    idx = torch.randperm(rel.numel(), device=rel.device)[: 1 << 20]
    r = rel[idx]
    print(json.dumps({"case": name, "max": float(rel.max()), "p999": float(torch.quantile(r, 0.999)),
                      "p99": float(torch.quantile(r, 0.99)), "median": float(r.median())}), flush=True)
m64 = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=DEV, dtype=torch.float64)
x64 = m64.simulate(12, n=65536)[..., :2].contiguous()
a = m64.log_likelihood(workload.pack_trials(x64)).clone(); b = m64.to(torch.float32).log_likelihood(workload.pack_trials(x64.float())).clone()
stats("cfg2", (b.double() / a - 1).abs())
m, _ = workload.bounded_system(4096, 1067, seed=5, device=DEV, dtype=torch.float32)
truth = lqg_amd.BoundedActor(T=1067, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, action_variability=0.5, device=DEV, dtype=torch.float32)
x = truth.simulate(13, n=1024)
ll = m.log_likelihood(workload.pack_trials(x)).clone()
ll64 = m.to(torch.float64).log_likelihood(workload.pack_trials(x.double())).clone()
stats("cfg3", (ll.double() / ll64 - 1).abs())
m64 = hand2d_system(1000, DEV, torch.float64)
x64 = m64.simulate(14, n=32768)[..., :4].contiguous()
a = m64.log_likelihood(workload.pack_trials(x64)).clone(); b = m64.to(torch.float32).log_likelihood(workload.pack_trials(x64.float())).clone()
stats("cfg4", (b.double() / a - 1).abs())
