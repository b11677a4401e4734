"""Wall time of the reference's maximum-likelihood fit (lqg/infer/mle.py: 2000 Adam steps) through lqg_amd.infer.max_likelihood."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer import max_likelihood
dev = torch.device("cuda")
truth = lqg_amd.BoundedActor(T=500, sigma_target=12.0, sigma_cursor=2.0, action_cost=0.3, action_variability=0.4, device=dev, dtype=torch.float64)
with torch.no_grad():
    x = truth.simulate(3, n=50)
x = torch.cat([x, x[:, -1:]], dim=1)
for graph in ("1", "0"):
    os.environ["LQG_GRAPH"] = graph
    max_likelihood(x, lqg_amd.BoundedActor, steps=20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    params, losses = max_likelihood(x, lqg_amd.BoundedActor, steps=2000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"LQG_GRAPH={graph}: 2000 Adam steps in {dt:.2f} s ({dt / 2000 * 1e3:.3f} ms per step); loss {float(losses[0]):.2f} -> {float(losses[-1]):.2f}; fit {({k: round(v, 3) for k, v in params.items()})}", flush=True)
