"""Wall time of the reference's NUTS driver call (lqg/infer/utils.py:14: infer(x, num_samples, num_warmup, model=...)) through
lqg_amd.infer.infer, chains batched on the candidate axis; hipGraph replay of the potential against eager launches."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer import infer
dev = torch.device("cuda")
truth = lqg_amd.BoundedActor(T=500, sigma_target=12.0, sigma_cursor=2.0, action_cost=0.3, action_variability=0.4, device=dev, dtype=torch.float64)
with torch.no_grad():
    x = truth.simulate(3, n=50)
x = torch.cat([x, x[:, -1:]], dim=1)
for graph in ("1", "0"):
    os.environ["LQG_GRAPH"] = graph
    t0 = time.perf_counter()
    mcmc = infer(x, num_samples=200, num_warmup=200, model=lqg_amd.BoundedActor, num_chains=4, seed=1, progress_bar=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = mcmc.get_samples()
    ex = mcmc.get_extra_fields() if hasattr(mcmc, "get_extra_fields") else {}
    n_eval = getattr(mcmc, "evaluations", None)
    print(f"LQG_GRAPH={graph}: 4 chains x (200 + 200) NUTS transitions in {dt:.2f} s; likelihood evaluations {n_eval}; "
          f"posterior means { {k: round(float(v.mean()), 3) for k, v in s.items()} }", flush=True)
