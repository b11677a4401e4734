"""Host-side profile of one-theta value+grad by central differences (9 candidates x 50 trials, BoundedActor T=500)."""
import os, sys, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer.gradient import value_and_grad
dev = torch.device("cuda")
truth = lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64)
with torch.no_grad():
    x = truth.simulate(3, n=50)
params = dict(action_variability=0.5, sigma_target=6.0, sigma_cursor=3.0, action_cost=0.1)
for _ in range(20):
    value_and_grad(x, lqg_amd.BoundedActor, params, method="fd")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    value_and_grad(x, lqg_amd.BoundedActor, params, method="fd")
torch.cuda.synchronize()
print("ms per eval", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    value_and_grad(x, lqg_amd.BoundedActor, params, method="fd")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
