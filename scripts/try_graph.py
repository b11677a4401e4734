import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.plan import LogLikelihoodPlan
dev = torch.device("cuda")
truth = lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64)
with torch.no_grad():
    x = truth.simulate(3, n=50)
C = 9
theta = torch.tensor([[0.5, 6.0, 3.0, 0.1]] * C, dtype=torch.float64, device=dev)
theta[:, 1] += torch.linspace(0, 1, C, device=dev, dtype=torch.float64)
names = ["action_variability", "sigma_target", "sigma_cursor", "action_cost"]

def build():
    return lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64, **{k: theta[:, i] for i, k in enumerate(names)})

m = build()
plan = LogLikelihoodPlan(m, x)
ref = plan.run().clone()
print("eager ok", ref.shape, plan.description[:80])
for mode in ("global", "thread_local", "relaxed"):
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                m2 = build()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g, capture_error_mode=mode):
            m2 = build()
        g.replay(); torch.cuda.synchronize()
        print(mode, "constructor captured OK; A equal:", torch.equal(m2.actor.W, m.actor.W))
        break
    except Exception as e:
        print(mode, "FAILED:", str(e)[:300])
        torch.cuda.synchronize()
