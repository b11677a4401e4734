import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer import graphed
dev = torch.device("cuda")
cls = lqg_amd.BoundedActor
truth = cls(T=500, device=dev, dtype=torch.float64)
with torch.no_grad():
    x = truth.simulate(3, n=50)[..., :2].contiguous()
names = ["action_variability", "sigma_target", "sigma_cursor", "action_cost"]
ev = graphed.GraphedFiniteDifference(x, cls, names, 1, h=1e-4)
m = ev._model(ev.theta)
print("decoupled:", m.decoupled(2, None))
try:
    print("capture:", ev.capture())
except Exception:
    traceback.print_exc()
