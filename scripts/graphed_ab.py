import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.infer.gradient import value_and_grad
dev = torch.device("cuda")
for cls, params in ((lqg_amd.BoundedActor, dict(action_variability=0.5, sigma_target=6.0, sigma_cursor=3.0, action_cost=0.1)),
                    (lqg_amd.SubjectiveActor, dict(action_cost=0.2, action_variability=0.5, subj_noise=1.0, subj_vel_noise=0.5, sigma_target=6.0, sigma_cursor=3.0)),
                    (lqg_amd.PointMassBoundedActor, dict(action_variability=0.5, sigma_target=6.0, sigma_cursor=3.0, action_cost=0.1))):
    truth = cls(T=500, device=dev, dtype=torch.float64)
    with torch.no_grad():
        x = truth.simulate(3, n=50)[..., :2].contiguous()
    os.environ["LQG_GRAPH"] = "0"
    v0, g0 = value_and_grad(x, cls, params, method="fd")
    for _ in range(5): value_and_grad(x, cls, params, method="fd")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): value_and_grad(x, cls, params, method="fd")
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 100 * 1e3
    os.environ["LQG_GRAPH"] = "1"
    v1, g1 = value_and_grad(x, cls, params, method="fd")
    for _ in range(5): value_and_grad(x, cls, params, method="fd")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        p2 = dict(params); p2["sigma_target"] = 6.0 + 0.01 * i
        value_and_grad(x, cls, p2, method="fd")
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 100 * 1e3
    err = max(abs(g1[k] / g0[k] - 1) for k in g0)
    print(f"{cls.__name__}: eager {te:.3f} ms  graphed {tg:.3f} ms  value rel diff {abs(v1 / v0 - 1):.1e}  grad rel diff {err:.1e}", flush=True)
