"""Latency of one log-likelihood evaluation for few systems x few trials (the inner loop of MLE / NUTS / finite differences):
sequential lane kernels (fused (system, trial) pairs where they apply) against the time-parallel path (scans + chunked
per-trial sweep).  Prints ms per plan.run() (median of 50 after warm-up)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.plan import LogLikelihoodPlan

dev = torch.device("cuda")
dtype = torch.float64 if "f64" in sys.argv else torch.float32
T = 500


def bench(plan, reps=50):
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        plan.run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2]


for model in ("bounded", "pointmass"):
    for n_sys in (1, 4, 9, 16, 32, 64):
        for n_tr in (8, 50, 400):
            sig = torch.linspace(4.0, 30.0, n_sys, device=dev, dtype=dtype) if n_sys > 1 else 6.0
            if model == "bounded":
                m = lqg_amd.BoundedActor(T=T, sigma_target=sig, device=dev, dtype=dtype)
            else:
                m = lqg_amd.PointMassBoundedActor(T=T, sigma_target=sig, action_variability=0.5, device=dev, dtype=dtype)
            with torch.no_grad():
                x = m.simulate(1, n=n_tr)
                x = (x[0] if n_sys > 1 else x)[..., :2].contiguous()
            out = []
            for scan in ("0", "1"):
                os.environ["LQG_SCAN"] = scan
                p = LogLikelihoodPlan(m, x)
                out.append(bench(p))
            print(f"{model:9s} {str(dtype)[-7:]} systems {n_sys:3d} trials {n_tr:4d}: sequential {out[0]:.3f} ms   time-parallel {out[1]:.3f} ms", flush=True)
