"""Host cost of building a throw-away LogLikelihoodPlan (what System.log_likelihood pays per call when the parameters change
every call: finite differences, NUTS)."""
import os, sys, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd.plan import LogLikelihoodPlan
dev = torch.device("cuda")
for scan in ("0", "1"):
    os.environ["LQG_SCAN"] = scan
    sig = torch.linspace(4.0, 30.0, 9, device=dev, dtype=torch.float64)
    x = None
    ts = []
    for it in range(30):
        m = lqg_amd.BoundedActor(T=500, sigma_target=sig + 0.01 * it, device=dev, dtype=torch.float64)
        if x is None:
            with torch.no_grad():
                x = m.simulate(1, n=50)[0].contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = LogLikelihoodPlan(m, x)
        t1 = time.perf_counter()
        p.run(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    ts = ts[5:]
    print("LQG_SCAN", scan, "build ms %.3f  run ms %.3f" % (1e3 * sorted(t[0] for t in ts)[len(ts) // 2], 1e3 * sorted(t[1] for t in ts)[len(ts) // 2]))
    pr = cProfile.Profile()
    m = lqg_amd.BoundedActor(T=500, sigma_target=sig + 0.5, device=dev, dtype=torch.float64)
    pr.enable(); p = LogLikelihoodPlan(m, x); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
