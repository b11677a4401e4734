"""ms of System.simulate for the bench's workload generator chunk (2^15 systems x 1 trial, T = 500, headline model)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqg_amd import workload
dev = torch.device("cuda")
for dt in (torch.float32, torch.float64):
    system, _ = workload.headline_system(1 << 15, 500, seed=1, device=dev, dtype=dt)
    for _ in range(2):
        x = system.simulate(3, n=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        x = system.simulate(3 + i, n=1)
    torch.cuda.synchronize()
    print(dt, "simulate 2^15 systems x 1 trial, T=500: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), tuple(x.shape))
