"""Per-trial sweep: one pass against time-chunked, as the number of trials (waves in flight) grows."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd import workload
from lqg_amd.plan import LogLikelihoodPlan
import bench_configs as bc
dev = torch.device("cuda")
m = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=torch.float32)
for log2n in (10, 12, 14, 16, 17, 18, 19, 20):
    n = 1 << log2n
    x = workload.pack_trials(m.simulate(12, n=n)[..., :2].contiguous())
    row = []
    for ch in ("0", "2", "4", "8", "16", ""):
        if ch:
            os.environ["LQG_TRIAL_CHUNKS"] = ch
        else:
            os.environ.pop("LQG_TRIAL_CHUNKS", None)
        ll, ph = bc.timed_loglik(m, x, 5)
        row.append("%s:%.3f" % (ch or "auto", ph["trial_ms"]))
    print("trials 2^%d (%d waves): per-trial sweep ms  " % (log2n, n // 64) + "  ".join(row), flush=True)
    del x
    torch.cuda.empty_cache()
