"""ms per log-likelihood evaluation of the reference's largest model (DelayedSubjectiveActor: x=26, b=39, m=65), T=500."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as bc
from lqg_amd.tracking.delay import DelayedSubjectiveActor
for dt in (torch.float32, torch.float64):
    md = DelayedSubjectiveActor(T=500, device="cuda", dtype=dt)
    x = md.simulate(21, n=256)[..., :2].contiguous()
    for scan in ("0", ""):              # sequential cooperative sweeps / the default (time-parallel sweeps for few long systems)
        os.environ["LQG_SCAN"] = scan
        for nt in (1, 256):
            ll, ph = bc.timed_loglik(md, x[:nt].contiguous(), 5)
            print("LQG_SCAN=%s" % (scan or "default"), dt, nt, {k: round(v, 3) for k, v in ph.items() if k.endswith("_ms")})
    os.environ.pop("LQG_SCAN")
    print("oracle", bc.oracle_check(md, x[:2].contiguous(), md.log_likelihood(x[:2].contiguous()), n_samples=2))
