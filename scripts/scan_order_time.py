"""System sweeps of the delay model (DelayedSubjectiveActor, m = 65, T = 500) by the time-parallel scans in their three level orders (hs Hillis-Steele, bk Brent-Kung, we work-efficient default) and by the
sequential cooperative kernels, against the number of systems (candidates): which route the plan should take where."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd import options
from lqg_amd.plan import LogLikelihoodPlan
from lqg_amd.tracking.delay import DelayedSubjectiveActor
dev = torch.device("cuda")
T, n = 500, 50
for dtype in (torch.float64, torch.float32):
    x = DelayedSubjectiveActor(T=T, device=dev, dtype=dtype).simulate(3, n=n)[..., :2].contiguous()
    for B in (1, 2, 4, 13, 32, 64):
        sig = torch.linspace(4.0, 9.0, B, device=dev, dtype=dtype)
        m = DelayedSubjectiveActor(T=T, device=dev, dtype=dtype, sigma_target=sig)
        row = {}
        ref = None
        for label, ov in (("seq", dict(SCAN="0")), ("hs", dict(SCAN="1", SCAN_ORDER="0")), ("bk", dict(SCAN="1", SCAN_ORDER="2")),
                          ("we", dict(SCAN="1", SCAN_ORDER="1"))):         # we: Brent-Kung around a scan of the block totals (the default)
            with options.override(**ov):
                p = LogLikelihoodPlan(m, x)
                out = p.run().clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    p.run()
                torch.cuda.synchronize()
                row[label] = (time.perf_counter() - t0) / 3 * 1e3
            if ref is None:
                ref = out.double()
            else:
                row[label + "_err"] = float((out.double() / ref - 1).abs().max())
        print(str(dtype)[6:], "systems", B, " ".join("%s %.3g" % kv for kv in row.items()), flush=True)
