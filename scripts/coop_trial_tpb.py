"""Row-parallel per-trial sweep of the delay model (k_coop_trial_rows, m = 65): most trials per workgroup (LQG_COOP_TRIAL_TPB) and 256- against 1024-thread workgroups ("w")
against the shape of the batch — every workgroup streams the step's whole operator block, trials that share a workgroup share that traffic."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd import options
from lqg_amd.plan import LogLikelihoodPlan
from lqg_amd.tracking.delay import DelayedSubjectiveActor
dev = torch.device("cuda")
T = 500
for dtype in (torch.float32, torch.float64):
    for B, n in ((4096, 120), (512, 120), (64, 120), (13, 50), (1, 256), (1, 2048)):
        if dtype == torch.float64 and B > 512:
            continue
        x = DelayedSubjectiveActor(T=T, device=dev, dtype=dtype).simulate(3, n=n)[..., :2].contiguous()
        m = DelayedSubjectiveActor(T=T, device=dev, dtype=dtype, sigma_target=torch.linspace(4.0, 9.0, B, device=dev, dtype=dtype))
        row, ref = {}, None
        for cap, wide in ((0, ""), (16, "0"), (32, "0"), (64, "0"), (128, "0"), (32, "1"), (64, "1"), (128, "1")):       # (0, ""): the rule
            with options.override(SCAN="0", COOP_TRIAL_TPB=cap, COOP_TRIAL_WIDE=wide):
                p = LogLikelihoodPlan(m, x, events=True)
                out = p.run().clone()
                reps = 1 if B > 512 else 3
                torch.cuda.synchronize()
                ms = []
                for _ in range(reps):
                    p.run()
                    torch.cuda.synchronize()
                    ms.append(p.phase_ms()[2])
                row[(cap, wide)] = min(ms)
            ref = out if ref is None else ref
            assert float((out.double() / ref.double() - 1).abs().max()) < 1e-5
        print(str(dtype)[6:], "systems", B, "trials", n, "per-trial sweep ms by cap:", " ".join("%d%s: %.3g" % (c, "w" if w == "1" else "", v) for (c, w), v in row.items()), flush=True)
