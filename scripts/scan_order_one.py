"""One configuration of scripts/scan_order_time.py for a kernel trace: python3 scripts/scan_order_one.py <systems> <order 0|1> [f32]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, lqg_amd
from lqg_amd import options
from lqg_amd.plan import LogLikelihoodPlan
from lqg_amd.tracking.delay import DelayedSubjectiveActor
dev = torch.device("cuda")
B, order = int(sys.argv[1]), sys.argv[2]
dtype = torch.float32 if len(sys.argv) > 3 else torch.float64
x = DelayedSubjectiveActor(T=500, device=dev, dtype=dtype).simulate(3, n=50)[..., :2].contiguous()
m = DelayedSubjectiveActor(T=500, device=dev, dtype=dtype, sigma_target=torch.linspace(4.0, 9.0, B, device=dev, dtype=dtype))
with options.override(SCAN="1", SCAN_ORDER=order):
    p = LogLikelihoodPlan(m, x)
    for _ in range(4):
        p.run()
torch.cuda.synchronize()
