import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import lqg_amd
from lqg_amd.infer import gradient
from lqg_amd.tracking.delay import DelayedSubjectiveActor
dev = torch.device("cuda")
def run(cls, p, n, T, reps=3, **kw):
    with torch.no_grad():
        m = cls(T=T, device=dev, dtype=torch.float64, **kw)
        d = 2
        x = m.simulate(0, n=n)[..., :d].contiguous()
    x = torch.cat([x, x[:, -1:]], dim=1)
    for mode in ("0", "1"):
        os.environ["LQG_COOP_ADJOINT"] = mode
        gradient.value_and_grad(x, cls, p, method="adjoint", **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            gradient.value_and_grad(x, cls, p, method="adjoint", **kw)
        torch.cuda.synchronize()
        print(cls.__name__, "n", n, "T", T, "COOP_ADJOINT", mode, "%.2f ms" % ((time.perf_counter() - t0) / reps * 1e3), flush=True)
run(lqg_amd.BoundedActor, dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4), 50, 500)
run(lqg_amd.SubjectiveActor, dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4), 50, 500)
run(DelayedSubjectiveActor, dict(sigma_target=6.0, sigma_cursor=3.0), 50, 100, reps=2)
