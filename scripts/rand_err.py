import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as OC
OC.build()
import lqg_amd
from test_gpu_random import random_system, SHAPES
from gpu_common import to_spec, np_
worst = 0
for case in range(10):
    rng = np.random.default_rng(1000 + case)
    x, b, u, y = SHAPES[case % len(SHAPES)]
    T = int(rng.integers(5, 60)); tv, affine = bool(case & 1), bool(case & 2)
    actor, dyn = random_system(rng, x, b, u, y, T, tv, affine)
    d = x if case % 3 else max(1, x - 1) if (x, b, u, y) != (4, 4, 2, 4) else x
    if (x, d) not in ((2, 2), (4, 4), (4, 2)): d = x
    S0 = None
    if case % 4 == 3:
        M = rng.standard_normal((b, b)); S0 = M @ M.T / b + 0.3 * np.eye(b)
    X, _, _, _ = OC.simulate(actor, dyn, rng.standard_normal((3, T, x)), rng.standard_normal((3, T, y)), Sigma0=S0)
    xs = X[..., :d]
    ref = OC.log_likelihood(actor, dyn, xs, S0)
    sys_ = lqg_amd.System(actor=to_spec(actor, torch.float32), dynamics=to_spec(dyn, torch.float32))
    S0t = None if S0 is None else torch.as_tensor(S0, dtype=torch.float32, device="cuda")
    got = np_(sys_.log_likelihood(torch.as_tensor(xs, dtype=torch.float32, device="cuda"), Sigma0=S0t))
    e = np.abs(got - ref).max() / np.abs(ref).max()
    worst = max(worst, e)
    print(case, (x, b, u, y), T, "ll scale %.1f rel err %.2e" % (np.abs(ref).max(), e))
print("worst", worst)
