"""Ad-hoc GPU sanity run (not a pytest file): prints error tables for every golden case."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from conftest import golden_names, load_golden, relerr
from gpu_common import np_, system_from_golden
from lqg_amd.control import lqr
from lqg_amd.belief import kf

for name in golden_names():
    g, actor, dyn = load_golden(name)
    for dtype in (torch.float64, torch.float32):
        s = system_from_golden(actor, dyn, dtype)
        S0 = torch.as_tensor(g["Sigma0"], dtype=dtype, device="cuda") if "Sigma0" in g else None
        gains = lqr.backward(s.actor); K = kf.forward(s.actor, S0)
        x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
        mu, Sig = s._moments(x, S0)
        ll = s.log_likelihood(x, Sigma0=S0)
        ll1 = s.log_likelihood(x[:1], Sigma0=S0)
        torch.cuda.synchronize()
        print(f"{name:22s} {str(dtype)[6:]:8s} L {relerr(np_(gains.L), g['L']):.1e} H {relerr(np_(gains.H), g['H']):.1e} "
              f"K {relerr(np_(K), g['K']):.1e} mu {relerr(np_(mu), g['mu']):.1e} Sig {relerr(np_(Sig), g['Sigma'][0]):.1e} "
              f"ll {np.abs(np_(ll)/g['ll']-1).max():.1e} ll1 {np.abs(np_(ll1)/g['ll'][:1]-1).max():.1e}", flush=True)
