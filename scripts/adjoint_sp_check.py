"""GPU check of the round-5 split adjoint (csrc/lqg_adjoint_sp.hpp) against oracle/lqg_adjoint_np.py: raw bars on the DENSE
pattern (every field requires grad -> full masks) for 1, 2 and several trials per system, both dtypes.
    python scripts/adjoint_sp_check.py [--compile-only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
import lqg_amd
from lqg_amd import specialize
from conftest import load_golden

CASES = ["bounded_T100", "subjective1d_T50", "relobs_T40", "pointmass_d2_T50"]


def build(name, dtype, device, n):
    import test_adjoint
    return test_adjoint._dense_grad_system(name, dtype, n, device=device)


def build_old(name, dtype, device, n):
    from gpu_common import system_from_golden
    g, actor, dyn = load_golden(name)
    x = g["x"]
    rng = np.random.default_rng(5)
    while x.shape[0] < n:
        x = np.concatenate([x, x[:1] + 0.05 * rng.standard_normal(x[:1].shape).cumsum(1)], 0)
    x = x[:n]
    s = system_from_golden(actor, dyn, dtype, device)
    ti = lambda spec: spec._replace(**{f: getattr(spec, f)[:1].expand_as(getattr(spec, f)) for f in ("A", "B", "F", "V", "W", "Q", "R", "q", "P", "r")})
    a = ti(s.actor)
    d = a if s.actor is s.dynamics else ti(s.dynamics)
    # every differentiable field requires grad -> adjoint_pattern() is the dense pattern
    def rg(spec):
        out = {}
        for f in spec._fields:
            t = getattr(spec, f)
            if f in ("A", "B", "F", "V", "W", "Q", "R"):
                base = t[:1].clone().requires_grad_(True)
                t = base.expand_as(getattr(spec, f))
            out[f] = t
        return spec._replace(**out)
    a2 = rg(a)
    d2 = rg(d)
    return lqg_amd.System(actor=a2, dynamics=d2), actor, dyn, x, g


def main():
    compile_only = "--compile-only" in sys.argv
    device = "cpu" if compile_only else "cuda"
    worst = 0.0
    for name in CASES:
        for dtype, tol in ((torch.float64, 1e-8), (torch.float32, 3e-4)):
            for n in (1, 2, 5):
                s, actor, dyn, x, g = build(name, dtype, device, n)
                dims, masks, key, live = specialize.adjoint_pattern(s, x.shape[-1])
                if compile_only:
                    if dtype == torch.float64 and n == 1:
                        print(name, dims, key, specialize.compile_adjoint_pattern(key, dims, masks, verbose=True, live=live), flush=True)
                    continue
                import lqg_adjoint_np as ADJ
                from lqg_amd import grad as G
                w = np.linspace(0.5, 1.5, n)
                ll_ref, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x, w)
                with torch.no_grad():
                    sw = G.Sweep(s.actor, s.dynamics, torch.as_tensor(x, dtype=dtype, device="cuda"), system=s)
                    assert sw.sp is not None
                    ll = sw.forward()
                    bars = sw.reverse(torch.as_tensor(w, dtype=dtype, device="cuda"))
                e_ll = np.abs(ll.double().cpu().numpy() - ll_ref).max() / np.abs(ll_ref).max()
                tot = {k: v.sum(1)[0].double().cpu().numpy() for k, v in bars.items()}
                sym2 = lambda M: M + M.T
                got = {"dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"], "dV": sym2(tot["dVV"]) @ dyn["V"][0],
                       "dW": sym2(tot["dWW"]) @ dyn["W"][0], "aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"],
                       "aF": tot["aF"], "aV": sym2(tot["aVV"]) @ actor["V"][0], "aW": sym2(tot["aWW"]) @ actor["W"][0],
                       "aQ": tot["aQ"], "aR": tot["aR"], "aQf": tot["aQf"]}
                ref = {"d" + k: v.sum(0) for k, v in gd.items()}
                ref.update({"a" + k: (v.sum(0) if v.ndim == 3 else v) for k, v in ga.items()})
                scale = max(np.abs(v).max() for v in ref.values())
                errs = {k: float(np.abs(got[k] - r).max() / max(np.abs(r).max(), 1e-3 * scale)) for k, r in ref.items()}
                bad = {k: v for k, v in errs.items() if not v < tol}
                worst = max(worst, max(errs.values()) / tol)
                print(f"{name:20s} {str(dtype)[6:]:8s} n={n}  ll {e_ll:.1e}  max bar err {max(errs.values()):.2e} (tol {tol:g})", "BAD " + str({k: float("%.2g" % v) for k, v in errs.items()}) if bad else "ok", flush=True)
    if not compile_only:
        print("WORST/TOL", worst)


if __name__ == "__main__":
    main()
