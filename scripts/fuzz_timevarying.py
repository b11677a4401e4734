"""Fuzz of the time-varying route of the pattern libraries (round 6): random systems whose spec stacks move in time under a random
sparsity pattern — with / without a cross cost P, an explicit Sigma0, several trials, both storage layouts ([B][T][r][c] and the
canonical [T][r][c][B]) — log-likelihood through LogLikelihoodPlan (must stay on the pattern library) against the fp64 C oracle.
    python scripts/fuzz_timevarying.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as OC
import lqg_amd
from lqg_amd import workload
from lqg_amd.plan import LogLikelihoodPlan

OC.build()
dev = torch.device("cuda")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncases):
    x_, b_, u_, y_, d_ = [(2, 2, 1, 2, 2), (2, 3, 1, 2, 2)][case % 2]
    B, T = int(rng.integers(70, 400)), int(rng.integers(8, 40))
    n = int(rng.choice([1, 1, 2, 3, 5]))
    dtype = torch.float64 if rng.random() < 0.6 else torch.float32
    use_P, use_S0, canon = rng.random() < 0.4, rng.random() < 0.4, rng.random() < 0.6

    def mask(r, c, keep_diag=False, p=0.6):
        m = rng.random((r, c)) < p
        if keep_diag:
            m[np.arange(min(r, c)), np.arange(min(r, c))] = True
        if not m.any():
            m[0, 0] = True
        return m

    def stack(base, m, jit=0.05, sym=False):
        z = base[None, None] * (1.0 + jit * rng.standard_normal((B, T) + base.shape)) * m
        if sym:
            z = 0.5 * (z + np.swapaxes(z, -1, -2))
        return z

    mAa, mAd = mask(b_, b_, True), mask(x_, x_, True)
    Aa = stack(np.eye(b_) + 0.05 * rng.standard_normal((b_, b_)), mAa, 0.01)
    Ad = stack(np.eye(x_) + 0.05 * rng.standard_normal((x_, x_)), mAd, 0.01)
    Ba = stack(0.3 * rng.standard_normal((b_, u_)), mask(b_, u_))
    Bd = stack(0.3 * rng.standard_normal((x_, u_)), mask(x_, u_))
    mFa = mask(y_, b_, True)
    Fa = stack(np.eye(y_, b_) + 0.1 * rng.standard_normal((y_, b_)), mFa)
    Fd = stack(np.eye(y_, x_) + 0.1 * rng.standard_normal((y_, x_)), mask(y_, x_, True))
    Va = stack(np.diag(rng.uniform(0.5, 1.5, b_)), np.eye(b_, dtype=bool))
    Vd = stack(np.diag(rng.uniform(0.5, 1.5, x_)) + 0.1 * rng.standard_normal((x_, x_)), mask(x_, x_, True))
    Wa = stack(np.diag(rng.uniform(0.5, 2.0, y_)), np.eye(y_, dtype=bool))
    Wd = stack(np.diag(rng.uniform(0.5, 2.0, y_)), np.eye(y_, dtype=bool))
    g = 0.5 * rng.standard_normal((b_, b_))
    Qb = g @ g.T + 0.1 * np.eye(b_)
    dgs = 1.0 + 0.05 * rng.standard_normal((B, T, b_))
    Q = Qb[None, None] * dgs[..., :, None] * dgs[..., None, :]                   # congruence: stays positive definite
    Rm = (0.5 + rng.random((B, T, u_, 1))) * np.eye(u_)[None, None]
    Qf = np.broadcast_to(Qb, (B, b_, b_)).copy()
    P = 0.05 * rng.standard_normal((B, T, u_, b_)) if use_P else np.zeros((B, T, u_, b_))
    zero = lambda *s: np.zeros(s)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)
    lay = (lambda a: workload.pack_systems(t(a))) if canon else t
    mz = lqg_amd.utils.mark_zero
    actor = lqg_amd.LQGSpec(Q=lay(Q), q=mz(t(zero(B, T, b_))), Qf=t(Qf), qf=mz(t(zero(B, b_))), P=(t(P) if use_P else mz(t(P))), R=lay(Rm),
                            r=mz(t(zero(B, T, u_))), A=lay(Aa), B=lay(Ba), V=lay(Va), F=lay(Fa), W=lay(Wa))
    zq = lambda *s: mz(t(zero(*s)))
    dyn = lqg_amd.LQGSpec(Q=zq(B, T, x_, x_), q=zq(B, T, x_), Qf=zq(B, x_, x_), qf=zq(B, x_), P=zq(B, T, u_, x_), R=zq(B, T, u_, u_),
                          r=zq(B, T, u_), A=lay(Ad), B=lay(Bd), V=lay(Vd), F=lay(Fd), W=lay(Wd))
    system = lqg_amd.System(actor=actor, dynamics=dyn)
    S0 = None
    if use_S0:
        h = 0.3 * rng.standard_normal((B, b_, b_))
        S0 = t(h @ np.swapaxes(h, -1, -2) + np.eye(b_))
    xs = t(np.cumsum(rng.standard_normal((B, n, T + 1, d_)), axis=2))
    with lqg_amd.options.override(TV_JIT_MIN_WORK=0):            # (small batches: compile their patterns anyway — that is what is fuzzed)
        plan = LogLikelihoodPlan(system, xs, Sigma0=S0)
        ll = plan.run().double().cpu().numpy()
    torch.cuda.synchronize()
    spec_ok = all(wk["specialised"] for wk in plan.work)
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    err = 0.0
    for j in sorted({int(v) for v in rng.integers(0, B, size=4)}):
        one = lambda sp: {f: (getattr(sp, f)[j] if getattr(sp, f).dim() == workload._batched_ndim(f) else getattr(sp, f)).double().cpu().numpy()
                          for f in sp._fields}
        ref = OC.log_likelihood(one(actor), one(dyn), xs[j].double().cpu().numpy(), None if S0 is None else S0[j].double().cpu().numpy())
        err = max(err, float(np.abs(ll[j] / ref - 1).max()))
    worst = max(worst, err / tol)
    print(f"case {case}: dims {(x_, b_, u_, y_, d_)} B={B} T={T} n={n} {str(dtype)[6:]} P={use_P} S0={use_S0} canonical={canon} "
          f"components={len(plan.work)} specialised={spec_ok} mixed={[bool(w['mixed']) for w in plan.work]} err={err:.2e} (tol {tol:g})", flush=True)
    assert spec_ok and err < tol, (case, err)
print("worst err / tol", worst)
