"""Which rounding drives the fp32 point-mass error at T = 1067?  A NumPy emulation of the fp32 per-trial sweep
(deviation form of the operator stream, include/lqg_hip.h ABI 2) over operators built in fp64 from the split restatement
(oracle/lqg_adjoint_split_np.py — test infrastructure, used here as a study tool only), with the operator applied three ways:

  f32    Fj - I rounded once to fp32 (the MIXED mode's stream before round 5)
  hilo   hi = fl32(F), lo = fl32(F - hi), mn = hi cv + lo cv      (what k_trial_sp<..., HL> computes)
  ops64  exact operators and products, fp32 state               (the floor of an fp32 state)

Error relative to max(|ll|, T d) against the same sweep in fp64.  Runs on the CPU (minutes):
    python scripts/pointmass_hilo_emulation.py [PointMassBoundedActor|BoundedActor]
Recorded (24 candidates of the bench's ranges, seed 5, synthetic tracking data):
    PointMassBoundedActor  f32 max 1.19e-6 median 7.5e-8 | hilo max 1.20e-7 | ops64 max 1.07e-7   max|F - I| 5.5 .. 74
    BoundedActor           f32 max 1.08e-7 median 8.5e-8                                           max|F - I| 0.09 .. 0.6"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lqg_amd                                   # noqa: E402
from lqg_amd import workload                     # noqa: E402
import lqg_adjoint_split_np as SP                # noqa: E402



def gram(V):
    return V @ V.T


def build_ops(m, d):
    """fp64 operator stream [(Fj - I, U2, Li)] x T, then (None, None, Li_T), of a time-invariant-or-not model m."""
    T = m.T
    a, dy = ({f: getattr(s, f).double().cpu().numpy() for f in s._fields} for s in (m.actor, m.dynamics))
    S, Ls = a["Qf"], [None] * T
    for t in range(T - 1, -1, -1):
        _, _, _, Ls[t], S = SP.riccati_step(S, a["Q"][t], a["R"][t], a["A"][t], a["B"][t], 1e-8)
    P, Sig, ops = gram(a["V"][0]), None, []
    for t in range(T):
        f = SP.system_step(Sig, P, Ls[t], (dy["A"][t], dy["B"][t], dy["F"][t], gram(dy["V"][t]), gram(dy["W"][t]),
                                           a["A"][t], a["B"][t], a["F"][t], gram(a["V"][t]), gram(a["W"][t])), d)
        ops.append((f["F"] - np.eye(f["F"].shape[0]), f["U2"], f["Li"]))
        P, Sig = f["P1"], f["Sig1"]
    ops.append((None, None, np.linalg.inv(np.linalg.cholesky(Sig[:d, :d]))))
    return ops


def sweep(ops, x, mode):
    """log-likelihoods of the trials x[n, T+1, d] under the deviation-form sweep; mode f64 | f32 | hilo | ops64."""
    T, d, M = len(ops) - 1, x.shape[-1], ops[0][0].shape[0]
    R = np.float64 if mode == "f64" else np.float32
    out = []
    for i in range(x.shape[0]):
        dO, muR, xprev, ll = np.zeros(d, R), np.zeros(M - d, R), x[i, 0].astype(R), 0.0
        for t in range(T + 1):
            Fm, U2, Li = ops[t]
            xt = x[i, t].astype(R)
            w = Li.astype(R) @ ((xt - xprev) - dO)
            if t > 0:
                ll += float(-0.5 * (w @ w)) + float(np.log(np.diag(Li)).sum()) - 0.5 * d * np.log(2 * np.pi)
            if t < T:
                cr = muR + U2.astype(R) @ w
                cv = np.concatenate([xt, cr]).astype(R)
                if mode == "f32":
                    mn = Fm.astype(np.float32) @ cv
                elif mode == "hilo":
                    hi = Fm.astype(np.float32)
                    mn = hi @ cv + (Fm - hi.astype(np.float64)).astype(np.float32) @ cv
                elif mode == "ops64":
                    mn = (Fm @ cv.astype(np.float64)).astype(np.float32)
                else:
                    mn = Fm @ cv
                dO, muR, xprev = mn[:d].astype(R), (cr + mn[d:]).astype(R), xt
        out.append(ll)
    return np.array(out)


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "PointMassBoundedActor"
    T, n, d, B = 1067, 3, 2, 24
    dev = torch.device("cpu")
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost")
    kw = {k: workload.log_uniform(B, *workload.RANGES[k], gen, dev, torch.float32) for k in names}
    rng = np.random.default_rng(3)
    tgt = np.cumsum(rng.standard_normal((n, T + 1)), axis=1) + 50.0
    cur = tgt + np.cumsum(rng.standard_normal((n, T + 1)) * 0.3, axis=1) * 0.2 + rng.standard_normal((n, T + 1)) * 0.5
    x = np.stack([tgt, cur], -1)
    res = []
    for c in range(B):
        m = getattr(lqg_amd, model)(T=T, device=dev, dtype=torch.float32, **{k: float(v[c]) for k, v in kw.items()})
        ops = build_ops(m, d)
        fmax = max(np.abs(o[0]).max() for o in ops[:-1])
        ref = sweep(ops, x, "f64")
        sc = np.maximum(np.abs(ref), T * d)
        e = {k: float((np.abs(sweep(ops, x, k) - ref) / sc).max()) for k in ("f32", "hilo", "ops64")}
        res.append(e)
        print(c, {k: "%.2e" % v for k, v in e.items()}, "max|F - I| %.2f" % fmax, flush=True)
    for k in ("f32", "hilo", "ops64"):
        print(k, "max %.2e median %.2e" % (max(r[k] for r in res), np.median([r[k] for r in res])))


if __name__ == "__main__":
    main()
