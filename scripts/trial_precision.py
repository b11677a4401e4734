"""Which part of the fp32 per-trial sweep carries the log-likelihood error at long horizons?  (round-2 review item 1)

CPU experiment (NumPy): the per-step trial operators (Fj - I_o, U2, Li, half-log-det; the layout k_trial reads) are built
from fp64 recursions of ONE system; the per-trial sweep (lqg_kernels_sp.hpp k_trial_sp, deviation form) then runs over n
trials in several precision mixes, all against the all-fp64 sweep on the SAME fp32-representable inputs:
  f32        operators rounded to fp32, state + arithmetic fp32 (the round-2 kernel)
  st64       operators rounded to fp32, state carried in fp64, mean update accumulated in fp64
  st64_w32   as st64, but the whitened innovation w and U2 w computed in fp32 (only the state recursion in fp64)
  comp       operators fp32, state as a two-float (hi, lo) pair, increments in fp32, TwoSum accumulate
  ops64      operators fp64, everything fp64, data fp32 (the floor set by the inputs = 0 by construction)
Usage: python scripts/trial_precision.py [bounded|hand] [n_trials] [T]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import lqg_np as NP


def bounded(T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, av=0.5, dt=1 / 60):
    A = np.eye(2); B = np.array([[0.0], [dt]]); F = np.eye(2)
    V = np.diag([1.0, av]); W = np.diag([sigma_target, sigma_cursor])
    Q = np.array([[1.0, -1.0], [-1.0, 1.0]]); R = np.array([[action_cost]])
    s = NP.time_stack_spec(A, B, F, V, W, Q, R, T)
    return s, s, 2


def hand(T, cursor_noise=0.1):
    dt, m, tau = 1.0 / 60.0, 1.0, 0.04
    A = np.zeros((5, 5)); A[0, 0] = 1.0
    A[1:, 1:] = [[1.0, dt, 0, 0], [0, 1.0, dt / m, 0], [0, 0, 1 - dt / tau, dt / tau], [0, 0, 0, 1 - dt / tau]]
    B = dt / tau * np.array([[0.0], [0], [0], [0], [1.0]]); F = np.eye(2, 5)
    V = np.diag([1.0, cursor_noise, 0, 0, 0.5]); W = np.diag([6.0, 6.0])
    Q = np.zeros((5, 5)); Q[:2, :2] = [[1, -1], [-1, 1]]; R = np.eye(1)
    s = NP.time_stack_spec(A, B, F, V, W, Q, R, T)
    return s, s, 2


def recursions(actor, dyn, R):
    """lqr.backward, kf.forward and the joint system of a TIME-INVARIANT model with every operation in precision R."""
    T = actor["A"].shape[0]
    A, B, F, V, W, Q, Rm = (actor[k][0].astype(R) for k in ("A", "B", "F", "V", "W", "Q", "R"))
    Ad, Bd, Fd, Vd, Wd = (dyn[k][0].astype(R) for k in ("A", "B", "F", "V", "W"))
    b, u = A.shape[0], B.shape[1]
    S = Q.copy(); Ls = [None] * T
    for t in range(T - 1, -1, -1):
        H = Rm + B.T @ S @ B; G = B.T @ S @ A
        ev = np.linalg.eigvalsh(H.astype(np.float64))[0]
        Ht = H + R(max(0.0, 1e-8 - ev)) * np.eye(u, dtype=R)
        L = -np.linalg.solve(Ht, G).astype(R)
        S = Q + A.T @ S @ A + L.T @ H @ L + L.T @ G + G.T @ L
        S = ((S + S.T) * R(0.5)).astype(R)
        Ls[t] = L
    P = V @ V.T; VV = V @ V.T; WW = W @ W.T
    Fj, Gj = [], []
    x = Ad.shape[0]
    for t in range(T):
        Pp = A @ P @ A.T + VV
        Gk = F @ Pp @ F.T + WW
        K = (Pp @ F.T @ np.linalg.inv(Gk).astype(R)).astype(R)
        P = Pp - K @ (F @ Pp)
        L = Ls[t]
        f = np.zeros((x + b, x + b), R); g = np.zeros((x + b, Vd.shape[1] + Wd.shape[1]), R)
        f[:x, :x] = Ad; f[:x, x:] = Bd @ L; f[x:, :x] = K @ Fd @ Ad
        f[x:, x:] = A + B @ L - K @ F @ A + K @ (Fd @ Bd - F @ B) @ L
        g[:x, :Vd.shape[1]] = Vd; g[x:, :Vd.shape[1]] = K @ Fd @ Vd; g[x:, Vd.shape[1]:] = K @ Wd
        Fj.append(f); Gj.append(g)
    return np.stack(Fj), np.stack(Gj)


def operators(actor, dyn, o, R=np.float64):
    Fj, Gj = recursions(actor, dyn, R)
    T, M = Fj.shape[0], Fj.shape[1]
    Sig = Gj[0] @ Gj[0].T
    ops = []
    for t in range(T + 1):
        Lc = np.linalg.cholesky(Sig[:o, :o]).astype(R); Li = np.linalg.inv(Lc).astype(R)
        U2 = Sig[o:, :o] @ Li.T
        hl = np.log(np.diag(Lc)).sum() + 0.5 * o * np.log(2 * np.pi)
        if t == T:
            ops.append((None, U2, Li, hl)); break
        Fd = Fj[t].copy(); Fd[:o, :o] -= np.eye(o, dtype=R)
        ops.append((Fd, U2, Li, hl))
        C = Sig[o:, o:] - U2 @ U2.T
        F2 = Fj[t][:, o:]
        Sig = F2 @ C @ F2.T + Gj[t] @ Gj[t].T
    return ops


def two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def sweep(ops, x, mode):
    n, T1, o = x.shape
    M = ops[0][0].shape[0]; RR = M - o
    f32 = np.float32
    OP = np.float64 if mode == "ops64" else f32
    ST = f32 if mode in ("f32", "comp", "dev32") else np.float64
    AR = f32 if mode in ("f32", "comp", "dev32", "dev64") else np.float64
    xprev = x[:, 0].astype(f32); dO = np.zeros((n, o), ST); muR = np.zeros((n, RR), ST)
    dOl = np.zeros((n, o), f32); muRl = np.zeros((n, RR), f32)
    acc = np.zeros(n); part = np.zeros(n, f32 if mode != "ops64" else np.float64)
    for t in range(T1):
        Fd, U2, Li, hl = ops[t]
        U2, Li = U2.astype(OP), Li.astype(OP)
        xt = x[:, t].astype(f32)
        dx = xt - xprev                                        # exact-ish difference of consecutive fp32 rows
        if mode == "comp":
            r = (dx - dO) - dOl
        else:
            r = dx.astype(ST) - dO
        WA = f32 if mode in ("f32", "comp", "st64_w32", "dev32", "dev64") else np.float64
        w = r.astype(WA) @ Li.astype(WA).T
        zz = (w * w).sum(1)
        if t > 0:
            part = part + (0.5 * zz + WA(hl)).astype(part.dtype)
        if t % 8 == 0 or t == T1 - 1:
            acc -= part.astype(np.float64); part[:] = 0
        if t == T1 - 1:
            break
        Fd = Fd.astype(OP)
        uw = w @ U2.astype(WA).T
        if mode == "comp":
            # c = muR + U2 w as a two-float; increments use the high part (+ low part through the small operator)
            ch, cl = two_sum(muR, uw)
            cl = cl + muRl
            cv = np.concatenate([xt, ch], 1)
            mn = cv @ Fd.T + np.concatenate([np.zeros((n, o), f32), cl], 1) @ Fd.T
            # unobserved rows: operator is F_rr itself (identity part inside): split as c + (F_rr - I) c
            Fr = Fd[o:].copy(); Fr[:, o:] -= np.eye(RR, dtype=f32)
            inc = cv @ Fr.T + np.concatenate([np.zeros((n, o), f32), cl], 1) @ Fr.T
            h, l = two_sum(ch, inc)
            muR, muRl = h, (l + cl).astype(f32)
            dO, dOl = mn[:, :o], np.zeros((n, o), f32)
        elif mode in ("dev64", "dev32"):
            # operator stream holds fl32(Fj - I) (identity subtracted BEFORE rounding); muR' = muR + (U2 w + (Fj - I)[r,:] cv)
            Fdev = ops[t][0].copy(); Fdev[o:, o:] -= np.eye(RR)
            Fdev = Fdev.astype(f32)
            c32 = (muR.astype(f32) + uw.astype(f32))
            cv = np.concatenate([xt, c32], 1)
            mn = cv @ Fdev.T
            dO = mn[:, :o].astype(dO.dtype)
            inc = uw.astype(f32) + mn[:, o:]
            muR = (muR + inc.astype(muR.dtype))
        else:
            c = muR + uw.astype(ST)
            cv = np.concatenate([xt.astype(AR), c.astype(AR)], 1)
            mn = cv @ Fd.astype(AR).T
            dO, muR = mn[:, :o].astype(ST), mn[:, o:].astype(ST)
        xprev = xt
    return acc


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "bounded"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 1067
    ncand = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rng = np.random.default_rng(3)
    lu = lambda lo, hi: float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    truth = (bounded if which == "bounded" else hand)(T)
    xd = truth[1]["A"].shape[1]; o = truth[2]
    X, _, _, _ = NP.simulate(truth[0], truth[1], rng.standard_normal((n, T, xd)), rng.standard_normal((n, T, truth[1]["F"].shape[1])))
    x = X[..., :o].astype(np.float32).astype(np.float64)
    worst = {}
    for c in range(ncand):
        if which == "bounded" and ncand > 1:
            par = dict(av=lu(0.1, 0.25), sigma_target=lu(1, 50), sigma_cursor=lu(1, 15), action_cost=lu(0.01, 0.05))
            actor, dyn, _ = bounded(T, **par)
        else:
            par = {}; actor, dyn = truth[0], truth[1]
        # the same fp32-representable inputs for every precision mix
        actor = {k: v.astype(np.float32).astype(np.float64) for k, v in actor.items()}; dyn = actor
        ops = operators(actor, dyn, o)
        ops32 = operators(actor, dyn, o, np.float32)
        ref = sweep(ops, x, "ops64")
        row = {}
        for mode in ("f32", "st64", "dev32", "dev64"):
            row[mode] = float(np.abs(sweep(ops, x, mode) / ref - 1).max())
        row["rec32_f32"] = float(np.abs(sweep(ops32, x, "f32") / ref - 1).max())
        row["rec32_st64"] = float(np.abs(sweep(ops32, x, "st64") / ref - 1).max())
        row["rec32_dev64"] = float(np.abs(sweep(ops32, x, "dev64") / ref - 1).max())
        print(json.dumps(dict(par=par, ll=float(np.abs(ref).mean()), **row)), flush=True)
        for k, v in row.items():
            worst[k] = max(worst.get(k, 0), v)
    print(json.dumps(dict(model=which, n=n, T=T, candidates=ncand, worst=worst)))


if __name__ == "__main__":
    main()
