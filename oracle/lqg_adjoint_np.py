"""Hand-derived reverse-mode sweep for the LQG log-likelihood in NumPy — TEST INFRASTRUCTURE ONLY.

CPU restatement of what the HIP adjoint kernels (lqg_amd/csrc/lqg_adjoint.hpp) compute, in the same order and with the
same intermediate quantities: d(sum_n g[n] ll[n]) / d(spec matrices) through all three scans of the path
(lqg/control/lqr.py:16-42 backward Riccati, lqg/belief/kf.py:6-21 Kalman, lqg/system.py:142-248 moment recursion +
Gaussian log-density) — what the reference gets from jax.grad (lqg/optim.py:142-147).  Pinned in tests/test_adjoint.py
against torch.autograd of oracle/lqg_torch_ref.py (itself equal to the golden vectors) and against central finite
differences of the C oracle.

Structure (one "lane" = one (system, trial) pair, as in the kernels):
  forward    Riccati keeps S_{t+1}, L_t; the forward sweep keeps the state BEFORE each step: P_t, Sigma_t, mu_t
  reverse    t = T-1..0: recompute the step from the kept state, then the adjoints of the log-density, the moment
             recursion, the joint system (-> Lbar_t, Kbar_t and the spec bars) and the Kalman step
  riccati    t = 0..T-1 (the recursion ran backward): consumes Lbar_t
Conventions: noise enters through the Gram matrices VV = V V', WW = W W' (bars are chained to V, W at the end); adjoints
of symmetric carries (Sigma, P, S) are symmetrised every step — without that the antisymmetric "gauge" part, which no
symmetric perturbation can see, grows geometrically and destroys the result by cancellation (observed at T = 100).
Bars of symmetric inputs (Q, Qf, R-block, Sigma0) are therefore the symmetrised gradients.  The eigenvalue-floor shift
(lqr.py:27-28) is held constant.  q, r, qf, P get no gradients (the likelihood ignores the affine gain l; P is zero in
every model).  Returns per-step bars [T, ...]; for a time-invariant spec sum axis 0.
"""
import numpy as np

sym = lambda M: 0.5 * (M + M.T)


def forward_step(Sig, mu, P0, L, sp, xt, t):
    """Everything the step computes from the state before it (shared by the forward and the reverse sweep)."""
    Ad, Bd, Fd, VVd, WWd, Aa, Ba, Fa, VVa, WWa = sp(t)
    xd, o = Ad.shape[0], xt.shape[0]
    Pp = Aa @ P0 @ Aa.T + VVa                                        # kf.py:10
    Gi = np.linalg.inv(Fa @ Pp @ Fa.T + WWa)                         # kf.py:11
    K = Pp @ Fa.T @ Gi                                               # kf.py:12
    Y = K @ Fd
    D = Fd @ Bd - Fa @ Ba
    F = np.block([[Ad, Bd @ L], [Y @ Ad, Aa + Ba @ L - K @ Fa @ Aa + K @ D @ L]])      # system.py:167-181
    GG = np.block([[VVd, VVd @ Y.T], [Y @ VVd, Y @ VVd @ Y.T + K @ WWd @ K.T]])        # G_j G_j'  system.py:194-202
    if Sig is None:
        Sig = GG                                                     # system.py:212
    m = Sig.shape[0]
    N = np.linalg.inv(Sig[:o, :o])
    r = xt - mu[:o]
    a = N @ r
    Wm = Sig[o:, :o] @ N                                             # regression of the rest on the observed dims
    c = np.concatenate([xt, mu[o:] + Wm @ r])                        # conditional mean     system.py:219-221
    C = np.zeros((m, m))
    C[o:, o:] = Sig[o:, o:] - Wm @ Sig[:o, o:]                       # conditional covariance  system.py:223-230
    FC = F @ C
    return dict(Pp=Pp, Gi=Gi, K=K, Y=Y, D=D, F=F, GG=GG, Sig=Sig, Wm=Wm, N=N, r=r, a=a, c=c, C=C, FC=FC,
                mu1=F @ c, Sig1=FC @ F.T + GG, P1=Pp - K @ (Fa @ Pp))               # kf.py:14


def lane(actor, dyn, x, g, Sigma0=None, eps=1e-8):
    """One (system, trial) pair: x[T+1, o], weight g -> ll, bars (dict name -> [T, ...]), Sigma0 bar."""
    T, o = x.shape[0] - 1, x.shape[1]
    xd, b = dyn["A"].shape[1], actor["A"].shape[1]
    m, u = xd + b, actor["B"].shape[2]
    gram = lambda V: V @ V.T

    def sp(t):
        return (dyn["A"][t], dyn["B"][t], dyn["F"][t], gram(dyn["V"][t]), gram(dyn["W"][t]),
                actor["A"][t], actor["B"][t], actor["F"][t], gram(actor["V"][t]), gram(actor["W"][t]))

    def riccati_step(S, t):
        Q, P, R, A, B = (actor[k][t] for k in ("Q", "P", "R", "A", "B"))
        H = R + B.T @ S @ B
        G = P + B.T @ S @ A
        Hti = np.linalg.inv(H + max(0.0, eps - np.linalg.eigvalsh(H)[0]) * np.eye(u))
        L = -Hti @ G
        return H, G, Hti, L, Q + A.T @ S @ A + L.T @ H @ L + L.T @ G + G.T @ L

    # ---- kernel A: Riccati backward, keep S_{t+1} and L_t
    S = actor["Qf"]
    Snext, Ls = [None] * T, [None] * T
    for t in range(T - 1, -1, -1):
        Snext[t] = S
        _, _, _, Ls[t], S = riccati_step(S, t)
    # ---- kernel B: forward sweep, keep the state before each step
    P = gram(actor["V"][0]) if Sigma0 is None else np.asarray(Sigma0, dtype=np.float64)
    mu, Sig = np.concatenate([x[0], np.zeros(m - o)]), None
    Ps, Sigs, mus = [None] * T, [None] * T, [None] * T
    ll = 0.0
    for t in range(T):
        f = forward_step(Sig, mu, P, Ls[t], sp, x[t], t)
        Ps[t], Sigs[t], mus[t] = P, f["Sig"], mu
        P, Sig, mu = f["P1"], f["Sig1"], f["mu1"]
        e = x[t + 1] - mu[:o]
        ll += -0.5 * (o * np.log(2 * np.pi) + np.linalg.slogdet(Sig[:o, :o])[1] + e @ np.linalg.solve(Sig[:o, :o], e))
    # ---- kernel C: reverse sweep
    names = ("dA", "dB", "dF", "dVV", "dWW", "aA", "aB", "aF", "aVV", "aWW", "aQ", "aR")
    shapes = dict(dA=(xd, xd), dB=(xd, u), dF=(dyn["F"].shape[1], xd), dVV=(xd, xd), dWW=(dyn["F"].shape[1],) * 2,
                  aA=(b, b), aB=(b, u), aF=(actor["F"].shape[1], b), aVV=(b, b), aWW=(actor["F"].shape[1],) * 2,
                  aQ=(b, b), aR=(u, u))
    bar = {k: np.zeros((T,) + shapes[k]) for k in names}
    Lbar = [None] * T
    mub, Sigb, Pb = np.zeros(m), np.zeros((m, m)), np.zeros((b, b))
    for t in range(T - 1, -1, -1):
        Ad, Bd, Fd, VVd, WWd, Aa, Ba, Fa, VVa, WWa = sp(t)
        L = Ls[t]
        f = forward_step(Sigs[t], mus[t], Ps[t], L, sp, x[t], t)
        F, K, Y, D = f["F"], f["K"], f["Y"], f["D"]
        # log-density of x[t+1]                                           system.py:244-248
        Ni = np.linalg.inv(f["Sig1"][:o, :o])
        w = Ni @ (x[t + 1] - f["mu1"][:o])
        mub[:o] += g * w
        Sigb[:o, :o] += 0.5 * g * (np.outer(w, w) - Ni)
        # Sig1 = F C F' + GG ; mu1 = F c
        Fb = 2.0 * Sigb @ f["FC"] + np.outer(mub, f["c"])
        GGb = Sigb
        # only the unobserved block of C and c carries information (the observed rows are 0 and x_t exactly): written
        # in terms of Wm = S_ro S_oo^-1 and a = S_oo^-1 r, no product of two inverses appears — the general formula
        # cancels O(cond(S_oo)^2) terms and is useless for the point-mass model
        Fr = F[:, o:]
        Ch = Fr.T @ Sigb @ Fr
        ch = Fr.T @ mub
        Wm, a = f["Wm"], f["a"]
        Wtc = Wm.T @ ch
        Sro = np.outer(ch, a) - 2.0 * Ch @ Wm
        Soo = Wm.T @ Ch @ Wm - np.outer(Wtc, a)
        mub = np.concatenate([-Wtc, ch])
        Sigb = np.block([[sym(Soo), 0.5 * Sro.T], [0.5 * Sro, Ch]])
        if t == 0:
            GGb = GGb + Sigb                                              # Sigma_0 = G_0 G_0'
        # joint system -> spec bars, Lbar, Kbar
        F11, F12, F21, F22 = Fb[:xd, :xd], Fb[:xd, xd:], Fb[xd:, :xd], Fb[xd:, xd:]
        G11, G21, G22 = GGb[:xd, :xd], GGb[xd:, :xd], GGb[xd:, xd:]
        Yb = F21 @ Ad.T + 2.0 * G21 @ VVd + 2.0 * G22 @ Y @ VVd
        KtF22 = K.T @ F22
        Kb = Yb @ Fd.T - F22 @ (Fa @ Aa).T + F22 @ (D @ L).T + 2.0 * G22 @ K @ WWd
        Db = KtF22 @ L.T
        bar["dA"][t] = F11 + Y.T @ F21
        bar["dB"][t] = F12 @ L.T + Fd.T @ Db
        bar["dF"][t] = K.T @ Yb + Db @ Bd.T
        bar["dVV"][t] = G11 + 2.0 * Y.T @ G21 + Y.T @ G22 @ Y
        bar["dWW"][t] = K.T @ G22 @ K
        aA = F22 - Fa.T @ KtF22
        aB = F22 @ L.T - Fa.T @ Db
        aF = -KtF22 @ Aa.T - Db @ Ba.T
        Lbar[t] = Bd.T @ F12 + Ba.T @ F22 + D.T @ KtF22
        # Kalman step adjoint                                             kf.py:10-14
        Pp, Gi = f["Pp"], f["Gi"]
        FPp = Fa @ Pp
        Kb = Kb - Pb @ FPp.T
        Ppb = Pb - (K @ Fa).T @ Pb + Kb @ Gi @ Fa
        aF += -K.T @ Pb @ Pp + Gi @ Kb.T @ Pp
        Gmb = -Gi @ (FPp @ Kb) @ Gi
        Ppb += Fa.T @ Gmb @ Fa
        Ppb = sym(Ppb)
        aF += (Gmb + Gmb.T) @ FPp
        bar["aWW"][t] = Gmb
        bar["aVV"][t] = Ppb
        aA += 2.0 * Ppb @ Aa @ Ps[t]
        bar["aA"][t], bar["aB"][t], bar["aF"][t] = aA, aB, aF
        Pb = sym(Aa.T @ Ppb @ Aa)
    Sigma0_bar = Pb
    # ---- kernel D: Riccati adjoint, forward in time
    Sb = np.zeros((b, b))
    for t in range(T):
        A, B = actor["A"][t], actor["B"][t]
        S = Snext[t]
        H, G, Hti, L, _ = riccati_step(S, t)
        bar["aQ"][t] = Sb
        Lb = Lbar[t] + 2.0 * (H @ L + G) @ Sb
        Gb = 2.0 * L @ Sb - Hti @ Lb
        Hb = L @ Sb @ L.T - Hti @ Lb @ L.T
        bar["aR"][t] = Hb
        SA, SB = S @ A, S @ B
        bar["aA"][t] += 2.0 * SA @ Sb + SB @ Gb
        bar["aB"][t] += SA @ Gb.T + SB @ (Hb + Hb.T)
        Sb = sym(A @ Sb @ A.T + B @ Gb @ A.T + B @ Hb @ B.T)
    return ll, bar, Sb, Sigma0_bar


def loglik_grad(actor, dyn, x, g=None, Sigma0=None, eps=1e-8):
    """x[n, T+1, o], weights g[n] -> ll[n], actor bars, dynamics bars (keyed like LQGSpec, per step), Sigma0 bar."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    g = np.ones(n) if g is None else np.asarray(g, dtype=np.float64)
    ll = np.zeros(n)
    tot, Qfb, S0b = None, 0.0, 0.0
    for i in range(n):
        ll[i], bar, qf, s0 = lane(actor, dyn, x[i], g[i], Sigma0, eps)
        tot = bar if tot is None else {k: tot[k] + bar[k] for k in bar}
        Qfb, S0b = Qfb + qf, S0b + s0
    ga = {"A": tot["aA"], "B": tot["aB"], "F": tot["aF"], "Q": tot["aQ"], "R": tot["aR"], "Qf": Qfb,
          "V": np.einsum("tij,tjk->tik", tot["aVV"] + np.swapaxes(tot["aVV"], 1, 2), actor["V"]),
          "W": np.einsum("tij,tjk->tik", tot["aWW"] + np.swapaxes(tot["aWW"], 1, 2), actor["W"])}
    gd = {"A": tot["dA"], "B": tot["dB"], "F": tot["dF"],
          "V": np.einsum("tij,tjk->tik", tot["dVV"] + np.swapaxes(tot["dVV"], 1, 2), dyn["V"]),
          "W": np.einsum("tij,tjk->tik", tot["dWW"] + np.swapaxes(tot["dWW"], 1, 2), dyn["W"])}
    if Sigma0 is None:
        ga["V"][0] += 2.0 * S0b @ actor["V"][0]
    return ll, ga, gd, S0b
