"""The reverse-mode sweep of oracle/lqg_adjoint_np.py re-cut along the forward path's own split — TEST INFRASTRUCTURE ONLY.

CPU restatement of what the round-5 HIP adjoint kernels (lqg_amd/csrc/lqg_adjoint_sp.hpp) compute, in the same order and with
the same intermediate quantities.  Same mathematics as `lqg_adjoint_np.lane` (the round-1 sweep: one (system, trial) pair per
lane, every matrix adjoint repeated per trial), pinned against it in tests/test_adjoint.py; what changes is WHERE the work is:

  per system, once      backward Riccati (lqr.py:16-42), Kalman (kf.py:6-21), joint system + moment recursion
                        (system.py:167-230) -> the per-step TRIAL OPERATORS  Fj, U2 = S_ro Lc^-T, Li = chol(S_oo)^-1
  per trial             mean recursion + log-density (system.py:219-221, 244-248) over the operators       [forward]
                        mu-bar recursion over the same operators, emitting per step the TRIAL SUMS          [reverse]
                            g    = sum_n g_n
                            W2_t = sum_n g_n a_n(t+1) a_n(t+1)'          a_n(t) = S_oo(t)^-1 (x_t - mu_o(t)) = Li_t' w_n(t)
                            MC_t = sum_n post_n(t) c_n(t)'               post = mu-bar after the density of x_{t+1} was added
                            CA_t = sum_n ch_n(t) a_n(t)'                 ch = Fr' post  (Fr = Fj[:, o:])
  per system, once      adjoints of the moment recursion (Sigma-bar), the joint system, the Kalman step and — forward in time —
                        the Riccati recursion, fed by the trial sums: the matrix part of the path is data-independent and the
                        adjoint recursion is linear in (mu-bar, Sigma-bar), so Sigma-bar = sum_n Sigma-bar_n runs ONCE.

`loglik_grad` has the signature and the return values of lqg_adjoint_np.loglik_grad (time-invariant specs: pass [T, ...] stacks
of one matrix; the bars come back summed over time in slot 0 of a [1, ...] stack when `ti=True`).
"""
import numpy as np

sym = lambda M: 0.5 * (M + M.T)
gram = lambda V: V @ V.T


def riccati_step(S, Q, R, A, B, eps):
    H = R + B.T @ S @ B
    G = B.T @ S @ A
    Hti = np.linalg.inv(H + max(0.0, eps - np.linalg.eigvalsh(H)[0]) * np.eye(H.shape[0]))
    L = -Hti @ G
    return H, G, Hti, L, Q + A.T @ S @ A + L.T @ H @ L + L.T @ G + G.T @ L


def system_step(Sig, P0, L, sp, o):
    """Data-independent part of one forward step from the state before it (P_t, Sigma_t)."""
    Ad, Bd, Fd, VVd, WWd, Aa, Ba, Fa, VVa, WWa = sp
    Pp = Aa @ P0 @ Aa.T + VVa
    FPp = Fa @ Pp
    Gi = np.linalg.inv(FPp @ Fa.T + WWa)
    K = FPp.T @ Gi
    Y = K @ Fd
    D = Fd @ Bd - Fa @ Ba
    Z = D @ L - Fa @ Aa
    F = np.block([[Ad, Bd @ L], [Y @ Ad, Aa + Ba @ L + K @ Z]])
    YVV, KWW = Y @ VVd, K @ WWd
    GG = np.block([[VVd, YVV.T], [YVV, YVV @ Y.T + KWW @ K.T]])
    if Sig is None:
        Sig = GG
    Lc = np.linalg.cholesky(Sig[:o, :o])
    Li = np.linalg.inv(Lc)
    U2 = Sig[o:, :o] @ Li.T
    Crr = Sig[o:, o:] - U2 @ U2.T
    Fr = F[:, o:]
    FCr = Fr @ Crr
    return dict(Pp=Pp, FPp=FPp, Gi=Gi, K=K, Y=Y, D=D, Z=Z, F=F, YVV=YVV, KWW=KWW, GG=GG, Sig=Sig, Li=Li, U2=U2, Crr=Crr, Fr=Fr,
                FCr=FCr, Sig1=sym(FCr @ Fr.T + GG), P1=sym(Pp - K @ FPp))


def loglik_grad(actor, dyn, x, g=None, Sigma0=None, eps=1e-8):
    x = np.asarray(x, dtype=np.float64)
    n, T, o = x.shape[0], x.shape[1] - 1, x.shape[2]
    g = np.ones(n) if g is None else np.asarray(g, dtype=np.float64)
    xd, b = dyn["A"].shape[1], actor["A"].shape[1]
    m, u = xd + b, actor["B"].shape[2]
    ny = actor["F"].shape[1]
    sp = lambda t: (dyn["A"][t], dyn["B"][t], dyn["F"][t], gram(dyn["V"][t]), gram(dyn["W"][t]),
                    actor["A"][t], actor["B"][t], actor["F"][t], gram(actor["V"][t]), gram(actor["W"][t]))
    ric = lambda S, t: riccati_step(S, actor["Q"][t], actor["R"][t], actor["A"][t], actor["B"][t], eps)

    # ---- per system: Riccati backward (keeps S_{t+1}, L_t), forward sweep (keeps P_t, Sigma_t; emits the trial operators)
    S = actor["Qf"]
    Snext, Ls = [None] * T, [None] * T
    for t in range(T - 1, -1, -1):
        Snext[t] = S
        _, _, _, Ls[t], S = ric(S, t)
    P = gram(actor["V"][0]) if Sigma0 is None else np.asarray(Sigma0, dtype=np.float64)
    Sig = None
    Ps, Sigs, ops = [None] * T, [None] * T, [None] * (T + 1)
    for t in range(T):
        f = system_step(Sig, P, Ls[t], sp(t), o)
        Ps[t], Sigs[t] = P, f["Sig"]
        ops[t] = (f["F"], f["U2"], f["Li"])
        P, Sig = f["P1"], f["Sig1"]
    LiT = np.linalg.inv(np.linalg.cholesky(Sig[:o, :o]))
    ops[T] = (None, None, LiT)

    # ---- per trial, forward: mean recursion and log-density over the operators; keeps w_t, c_t
    ll = np.zeros(n)
    ws = np.zeros((n, T + 1, o))
    cs = np.zeros((n, T, m))
    for i in range(n):
        mu = np.concatenate([x[i, 0], np.zeros(m - o)])
        for t in range(T + 1):
            F, U2, Li = ops[t]
            w = Li @ (x[i, t] - mu[:o])
            ws[i, t] = w
            if t > 0:
                ll[i] += -0.5 * (w @ w) + np.log(np.diag(Li)).sum() - 0.5 * o * np.log(2 * np.pi)
            if t < T:
                c = np.concatenate([x[i, t], mu[o:] + U2 @ w])
                cs[i, t] = c
                mu = F @ c

    # ---- per trial, reverse: mu-bar recursion, emits the trial sums per step
    G0 = g.sum()
    W2 = np.zeros((T, o, o))
    MC = np.zeros((T, m, m))
    CA = np.zeros((T, m - o, o))
    for i in range(n):
        pre = np.zeros(m)
        for t in range(T - 1, -1, -1):
            F, U2, Li = ops[t]
            a1 = ops[t + 1][2].T @ ws[i, t + 1]                       # a_n(t+1)
            a0 = Li.T @ ws[i, t]                                      # a_n(t)
            post = pre.copy()
            post[:o] += g[i] * a1
            W2[t] += g[i] * np.outer(a1, a1)
            MC[t] += np.outer(post, cs[i, t])
            ch = F[:, o:].T @ post
            CA[t] += np.outer(ch, a0)
            Wm = U2 @ Li
            pre = np.concatenate([-Wm.T @ ch, ch])

    # ---- per system, reverse, in TWO backward sweeps as the round-6 kernels run them (the Kalman-covariance adjoint reads nothing
    # of the moment recursion but K-bar_t): (i) Sigma-bar recursion, joint system and the bars they own, fed by the trial sums,
    # leaving K-bar_t (without its P-bar term) and L-bar_t per step [k_asp_sys_rev]; (ii) the P-bar recursion over K-bar_t
    # [k_asp_kal_rev]; then (iii), forward in time, the Riccati adjoint over L-bar_t [k_asp_ric_rev]
    names = ("dA", "dB", "dF", "dVV", "dWW", "aA", "aB", "aF", "aVV", "aWW", "aQ", "aR")
    shapes = dict(dA=(xd, xd), dB=(xd, u), dF=(ny, xd), dVV=(xd, xd), dWW=(ny, ny), aA=(b, b), aB=(b, u), aF=(ny, b),
                  aVV=(b, b), aWW=(ny, ny), aQ=(b, b), aR=(u, u))
    bar = {k: np.zeros((T,) + shapes[k]) for k in names}
    Lbar, Kbar = [None] * T, [None] * T
    Sigb = np.zeros((m, m))
    for t in range(T - 1, -1, -1):
        Ad, Bd, Fd, VVd, WWd, Aa, Ba, Fa, VVa, WWa = sp(t)
        L = Ls[t]
        f = system_step(Sigs[t], Ps[t], L, sp(t), o)
        F, K, Y, D, Z, Fr = f["F"], f["K"], f["Y"], f["D"], f["Z"], f["Fr"]
        Li1 = ops[t + 1][2]
        Sigb[:o, :o] += 0.5 * (W2[t] - G0 * (Li1.T @ Li1))            # log-density of x[t+1], all trials
        Fb = MC[t].copy()
        Fb[:, o:] += 2.0 * Sigb @ f["FCr"]
        GGb = Sigb.copy()
        Wm = f["U2"] @ f["Li"]
        Ch = sym(Fr.T @ Sigb @ Fr)
        ChW = Ch @ Wm
        Sro = CA[t] - 2.0 * ChW
        Soo = Wm.T @ ChW - Wm.T @ CA[t]
        Sigb = np.block([[sym(Soo), 0.5 * Sro.T], [0.5 * Sro, Ch]])
        if t == 0:
            GGb = GGb + Sigb
        F11, F12, F21, F22 = Fb[:xd, :xd], Fb[:xd, xd:], Fb[xd:, :xd], Fb[xd:, xd:]
        G11, G21, G22 = GGb[:xd, :xd], GGb[xd:, :xd], GGb[xd:, xd:]
        Yb = F21 @ Ad.T + 2.0 * G21 @ VVd + 2.0 * G22 @ f["YVV"]
        KtF22 = K.T @ F22
        Kb = Yb @ Fd.T + F22 @ Z.T + 2.0 * G22 @ f["KWW"]
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
        Kbar[t] = Kb
        bar["aA"][t], bar["aB"][t], bar["aF"][t] = aA, aB, aF
    Pb = np.zeros((b, b))
    for t in range(T - 1, -1, -1):
        Ad, Bd, Fd, VVd, WWd, Aa, Ba, Fa, VVa, WWa = sp(t)
        f = system_step(Sigs[t], Ps[t], Ls[t], sp(t), o)
        K, Pp, Gi, FPp = f["K"], f["Pp"], f["Gi"], f["FPp"]
        aA, aF = bar["aA"][t], bar["aF"][t]
        Kb = Kbar[t] - Pb @ FPp.T
        KbGi = Kb @ Gi
        Ppb = Pb - (K @ Fa).T @ Pb + KbGi @ Fa
        aF += (KbGi - Pb @ K).T @ Pp
        Gmb = -Gi @ (FPp @ KbGi)
        Ppb = sym(Ppb + Fa.T @ Gmb @ Fa)
        aF += (Gmb + Gmb.T) @ FPp
        bar["aWW"][t] = Gmb
        bar["aVV"][t] = Ppb
        aA += 2.0 * Ppb @ Aa @ Ps[t]
        bar["aA"][t], bar["aF"][t] = aA, aF
        Pb = sym(Aa.T @ Ppb @ Aa)
    S0b = Pb
    Sb = np.zeros((b, b))
    for t in range(T):
        A, B = actor["A"][t], actor["B"][t]
        S = Snext[t]
        H, G, Hti, L, _ = ric(S, t)
        bar["aQ"][t] = Sb
        Lb = Lbar[t] + 2.0 * (H @ L + G) @ Sb
        Gb = 2.0 * L @ Sb - Hti @ Lb
        Hb = L @ Sb @ L.T - Hti @ Lb @ L.T
        bar["aR"][t] = Hb
        SA, SB = S @ A, S @ B
        bar["aA"][t] += 2.0 * SA @ Sb + SB @ Gb
        bar["aB"][t] += SA @ Gb.T + SB @ (Hb + Hb.T)
        Sb = sym(A @ Sb @ A.T + B @ Gb @ A.T + B @ Hb @ B.T)
    tot = bar
    ga = {"A": tot["aA"], "B": tot["aB"], "F": tot["aF"], "Q": tot["aQ"], "R": tot["aR"], "Qf": Sb,
          "V": np.einsum("tij,tjk->tik", tot["aVV"] + np.swapaxes(tot["aVV"], 1, 2), actor["V"]),
          "W": np.einsum("tij,tjk->tik", tot["aWW"] + np.swapaxes(tot["aWW"], 1, 2), actor["W"])}
    gd = {"A": tot["dA"], "B": tot["dB"], "F": tot["dF"],
          "V": np.einsum("tij,tjk->tik", tot["dVV"] + np.swapaxes(tot["dVV"], 1, 2), dyn["V"]),
          "W": np.einsum("tij,tjk->tik", tot["dWW"] + np.swapaxes(tot["dWW"], 1, 2), dyn["W"])}
    if Sigma0 is None:
        ga["V"][0] += 2.0 * S0b @ actor["V"][0]
    return ll, ga, gd, S0b, dict(W2=W2, MC=MC, CA=CA, g=G0)
