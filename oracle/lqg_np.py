"""NumPy (fp64) restatement of the reference's LQG solve path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under lqg_amd/ does.  It is the *checker*, never the product path.

Parity status: PINNED against the reference's own source executed in the build container under
oracle/jax_standin.py (NumPy/LAPACK fp64 in place of XLA's LAPACK calls; golden vectors in
tests/golden/, generator oracle/gen_golden.py), plus reference-independent analytic pins
(closed-form K[0], L[T-1]; brute-force joint Gaussian; DARE steady state) in tests/test_oracle.py.
The reference's tests hold no golden numbers, so there are none to check beyond these.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Specs are plain dicts of (T, ...) stacked arrays with the LQGSpec field names (lqg/spec.py:5-19).
"""
import numpy as np

FIELDS = ("Q", "q", "Qf", "qf", "P", "R", "r", "A", "B", "V", "F", "W")


def time_stack_spec(A, B, F, V, W, Q, R, T):
    """lqg/utils.py:10-35: replicate the 7 matrices T times; q, P, r = 0; Qf = Q[-1]; qf = 0."""
    A, B, F, V, W, Q, R = (np.asarray(m, dtype=np.float64) for m in (A, B, F, V, W, Q, R))
    rep = lambda m: np.broadcast_to(m, (T,) + m.shape).copy()
    b, u = Q.shape[0], R.shape[0]
    return dict(A=rep(A), B=rep(B), F=rep(F), V=rep(V), W=rep(W), Q=rep(Q), R=rep(R),
                q=np.zeros((T, b)), Qf=Q.copy(), qf=np.zeros(b),
                P=np.zeros((T, u, b)), r=np.zeros((T, u)))


def dynamics_spec(A, B, F, V, W, T):
    """lqg/system.py:331-344: Dynamics() zero-fills Q and R."""
    x, u = np.shape(A)[0], np.shape(B)[1]
    return time_stack_spec(A, B, F, V, W, np.zeros((x, x)), np.zeros((u, u)), T)


def backward(spec, eps=1e-8):
    """lqg/control/lqr.py:16-42: finite-horizon Riccati recursion with affine terms.

    Returns L[T,u,b], l[T,u], H[T,u,u] (H is the *regularised* Ht, lqr.py:36,42) in forward time order."""
    T = spec["A"].shape[0]
    u, b = spec["B"].shape[2], spec["A"].shape[1]
    S, s = spec["Qf"].copy(), spec["qf"].copy()                       # lqr.py:38
    L, l, Hs = np.empty((T, u, b)), np.empty((T, u)), np.empty((T, u, u))
    for t in range(T - 1, -1, -1):                                    # reverse=True, lqr.py:40
        Q, q, P, R, r, A, B = (spec[k][t] for k in ("Q", "q", "P", "R", "r", "A", "B"))
        H = R + B.T @ S @ B                                           # lqr.py:22
        G = P + B.T @ S @ A                                           # lqr.py:23
        g = r + B.T @ s                                               # lqr.py:24
        evals = np.linalg.eigvalsh(H)                                 # lqr.py:27
        Ht = H + max(0.0, eps - evals[0]) * np.eye(u)                 # lqr.py:28
        Lt = -np.linalg.solve(Ht, G)                                  # lqr.py:30
        lt = -np.linalg.solve(Ht, g)                                  # lqr.py:31
        S_new = Q + A.T @ S @ A + Lt.T @ H @ Lt + Lt.T @ G + G.T @ Lt  # lqr.py:33 (unregularised H)
        s = q + A.T @ s + G.T @ lt + Lt.T @ H @ lt + Lt.T @ g         # lqr.py:34
        S = S_new
        L[t], l[t], Hs[t] = Lt, lt, Ht
    return L, l, Hs


def forward(spec, Sigma0):
    """lqg/belief/kf.py:6-21: Kalman gain recursion, explicit inverse, non-Joseph update. K[T,b,y]."""
    T = spec["A"].shape[0]
    b, y = spec["A"].shape[1], spec["F"].shape[1]
    P = np.array(Sigma0, dtype=np.float64)
    K = np.empty((T, b, y))
    for t in range(T):
        A, F, V, W = (spec[k][t] for k in ("A", "F", "V", "W"))
        P = A @ P @ A.T + V @ V.T                                     # kf.py:10
        G = F @ P @ F.T + W @ W.T                                     # kf.py:11
        K[t] = P @ F.T @ np.linalg.inv(G)                             # kf.py:12
        P = (np.eye(b) - K[t] @ F) @ P                                # kf.py:14
    return K


def default_sigma0(actor):
    """lqg/system.py:79,160: Sigma0 defaults to actor.V[0] @ actor.V[0].T."""
    return actor["V"][0] @ actor["V"][0].T


def joint_system(actor, dyn, L, K):
    """lqg/system.py:167-207: joint (state, belief) dynamics F_j[T,m,m] and noise factor G_j[T,m,nv+nw]."""
    T = dyn["A"].shape[0]
    x, b = dyn["A"].shape[1], actor["A"].shape[1]
    nv, nw = dyn["V"].shape[2], dyn["W"].shape[2]
    Fj, Gj = np.zeros((T, x + b, x + b)), np.zeros((T, x + b, nv + nw))
    for t in range(T):
        Ad, Bd, Fd, Vd, Wd = (dyn[k][t] for k in ("A", "B", "F", "V", "W"))
        Aa, Ba, Fa = (actor[k][t] for k in ("A", "B", "F"))
        Fj[t, :x, :x] = Ad                                            # system.py:169
        Fj[t, :x, x:] = Bd @ L[t]
        Fj[t, x:, :x] = K[t] @ Fd @ Ad                                # system.py:172
        Fj[t, x:, x:] = (Aa + Ba @ L[t] - K[t] @ Fa @ Aa
                         + K[t] @ (Fd @ Bd - Fa @ Ba) @ L[t])         # system.py:173-181
        Gj[t, :x, :nv] = Vd                                           # system.py:194-199
        Gj[t, x:, :nv] = K[t] @ Fd @ Vd                               # system.py:202
        Gj[t, x:, nv:] = K[t] @ Wd
    return Fj, Gj


def conditional_moments(actor, dyn, x, Sigma0=None, eps=1e-8):
    """lqg/system.py:142-235 for ONE trajectory x[T+1,d] -> mu[T,m], Sigma[T,m,m].

    Quirks kept: Sigma0 only feeds kf.forward (the moment recursion starts from
    G_j[0] G_j[0]^T, system.py:212); the affine gain l is ignored; mu0 is zero outside the observed block."""
    T1, o = x.shape
    x_dim, b = dyn["A"].shape[1], actor["A"].shape[1]
    L, _, _ = backward(actor, eps)                                    # system.py:157
    K = forward(actor, default_sigma0(actor) if Sigma0 is None else Sigma0)  # system.py:158-161
    Fj, Gj = joint_system(actor, dyn, L, K)
    T = Fj.shape[0]
    mu = np.concatenate([x[0], np.zeros(x_dim - o + b)])              # system.py:211
    Sig = Gj[0] @ Gj[0].T                                             # system.py:212
    mus, Sigs = np.empty((T, x_dim + b)), np.empty((T, x_dim + b, x_dim + b))
    for t in range(T):                                                # scan over (F, G, x[:-1]), system.py:233
        F, G = Fj[t], Gj[t]
        FS = F @ Sig
        mu = F @ mu + FS[:, :o] @ np.linalg.solve(Sig[:o, :o], x[t] - mu[:o])       # system.py:219-221
        Sig = (F @ Sig @ F.T + G @ G.T
               - FS[:, :o] @ np.linalg.solve(Sig[:o, :o], (Sig @ F.T)[:o, :]))     # system.py:223-230
        mus[t], Sigs[t] = mu, Sig
    return mus, Sigs


def mvn_logpdf(value, mu, Sigma):
    """numpyro MultivariateNormal.log_prob (third-party, numpyro 0.19.0): Cholesky, tri-solve, log-det."""
    d = value.shape[-1]
    Lc = np.linalg.cholesky(Sigma)
    z = np.linalg.solve(Lc, value - mu)
    return -0.5 * (d * np.log(2 * np.pi) + z @ z) - np.log(np.diag(Lc)).sum()


def log_likelihood(actor, dyn, x, Sigma0=None, eps=1e-8):
    """lqg/system.py:237-248: x[n,T+1,d] -> ll[n] = sum_t log N(x[t+1]; mu_t[:d], Sigma_t[:d,:d])."""
    x = np.asarray(x, dtype=np.float64)
    n, T1, d = x.shape
    out = np.empty(n)
    for i in range(n):                                                # vmap over trials, system.py:241
        mu, Sig = conditional_moments(actor, dyn, x[i], Sigma0, eps)
        out[i] = sum(mvn_logpdf(x[i, t + 1], mu[t, :d], Sig[t, :d, :d]) for t in range(T1 - 1))
    return out


def simulate(actor, dyn, eps_noise, eta_noise, x0=None, xhat0=None, Sigma0=None, eps=1e-8):
    """lqg/system.py:62-140 with the standard-normal draws supplied by the caller
    (eps_noise[n,T,xdim], eta_noise[n,T,ydim]; the reference draws them from jax.random, :102-105).

    Returns x[n,T+1,x], xhat[n,T+1,b], y[n,T,y], u[n,T,u]."""
    n, T, xd = eps_noise.shape
    b, ud, yd = actor["A"].shape[1], dyn["B"].shape[2], dyn["F"].shape[1]
    L, l, _ = backward(actor, eps)                                    # system.py:81
    K = forward(actor, default_sigma0(actor) if Sigma0 is None else Sigma0)  # system.py:82
    X, XH = np.zeros((n, T + 1, xd)), np.zeros((n, T + 1, b))
    Y, U = np.zeros((n, T, yd)), np.zeros((n, T, ud))
    for i in range(n):
        xs = np.zeros(xd) if x0 is None else np.array(x0, dtype=np.float64)
        xh = np.zeros(b) if xhat0 is None else np.array(xhat0, dtype=np.float64)
        X[i, 0], XH[i, 0] = xs, xh
        for t in range(T):
            u = L[t] @ xh + l[t]                                      # system.py:110
            xs = dyn["A"][t] @ xs + dyn["B"][t] @ u + dyn["V"][t] @ eps_noise[i, t]   # :113-117
            yy = dyn["F"][t] @ xs + dyn["W"][t] @ eta_noise[i, t]     # :120
            xp = actor["A"][t] @ xh + actor["B"][t] @ u               # :123
            xh = xp + K[t] @ (yy - actor["F"][t] @ xp)                # :124
            X[i, t + 1], XH[i, t + 1], Y[i, t], U[i, t] = xs, xh, yy, u
    return X, XH, Y, U


def brute_force_loglik(actor, dyn, x, Sigma0=None, eps=1e-8):
    """Reference-independent check: log p(x_{1:T} | x_0) from the stacked joint linear-Gaussian system.

    z_k = (state, belief) after k steps with z_0 ~ N(mu0, G_0 G_0^T) *conditioned on* x_0 (the recursion
    of system.py:209-233 conditions step 0 on x[0] whose mean already equals x[0]); z_{k+1} = F_k z_k + G_k w_k.
    Builds the full Gaussian over (o_0 .. o_T) and conditions on o_0 directly."""
    T1, o = x.shape
    T = T1 - 1
    xd, b = dyn["A"].shape[1], actor["A"].shape[1]
    m = xd + b
    L, _, _ = backward(actor, eps)
    K = forward(actor, default_sigma0(actor) if Sigma0 is None else Sigma0)
    Fj, Gj = joint_system(actor, dyn, L, K)
    mean = np.zeros((T + 1, m))
    mean[0] = np.concatenate([x[0], np.zeros(m - o)])
    cov = np.zeros((T + 1, m, T + 1, m))
    cov[0, :, 0, :] = Gj[0] @ Gj[0].T
    # propagate: cov[i,:,j,:] = Cov(z_i, z_j)
    for k in range(T):
        mean[k + 1] = Fj[k] @ mean[k]
        for j in range(k + 1):
            c = Fj[k] @ cov[k, :, j, :]
            cov[k + 1, :, j, :] = c
            cov[j, :, k + 1, :] = c.T
        cov[k + 1, :, k + 1, :] = Fj[k] @ cov[k, :, k, :] @ Fj[k].T + Gj[k] @ Gj[k].T
    mo = mean[:, :o].reshape(-1)
    Co = cov[:, :o, :, :o].reshape((T + 1) * o, (T + 1) * o)
    v = np.asarray(x, dtype=np.float64).reshape(-1)
    # condition on the first o entries (x_0)
    C00, C10, C11 = Co[:o, :o], Co[o:, :o], Co[o:, o:]
    mc = mo[o:] + C10 @ np.linalg.solve(C00, v[:o] - mo[:o])
    Cc = C11 - C10 @ np.linalg.solve(C00, C10.T)
    return mvn_logpdf(v[o:], mc, Cc)
