"""Differentiable restatement of the LQG log-likelihood in plain torch (CPU, fp64) — TEST INFRASTRUCTURE ONLY.

Same literal algorithm as oracle/lqg_np.py (which is pinned against the reference-generated golden vectors), written
with torch ops so that torch.autograd supplies the ground-truth gradient d ll / d(spec matrices) that the reference
obtains from jax.grad (lqg/optim.py:142-147, notebooks/Tutorial.ipynb cell 40).  Used by tests/ to pin
oracle/lqg_adjoint_np.py and the HIP adjoint kernels.  Time-invariant or time-varying specs: dicts of tensors with the
time axis first, as lqg_np.  The eigenvalue floor (lqr.py:27-28) is applied with the shift held constant (it is zero
whenever R + B'SB is positive definite, which holds for every model of the zoo)."""
import math

import torch


def backward(spec, eps=1e-8):
    T = spec["A"].shape[0]
    S = spec["Qf"]
    Ls = [None] * T
    for t in range(T - 1, -1, -1):
        Q, P, R, A, B = (spec[k][t] for k in ("Q", "P", "R", "A", "B"))
        H = R + B.T @ S @ B
        G = P + B.T @ S @ A
        shift = max(0.0, eps - float(torch.linalg.eigvalsh(H.detach())[0]))
        Ht = H + shift * torch.eye(H.shape[0], dtype=H.dtype)
        L = -torch.linalg.solve(Ht, G)
        S = Q + A.T @ S @ A + L.T @ H @ L + L.T @ G + G.T @ L
        Ls[t] = L
    return Ls


def forward(spec, Sigma0):
    T = spec["A"].shape[0]
    P = Sigma0
    Ks = []
    I = torch.eye(P.shape[0], dtype=P.dtype)
    for t in range(T):
        A, F, V, W = (spec[k][t] for k in ("A", "F", "V", "W"))
        P = A @ P @ A.T + V @ V.T
        G = F @ P @ F.T + W @ W.T
        K = P @ F.T @ torch.linalg.inv(G)
        P = (I - K @ F) @ P
        Ks.append(K)
    return Ks


def log_likelihood(actor, dyn, x, Sigma0=None, eps=1e-8):
    """x[n, T+1, d] -> ll[n]; differentiable w.r.t. every tensor in actor / dyn (and Sigma0)."""
    n, T1, o = x.shape
    T = T1 - 1
    xd, b = dyn["A"].shape[1], actor["A"].shape[1]
    Ls = backward(actor, eps)
    Ks = forward(actor, actor["V"][0] @ actor["V"][0].T if Sigma0 is None else Sigma0)
    out = []
    for i in range(n):
        mu = torch.cat([x[i, 0], torch.zeros(xd - o + b, dtype=x.dtype)])
        Sig = None
        ll = 0.0
        for t in range(T):
            Ad, Bd, Fd, Vd, Wd = (dyn[k][t] for k in ("A", "B", "F", "V", "W"))
            Aa, Ba, Fa = (actor[k][t] for k in ("A", "B", "F"))
            L, K = Ls[t], Ks[t]
            F = torch.cat([torch.cat([Ad, Bd @ L], 1),
                           torch.cat([K @ Fd @ Ad, Aa + Ba @ L - K @ Fa @ Aa + K @ (Fd @ Bd - Fa @ Ba) @ L], 1)], 0)
            G = torch.cat([torch.cat([Vd, torch.zeros(xd, Wd.shape[1], dtype=x.dtype)], 1),
                           torch.cat([K @ Fd @ Vd, K @ Wd], 1)], 0)
            if Sig is None:
                Sig = G @ G.T
            FS = F @ Sig
            Soo = Sig[:o, :o]
            mu = F @ mu + FS[:, :o] @ torch.linalg.solve(Soo, x[i, t] - mu[:o])
            Sig = F @ Sig @ F.T + G @ G.T - FS[:, :o] @ torch.linalg.solve(Soo, (Sig @ F.T)[:o, :])
            Lc = torch.linalg.cholesky(Sig[:o, :o])
            z = torch.linalg.solve_triangular(Lc, (x[i, t + 1] - mu[:o])[:, None], upper=False)[:, 0]
            ll = ll - 0.5 * (o * math.log(2 * math.pi) + z @ z) - torch.log(torch.diagonal(Lc)).sum()
        out.append(ll)
    return torch.stack(out)
