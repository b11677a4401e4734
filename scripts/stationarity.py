"""When do the per-system recursions of the headline model become BITWISE stationary?  (round-2 review item 2)

CPU experiment, no GPU needed: one component of SubjectiveActor(dim=2) (x=2, b=3, u=1, y=2; subjective.py:18-44) with
the bench's log-uniform candidates (lqg_amd/workload.py RANGES), the three per-system recursions of the hot path run
literally in the requested precision with NumPy (lqr.py:16-42, kf.py:6-21, system.py:209-235), batched over
candidates.  Reports, per candidate and per 64-candidate wave (consecutive candidates, as the kernel maps them):
  t_K   first forward step from which the Kalman covariance P repeats bitwise
  t_L   number of steps before the END of the horizon over which the control gain L_t still changes
  t_S   first forward step from which the joint covariance Sigma repeats bitwise (given stationary K, L)
A wave can leave the full step only inside  [max t_S over its lanes,  T - max t_L over its lanes].
Usage: python scripts/stationarity.py [f32|f64] [n_candidates] [T]  -> JSON on stdout."""
import json
import math
import sys

import numpy as np

RANGES = dict(action_variability=(0.1, 2.0), sigma_target=(1.0, 50.0), sigma_cursor=(1.0, 15.0),
              action_cost=(0.01, 10.0), subj_noise=(0.5, 2.0), subj_vel_noise=(0.1, 2.0))


def main():
    dt_name = sys.argv[1] if len(sys.argv) > 1 else "f32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 14
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    R = np.float32 if dt_name == "f32" else np.float64
    rng = np.random.default_rng(1234)
    p = {k: np.exp(rng.uniform(math.log(lo), math.log(hi), n)).astype(R) for k, (lo, hi) in RANGES.items()}
    dt = R(1.0 / 60)
    z, o = np.zeros(n, R), np.ones(n, R)

    def mat(rows):
        return np.stack([np.stack(r, -1) for r in rows], -2).astype(R)

    # true dynamics (target, cursor) and the actor's model (target, cursor, target velocity)
    Ad = mat([[o, z], [z, o]]); Bd = mat([[z], [o * dt]]); Fd = mat([[o, z], [z, o]])
    Vd = mat([[o, z], [z, p["action_variability"]]]); Wd = mat([[p["sigma_target"], z], [z, p["sigma_cursor"]]])
    Aa = mat([[o, z, o * dt], [z, o, z], [z, z, o]]); Ba = mat([[z], [o * dt], [z]])
    Fa = mat([[o, z, z], [z, o, z]])
    Va = mat([[p["subj_noise"], z, z], [z, p["action_variability"], z], [z, z, p["subj_vel_noise"]]])
    Wa = Wd
    Q = mat([[o, -o, z], [-o, o, z], [z, z, z]]); Rm = mat([[p["action_cost"]]])
    tr = lambda a: np.swapaxes(a, -1, -2)

    # ---- Riccati backward (lqr.py:16-42); u = 1 so the eigenvalue floor is max(0, eps - H)
    S = Q.copy()
    Ls = np.empty((T, n, 1, 3), R)
    for t in range(T - 1, -1, -1):
        H = Rm + tr(Ba) @ S @ Ba
        G = tr(Ba) @ S @ Aa
        Ht = H + np.maximum(R(0), R(1e-8) - H)
        L = -G / Ht
        S = Q + tr(Aa) @ S @ Aa + tr(L) @ H @ L + tr(L) @ G + tr(G) @ L
        Ls[t] = L
    # steps before the end over which L still changes: largest k such that L[T-k] != L[T-k-1]
    changed = (Ls[1:] != Ls[:-1]).any(axis=(2, 3))              # [T-1, n]: L_{t+1} != L_t
    first_change = np.where(changed.any(0), changed.argmax(0), T - 1)   # smallest t with L_{t+1} != L_t
    t_L = (T - 1) - first_change                                # L_t is constant for t <= first_change
    # ---- Kalman forward (kf.py:6-21)
    P = Va @ tr(Va)
    Ks = np.empty((T, n, 3, 2), R)
    t_K = np.full(n, T)
    for t in range(T):
        Pp = Aa @ P @ tr(Aa) + Va @ tr(Va)
        Gk = Fa @ Pp @ tr(Fa) + Wa @ tr(Wa)
        K = Pp @ tr(Fa) @ np.linalg.inv(Gk).astype(R)
        Pn = Pp - K @ Fa @ Pp
        same = (Pn == P).all(axis=(1, 2))
        t_K = np.where(same & (t_K == T), t, t_K)                # first step of the FINAL run of repeats
        t_K = np.where(~same, T, t_K)
        P = Pn
        Ks[t] = K
    # ---- joint system and moment recursion (system.py:167-235), observed block o = 2
    Sg = None
    t_S = np.full(n, T)
    for t in range(T):
        K, L = Ks[t], Ls[t]
        F11, F12 = Ad, Bd @ L
        F21 = K @ Fd @ Ad
        F22 = Aa + Ba @ L - K @ Fa @ Aa + K @ (Fd @ Bd - Fa @ Ba) @ L
        Fj = np.concatenate([np.concatenate([F11, F12], -1), np.concatenate([F21, F22], -1)], -2)
        Gj = np.concatenate([np.concatenate([Vd, np.zeros((n, 2, 2), R)], -1),
                             np.concatenate([K @ Fd @ Vd, K @ Wd], -1)], -2)
        GG = Gj @ tr(Gj)
        if Sg is None:
            Sg = GG
        FS = Fj @ Sg
        gain = FS[:, :, :2] @ np.linalg.inv(Sg[:, :2, :2]).astype(R)
        Sn = FS @ tr(Fj) + GG - gain @ tr(FS[:, :, :2])
        same = (Sn == Sg).all(axis=(1, 2))
        t_S = np.where(same & (t_S == T), t, t_S)
        t_S = np.where(~same, T, t_S)
        Sg = Sn
    w = lambda a: a.reshape(-1, 64).max(1)
    window = np.maximum(0, (T - t_L) - np.maximum(t_S, t_K))
    wave_window = np.maximum(0, (T - w(t_L)) - np.maximum(w(t_S), w(t_K)))
    q = lambda a: {f"p{int(100 * f)}": float(np.quantile(a, f)) for f in (0.1, 0.25, 0.5, 0.75, 0.9)}
    out = dict(dtype=dt_name, candidates=n, T=T,
               per_candidate=dict(t_K=q(t_K), t_L=q(t_L), t_S=q(t_S), steady_window_steps=q(window),
                                  frac_with_any_window=float((window > 0).mean()),
                                  mean_window_frac_of_T=float(window.mean() / T)),
               per_wave_of_64=dict(t_K=q(w(t_K)), t_L=q(w(t_L)), t_S=q(w(t_S)), steady_window_steps=q(wave_window),
                                   frac_with_any_window=float((wave_window > 0).mean()),
                                   mean_window_frac_of_T=float(wave_window.mean() / T)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
