"""Generate tests/golden/*.npz by running the REFERENCE's own source (read-only at /root/reference)
under oracle/jax_standin.py (NumPy/LAPACK fp64 in place of jax/numpyro, which are not installed).

Build-container only: /root/reference does not exist on the GPU box; only the .npz vectors travel.
Run:  python oracle/gen_golden.py            (rewrites every fixture deterministically)

Each fixture holds, in fp64:
  actor_<field>, dyn_<field>   the two LQGSpec stacks exactly as the reference built them
  x[n,T+1,d]                   observed trajectories (simulated from the model, first d state dims)
  L,l,H                        lqg.control.lqr.backward(actor)            (lqg/control/lqr.py:16-42)
  K                            lqg.belief.kf.forward(actor, Sigma0)       (lqg/belief/kf.py:6-21)
  mu[n,T,m], Sigma[n,T,m,m]    System.conditional_moments per trial       (lqg/system.py:142-235)
  ll[n]                        System.log_likelihood(x)                   (lqg/system.py:246-248)
  sim_eps, sim_eta, sim_x, sim_xhat, sim_y, sim_u   System.simulate with the normal draws recorded
                                                                          (lqg/system.py:62-140)
  Sigma0 (optional)            non-default initial belief covariance passed to the calls above
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import jax_standin  # noqa: E402

jax_standin.install("/root/reference")

import jax  # noqa: E402  (the stand-in)
import jax.numpy as jnp  # noqa: E402
from jax import random  # noqa: E402
from jax.scipy import linalg as jsl  # noqa: E402
from lqg.belief import kf  # noqa: E402
from lqg.control import lqr  # noqa: E402
from lqg.spec import LQGSpec  # noqa: E402
from lqg.system import LQG, Actor, Dynamics, System  # noqa: E402
from lqg.tracking import (BoundedActor, OptimalActor, PointMassBoundedActor,  # noqa: E402
                          RelativeObservationBoundedActor, SubjectiveActor)
from lqg.tracking.delay import TemporalDelayModel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
FIELDS = LQGSpec._fields


class NoiseRecorder:
    """Wrap the stand-in's random.normal so the draws System.simulate consumes are kept."""

    def __init__(self):
        self.draws = []
        self._orig = random.normal

    def __enter__(self):
        def rec(key, shape=()):
            v = self._orig(key, shape=shape)
            self.draws.append(np.array(v))
            return v
        random.normal = rec
        jax.random.normal = rec
        return self

    def __exit__(self, *a):
        random.normal = self._orig
        jax.random.normal = self._orig


def run_case(name, system, n, d, seed, Sigma0=None, x0=None, sigma_copies=None):
    """sigma_copies: keep only that many of the n per-trial Sigma stacks (they are identical: Sigma does not depend on
    the data) — for the large cases, to keep the fixture small."""
    T = system.T
    with NoiseRecorder() as rec:
        sx, sxh, sy, su = system.simulate(random.PRNGKey(seed), n=n, x0=x0, Sigma0=Sigma0, return_all=True)
    eps = np.stack(rec.draws[0::2])      # simulate_trial draws epsilon then eta (system.py:102-105)
    eta = np.stack(rec.draws[1::2])
    x = np.array(sx[:, :, :d])
    gains = lqr.backward(system.actor)
    S0 = system.actor.V[0] @ system.actor.V[0].T if Sigma0 is None else Sigma0
    K = kf.forward(system.actor, S0)
    mus, Sigs = [], []
    for i in range(n):
        mu, Sig = system.conditional_moments(x[i], Sigma0=Sigma0)
        mus.append(mu), Sigs.append(Sig)
    ll = system.log_likelihood(x, Sigma0=Sigma0)
    for S_ in Sigs[1:]:
        assert np.array_equal(np.asarray(S_), np.asarray(Sigs[0]))        # Sigma is data-independent
    out = {"x": x, "L": gains.L, "l": gains.l, "H": gains.H, "K": K,
           "mu": np.stack(mus), "Sigma": np.stack(Sigs[:sigma_copies]), "ll": ll,
           "sim_eps": eps, "sim_eta": eta, "sim_x": sx, "sim_xhat": sxh, "sim_y": sy, "sim_u": su}
    if Sigma0 is not None:
        out["Sigma0"] = np.array(Sigma0)
    if x0 is not None:
        out["x0"] = np.array(x0)
    for f in FIELDS:
        out["actor_" + f] = np.array(getattr(system.actor, f))
        out["dyn_" + f] = np.array(getattr(system.dynamics, f))
    out = {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}
    assert np.isfinite(out["ll"]).all(), name
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name:28s} T={T:4d} n={n} x={system.xdim} b={system.bdim} u={system.udim} y={system.ydim} d={d} "
          f"ll[0]={out['ll'][0]:.10f}")


def tutorial_lqg(T):
    """notebooks/Tutorial.ipynb cell 14 (BASELINE config 1)."""
    dt = 1.0 / 60.0
    A = jnp.eye(2)
    B = jnp.array([[0.0], [dt]])
    V = jnp.diag(jnp.array([1.0, 0.5]))
    C = jnp.eye(2)
    W = jnp.diag(jnp.array([6.0, 1.0]))
    Q = jnp.array([[1.0, -1.0], [-1.0, 1.0]])
    R = jnp.eye(1) * 0.05
    return LQG(A, B, C, V, W, Q, R, T=T)


def hand2d(T, cursor_noise=0.1):
    """2-D version of notebooks/HandModel.ipynb's HandMotionModelTrackingTask (BASELINE config 4),
    observed dims (target, cursor position per axis) permuted to the front as SubjectiveActor does
    (lqg/tracking/subjective.py:7-12,38-44), and position noise `cursor_noise` added so that the observed
    block of V V^T is non-singular (the notebook's V makes the likelihood singular)."""
    dt, m, tau = 1.0 / 60.0, 1.0, 0.04
    A1 = jsl.block_diag(jnp.eye(1), jnp.array([[1.0, dt, 0.0, 0.0], [0.0, 1.0, dt / m, 0.0],
                                               [0.0, 0.0, 1.0 - dt / tau, dt / tau],
                                               [0.0, 0.0, 0.0, 1.0 - dt / tau]]))
    B1 = dt / tau * jnp.array([[0.0], [0.0], [0.0], [0.0], [1.0]])
    F1 = jnp.eye(2, 5)
    V1 = jnp.diag(jnp.array([1.0, cursor_noise, 0.0, 0.0, 0.5]))
    W1 = jnp.diag(jnp.array([6.0, 6.0]))
    Q1 = jsl.block_diag(jnp.array([[1.0, -1.0], [-1.0, 1.0]]), jnp.zeros((3, 3)))
    A, B, F = jsl.block_diag(A1, A1), jsl.block_diag(B1, B1), jsl.block_diag(F1, F1)
    V, W, Q = jsl.block_diag(V1, V1), jsl.block_diag(W1, W1), jsl.block_diag(Q1, Q1)
    R = jnp.eye(2) * 1.0
    perm = [0, 1, 5, 6, 2, 3, 4, 7, 8, 9]
    A, B, V, F, Q = A[perm][:, perm], B[perm], V[perm], F[:, perm], Q[perm][:, perm]
    spec = Actor(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
    return System(actor=spec, dynamics=spec)


def time_varying(T, seed):
    """Hand-built genuinely time-varying actor != dynamics system with non-zero affine terms q, r, P, qf
    (exercises every LQGSpec field; x=2, b=3, u=1, y=2 like SubjectiveActor(dim=1))."""
    rng = np.random.default_rng(seed)
    base = SubjectiveActor(dim=1, T=T, action_cost=0.3, sigma_target=5.0, sigma_cursor=2.0)

    def jitter(a, scale=0.05):
        return np.array(a) * (1.0 + scale * rng.standard_normal(a.shape)) + 0.01 * scale * rng.standard_normal(a.shape)

    act, dyn = base.actor, base.dynamics
    b, u = 3, 1
    Qj = jitter(act.Q)
    Qj = 0.5 * (Qj + np.swapaxes(Qj, 1, 2)) + 0.05 * np.eye(b)
    Rj = np.abs(jitter(act.R)) + 0.01
    actor = LQGSpec(Q=Qj, q=0.1 * rng.standard_normal((T, b)), Qf=Qj[-1] * 2.0, qf=0.1 * rng.standard_normal(b),
                    P=0.02 * rng.standard_normal((T, u, b)), R=Rj, r=0.05 * rng.standard_normal((T, u)),
                    A=jitter(act.A, 0.01), B=jitter(act.B), V=jitter(act.V), F=jitter(act.F, 0.01), W=jitter(act.W))
    x = 2
    dynamics = LQGSpec(Q=np.zeros((T, x, x)), q=np.zeros((T, x)), Qf=np.zeros((x, x)), qf=np.zeros(x),
                       P=np.zeros((T, u, x)), R=np.zeros((T, u, u)), r=np.zeros((T, u)),
                       A=jitter(dyn.A, 0.01), B=jitter(dyn.B), V=jitter(dyn.V), F=jitter(dyn.F, 0.01),
                       W=jitter(dyn.W))
    return System(actor=actor, dynamics=dynamics)


def tracking_io_case():
    """lqg.io.load_tracking_data on a small synthetic data.mat (same fields/dtypes as the Bonnen et al. file:
    sigma uint8 [n], target float64 [n, S], response uint16 [n, S]); inputs and outputs go into one fixture."""
    import tempfile

    import scipy.io as spio
    from lqg.io import load_tracking_data

    rng = np.random.default_rng(77)
    n, S = 12, 400
    sigma = np.repeat(np.array([11, 13, 17, 21], dtype=np.uint8), 3)[rng.permutation(n)]
    target = 600.0 + np.cumsum(rng.standard_normal((n, S)) * 3.0, axis=1)
    response = np.clip(np.round(np.roll(target, 9, axis=1) + rng.standard_normal((n, S)) * 4.0), 0, 65535).astype(np.uint16)
    out = {"sigma": sigma, "target": target, "response": response}
    with tempfile.TemporaryDirectory() as tmp:
        spio.savemat(os.path.join(tmp, "data.mat"), out)
        for tag, kw in (("default", dict(delay=12, clip=120, subtract_mean=True)),
                        ("nodelay", dict(delay=0, clip=50, subtract_mean=True)),
                        ("raw", dict(delay=5, clip=0, subtract_mean=False))):
            data, sigmas = load_tracking_data(data_path=tmp, **kw)
            out["data_" + tag], out["sigmas_" + tag] = np.asarray(data), np.asarray(sigmas)
    os.makedirs(os.path.join(OUT, "io"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "io", "tracking_small.npz"), **out)
    print("io/tracking_small", {k: v.shape for k, v in out.items() if k.startswith("data_")})


def main():
    os.makedirs(OUT, exist_ok=True)
    # BASELINE config 1: tutorial LQG, state dim 2, T=100
    run_case("tutorial_lqg_T100", tutorial_lqg(100), n=4, d=2, seed=11)
    # analytic-pin case of SURVEY.md §4
    run_case("bounded_T100", BoundedActor(T=100, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05,
                                          action_variability=0.5), n=3, d=2, seed=12)
    run_case("optimal_T30", OptimalActor(T=30), n=2, d=2, seed=13)
    run_case("relobs_T40", RelativeObservationBoundedActor(T=40), n=2, d=2, seed=14)
    # x != b
    run_case("subjective1d_T50", SubjectiveActor(dim=1, T=50), n=3, d=2, seed=15)
    # headline dims (x=4, b=6, u=2, y=4, d=4)
    run_case("subjective2d_T60", SubjectiveActor(dim=2, T=60, action_cost=0.5, sigma_cursor=3.0,
                                                 subj_noise=1.3, subj_vel_noise=0.7), n=3, d=4, seed=16)
    run_case("subjective2d_T500", SubjectiveActor(dim=2, T=500), n=3, d=4, seed=21, sigma_copies=1)
    # BASELINE config 2 dims (x=b=4, u=1, y=3); partial (d=2) and full (d=4) observation of the data
    run_case("pointmass_d2_T50", PointMassBoundedActor(T=50, action_variability=0.5), n=2, d=2, seed=17)
    run_case("pointmass_d4_T50", PointMassBoundedActor(T=50, action_variability=0.5), n=2, d=4, seed=17)
    run_case("bounded2d_T40", BoundedActor(dim=2, T=40, action_cost=0.2), n=2, d=4, seed=18)
    # BASELINE config 4 dims (x=b=10, u=2, y=4, m=20)
    run_case("hand2d_T40", hand2d(40), n=2, d=4, seed=19)
    # all LQGSpec fields time-varying, affine terms non-zero, custom Sigma0 and x0
    S0 = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, 0.2], [0.1, 0.2, 1.0]])
    run_case("timevarying_T30", time_varying(30, seed=5), n=3, d=2, seed=20, Sigma0=S0,
             x0=np.array([0.5, -0.25]))
    # temporal-delay augmentation (lqg/tracking/delay.py): shift-register states, singular process noise
    run_case("delay1_bounded_T30", TemporalDelayModel(BoundedActor(T=30, sigma_target=6.0, sigma_cursor=1.0,
                                                                   action_cost=0.05), delay=1), n=2, d=2, seed=22)
    run_case("delay2_bounded_T30", TemporalDelayModel(BoundedActor(T=30, sigma_target=6.0, sigma_cursor=1.0,
                                                                   action_cost=0.05), delay=2), n=2, d=2, seed=23)
    run_case("delay1_subjective1d_T30", TemporalDelayModel(SubjectiveActor(dim=1, T=30, action_cost=0.5), delay=1),
             n=2, d=2, seed=24)
    # the reference's DelayedSubjectiveActor (lqg/tracking/delay.py:44-51): SubjectiveActor with that class's default
    # parameters + a delay of 12 steps -> x=26, b=39, m=65.  The class fixes T=1000 through SubjectiveActor's default;
    # the same constructor lines at a short horizon keep the fixture small.
    run_case("delay12_subjective1d_T30",
             TemporalDelayModel(SubjectiveActor(process_noise=1., action_cost=0.5, action_variability=0.5, subj_noise=1.,
                                                subj_vel_noise=10., sigma_target=6., sigma_cursor=3., dt=1. / 60, T=30),
                                delay=12), n=2, d=2, seed=25, sigma_copies=1)
    tracking_io_case()


if __name__ == "__main__":
    main()
