"""-m gpu: randomised parity of the HIP path against the fp64 C oracle (seeded): random dense stable systems of the
compiled shapes — time-invariant and time-varying, with and without affine cost terms, partial observation, custom
Sigma0 — plus the reference-independent pins (brute-force joint Gaussian, DARE steady state) run on the GPU path."""
import numpy as np
import pytest
import scipy.linalg as sla
import torch

import lqg_amd
import lqg_np as O
from gpu_common import np_, to_spec
from lqg_amd.belief import kf
from lqg_amd.control import lqr

pytestmark = pytest.mark.gpu
SHAPES = [(2, 2, 1, 2), (2, 3, 1, 2), (4, 4, 1, 3), (4, 4, 2, 4), (2, 2, 1, 1)]     # (x, b, u, y)


def random_system(rng, x, b, u, y, T, time_varying, affine):
    """Random well-posed actor / dynamics specs (dense matrices: no structure for the specialiser to exploit)."""
    def stack(fn):
        return np.stack([fn() for _ in range(T)]) if time_varying else np.broadcast_to(fn(), (T,) + fn().shape).copy()

    def spd(n, lo=0.2):
        M = rng.standard_normal((n, n))
        return M @ M.T / n + lo * np.eye(n)

    base = dict(Aa=np.eye(b) * 0.95 + 0.05 * rng.standard_normal((b, b)), Ba=0.3 * rng.standard_normal((b, u)),
                Fa=rng.standard_normal((y, b)), Va=0.5 * rng.standard_normal((b, b)) + 0.5 * np.eye(b),
                Wa=np.diag(1.0 + rng.random(y)), Q=spd(b), R=spd(u, 0.5),
                Ad=np.eye(x) * 0.95 + 0.05 * rng.standard_normal((x, x)), Bd=0.3 * rng.standard_normal((x, u)),
                Fd=rng.standard_normal((y, x)), Vd=0.5 * rng.standard_normal((x, x)) + 0.7 * np.eye(x),
                Wd=np.diag(1.0 + rng.random(y)))
    jit = (lambda M: M * (1 + 0.02 * rng.standard_normal(M.shape))) if time_varying else (lambda M: M)
    sym = lambda M: 0.5 * (M + M.T)
    mk = lambda k, s=False: np.stack([sym(jit(base[k])) if s else jit(base[k]) for _ in range(T)]) if time_varying \
        else np.broadcast_to(base[k], (T,) + base[k].shape).copy()
    actor = dict(A=mk("Aa"), B=mk("Ba"), F=mk("Fa"), V=mk("Va"), W=mk("Wa"), Q=mk("Q", True), R=mk("R", True),
                 q=(0.1 * rng.standard_normal((T, b)) if affine else np.zeros((T, b))),
                 P=(0.05 * rng.standard_normal((T, u, b)) if affine else np.zeros((T, u, b))),
                 r=(0.1 * rng.standard_normal((T, u)) if affine else np.zeros((T, u))),
                 Qf=base["Q"] * 1.5, qf=(0.1 * rng.standard_normal(b) if affine else np.zeros(b)))
    dyn = dict(A=mk("Ad"), B=mk("Bd"), F=mk("Fd"), V=mk("Vd"), W=mk("Wd"), Q=np.zeros((T, x, x)), R=np.zeros((T, u, u)),
               q=np.zeros((T, x)), P=np.zeros((T, u, x)), r=np.zeros((T, u)), Qf=np.zeros((x, x)), qf=np.zeros(x))
    return actor, dyn


# fp32 log-likelihood: measured worst case over the ten systems 8.1e-7 of the largest |ll| (scripts/rand_err.py) — inside the
# north star's 1e-6; asserted at 2e-6 (short horizons: |ll| is only 20-200, so one ulp of a term weighs more than at T = 500)
@pytest.mark.parametrize("dtype,tol_ll,tol_m", [(torch.float64, 1e-9, 1e-8), (torch.float32, 2e-6, 5e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("case", range(10))
def test_random_systems_match_the_oracle(oracle_lib, case, dtype, tol_ll, tol_m):
    rng = np.random.default_rng(1000 + case)
    x, b, u, y = SHAPES[case % len(SHAPES)]
    T = int(rng.integers(5, 60))
    tv, affine = bool(case & 1), bool(case & 2)
    actor, dyn = random_system(rng, x, b, u, y, T, tv, affine)
    d = x if case % 3 else max(1, x - 1) if (x, b, u, y) != (4, 4, 2, 4) else x          # sometimes partial observation
    if (x, d) not in ((2, 2), (4, 4), (4, 2)):
        d = x
    S0 = None
    if case % 4 == 3:
        M = rng.standard_normal((b, b))
        S0 = M @ M.T / b + 0.3 * np.eye(b)
    n = 3
    X, _, _, _ = oracle_lib.simulate(actor, dyn, rng.standard_normal((n, T, x)), rng.standard_normal((n, T, y)), Sigma0=S0)
    xs = X[..., :d]
    ref_ll = oracle_lib.log_likelihood(actor, dyn, xs, S0)
    ref_mu, ref_Sig = oracle_lib.conditional_moments(actor, dyn, xs, S0)
    a_t = to_spec(actor, dtype)
    d_t = to_spec(dyn, dtype)
    sys_ = lqg_amd.System(actor=a_t, dynamics=d_t)
    S0t = None if S0 is None else torch.as_tensor(S0, dtype=dtype, device="cuda")
    xt = torch.as_tensor(xs, dtype=dtype, device="cuda")
    scale = np.abs(ref_ll).max()
    assert np.abs(np_(sys_.log_likelihood(xt, Sigma0=S0t)) - ref_ll).max() < tol_ll * scale          # several trials
    assert abs(float(sys_.log_likelihood(xt[:1], Sigma0=S0t)[0]) - ref_ll[0]) < tol_ll * scale      # one trial
    mu, Sig = sys_._moments(xt, S0t)
    assert np.abs(np_(mu) - ref_mu).max() < tol_m * max(1.0, np.abs(ref_mu).max())
    assert np.abs(np_(Sig) - ref_Sig).max() < tol_m * max(1.0, np.abs(ref_Sig).max())
    L, l, H = oracle_lib.riccati_backward(actor)
    g = lqr.backward(a_t)
    assert np.abs(np_(g.L) - L).max() < tol_m * max(1.0, np.abs(L).max())
    assert np.abs(np_(g.l) - l).max() < tol_m * max(1.0, np.abs(L).max())
    K = oracle_lib.kalman_forward(actor, S0)
    assert np.abs(np_(kf.forward(a_t, S0t)) - K).max() < tol_m * max(1.0, np.abs(K).max())


def test_gpu_loglik_equals_brute_force_joint_gaussian():
    """Reference-independent: the HIP path against log p(x_1..x_T | x_0) of the stacked linear-Gaussian system."""
    T = 6
    m = lqg_amd.SubjectiveActor(dim=1, T=T, subj_noise=1.3, subj_vel_noise=0.6, sigma_cursor=1.0, action_cost=0.05,
                                device="cuda", dtype=torch.float64)
    x = m.simulate(3, n=2)
    act = {f: getattr(m.actor, f).cpu().numpy() for f in O.FIELDS}
    dyn = {f: getattr(m.dynamics, f).cpu().numpy() for f in O.FIELDS}
    ll = m.log_likelihood(x)
    for i in range(2):
        assert abs(float(ll[i]) - O.brute_force_loglik(act, dyn, x[i].cpu().numpy())) < 1e-10


def test_gpu_kalman_gain_converges_to_dare():
    m = lqg_amd.BoundedActor(T=3000, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05, device="cuda",
                             dtype=torch.float64)
    K = kf.forward(m.actor, None)
    A, F, V, W = (getattr(m.actor, f)[0].cpu().numpy() for f in ("A", "F", "V", "W"))
    P = sla.solve_discrete_are(A.T, F.T, V @ V.T, W @ W.T)
    Kss = P @ F.T @ np.linalg.inv(F @ P @ F.T + W @ W.T)
    assert np.allclose(np_(K[-1]), Kss, rtol=1e-8, atol=1e-12)
    g = lqr.backward(m.actor)
    c = (1.0 / 60) / (0.05 + (1.0 / 60) ** 2)
    assert np.allclose(np_(g.L[-1]), [[c, -c]], rtol=1e-12)                      # closed form of the last gain


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)], ids=["f64", "f32"])
def test_temporal_delay_model_on_the_hip_path(oracle_lib, dtype, tol):
    """lqg/tracking/delay.py: a delayed BoundedActor (shape compiled on first use) — likelihood over a candidate
    axis against the fp64 oracle, and the delay-12 model of the reference on the cooperative kernels."""
    from lqg_amd.tracking.delay import DelayedSubjectiveActor, TemporalDelayModel
    sig = torch.tensor([4.0, 8.0, 16.0], dtype=dtype, device="cuda")
    m = TemporalDelayModel(lqg_amd.BoundedActor(T=80, sigma_target=sig, action_cost=0.1, device="cuda", dtype=dtype), delay=2)
    one = TemporalDelayModel(lqg_amd.BoundedActor(T=80, sigma_target=8.0, action_cost=0.1, device="cuda", dtype=dtype), delay=2)
    x = one.simulate(5, n=6)[..., :2]
    ll = m.log_likelihood(x)
    assert ll.shape == (3, 6)
    for c, s in enumerate((4.0, 8.0, 16.0)):
        ref_m = TemporalDelayModel(lqg_amd.BoundedActor(T=80, sigma_target=s, action_cost=0.1, device="cpu", dtype=torch.float64), delay=2)
        act = {f: getattr(ref_m.actor, f).numpy().copy() for f in O.FIELDS}
        dyn = {f: getattr(ref_m.dynamics, f).numpy().copy() for f in O.FIELDS}
        ref = oracle_lib.log_likelihood(act, dyn, x.double().cpu().numpy())
        assert np.abs(np_(ll[c]) - ref).max() < tol * np.abs(ref).max()
    # the delay-12 model of the reference (x=26, b=39): beyond the lane kernels, served by the cooperative run-time-dims
    # kernels (simulate included) — against the fp64 oracle
    big = DelayedSubjectiveActor(T=20, device="cuda", dtype=dtype)
    xb = big.simulate(7, n=3)[..., :2].contiguous()
    ref_b = DelayedSubjectiveActor(T=20, device="cpu", dtype=torch.float64)
    act = {f: getattr(ref_b.actor, f).numpy().copy() for f in O.FIELDS}
    dyn = {f: getattr(ref_b.dynamics, f).numpy().copy() for f in O.FIELDS}
    ref = oracle_lib.log_likelihood(act, dyn, xb.double().cpu().numpy())
    assert np.abs(np_(big.log_likelihood(xb)) - ref).max() < tol * np.abs(ref).max()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-6)], ids=["f64", "f32"])
@pytest.mark.parametrize("ctor,kw,d", [("SubjectiveActor", dict(dim=1), 2), ("BoundedActor", dict(dim=1), 2),
                                       ("SubjectiveActor", dict(dim=2), 4), ("PointMassBoundedActor", {}, 2)])
def test_explicit_dense_sigma0_on_the_specialised_path(oracle_lib, ctor, kw, d, dtype, tol):
    """An explicit (dense) Sigma0 selects the DENSE_P variant of the specialised forward kernel — the Kalman covariance
    cannot carry the structural mask derived from V V' — for one, two and several trials; against the fp64 oracle."""
    m = getattr(lqg_amd, ctor)(T=70, device="cuda", dtype=dtype, **kw)
    b = m.bdim
    rng = np.random.default_rng(5)
    M = rng.standard_normal((b, b))
    S0 = M @ M.T / b + 0.5 * np.eye(b)                                  # dense: couples every belief state
    S0t = torch.as_tensor(S0, dtype=dtype, device="cuda")
    with torch.no_grad():
        x = m.simulate(8, n=5)[..., :d].contiguous()
    act = {f: getattr(m.actor, f).double().cpu().numpy().copy() for f in O.FIELDS}
    dyn = {f: getattr(m.dynamics, f).double().cpu().numpy().copy() for f in O.FIELDS}
    ref = oracle_lib.log_likelihood(act, dyn, x.double().cpu().numpy(), S0)
    for n in (1, 2, 5):
        got = np_(m.log_likelihood(x[:n], Sigma0=S0t))
        assert np.abs(got - ref[:n]).max() < tol * np.abs(ref).max(), n
    assert np.abs(np_(m.log_likelihood(x)) - oracle_lib.log_likelihood(act, dyn, x.double().cpu().numpy())).max() \
        < tol * np.abs(ref).max()                                       # and the default Sigma0 = V V' on the same data


@pytest.mark.parametrize("ctor,dim,names", [
    ("BoundedActor", 1, ("action_variability", "action_cost", "sigma_target", "sigma_cursor")),
    ("BoundedActor", 2, ("action_variability", "action_cost", "sigma_target", "sigma_cursor")),
    ("OptimalActor", 2, ("action_variability", "sigma_target", "sigma_cursor")),
    ("RelativeObservationBoundedActor", 2, ("action_variability", "sigma", "action_cost")),
    ("SubjectiveActor", 1, ("action_variability", "action_cost", "subj_noise", "subj_vel_noise", "sigma_target", "sigma_cursor")),
    ("SubjectiveActor", 2, ("action_variability", "action_cost", "subj_noise", "subj_vel_noise", "sigma_target", "sigma_cursor")),
])
def test_zoo_models_over_wide_parameter_ranges(oracle_lib, ctor, dim, names):
    """Every structural shortcut of the default path (class-level sparsity pattern, decoupling, identical axes solved once,
    loop-carried Kalman mask) on 48 candidates drawn log-uniformly over four decades, fp64, against the literal C oracle."""
    rng = np.random.default_rng(sum(map(ord, ctor)) * 10 + dim)            # deterministic per case
    C_ = 48
    cand = {k: torch.as_tensor(10.0 ** rng.uniform(-1.5, 2.0, C_), dtype=torch.float64, device="cuda") for k in names}
    m = getattr(lqg_amd, ctor)(dim=dim, T=60, device="cuda", dtype=torch.float64, **cand)
    d = 2 * dim
    with torch.no_grad():
        x = m.simulate(17, n=2)[..., :d].contiguous()                  # [C, 2, T+1, d]
    ll = np_(m.log_likelihood(x))
    assert ll.shape == (C_, 2) and np.isfinite(ll).all()
    F = O.FIELDS
    host = lambda spec: {f: (getattr(spec, f) if getattr(spec, f).dim() == (3 if f in ("Qf",) else (2 if f == "qf" else (3 if f in ("q", "r") else 4)))
                             else getattr(spec, f).expand(C_, *getattr(spec, f).shape)).double().cpu().numpy().copy() for f in F}
    ref = oracle_lib.log_likelihood(host(m.actor), host(m.dynamics), x.cpu().numpy())
    assert np.abs(ll / ref - 1).max() < 1e-9
