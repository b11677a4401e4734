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


# fp32 log-likelihood: the north star's 1e-6 of the largest |ll| (measured worst case over the ten systems 8.1e-7,
# scripts/rand_err.py; short horizons: |ll| is only 20-200, so one ulp of a term weighs more than at T = 500)
@pytest.mark.parametrize("dtype,tol_ll,tol_m", [(torch.float64, 1e-9, 1e-8), (torch.float32, 1e-6, 5e-4)], ids=["f64", "f32"])
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


# ---------------------------------------------------------------------------------------------------------------------------
# in-kernel draws of System.simulate (lqg_simulate_rng: counter-based Philox4x32-10 + Box-Muller, csrc/lqg_rng.hpp)
def _identity_noise_system(xdim, T, dtype, n_sys=None, ydim=None):
    """A = 0, B = 0, V = I, F = [I 0], W = I: x_{t+1} = eps_t and y_t = x_{t+1}[:y] + eta_t — simulate() hands back its draws."""
    from lqg_amd.system import Actor, System
    dev = torch.device("cuda")
    ydim = xdim if ydim is None else ydim
    lead = () if n_sys is None else (n_sys,)
    z = torch.zeros(lead + (xdim, xdim), dtype=dtype, device=dev)
    e = torch.eye(xdim, dtype=dtype, device=dev).expand(lead + (xdim, xdim)).contiguous()
    F = torch.eye(ydim, xdim, dtype=dtype, device=dev).expand(lead + (ydim, xdim)).contiguous()
    W = torch.eye(ydim, dtype=dtype, device=dev).expand(lead + (ydim, ydim)).contiguous()
    B = torch.zeros(lead + (xdim, 1), dtype=dtype, device=dev)
    spec = Actor(A=z, B=B, F=F, V=e, W=W, Q=e, R=torch.ones(lead + (1, 1), dtype=dtype, device=dev), T=T)
    return System(actor=spec, dynamics=spec)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_in_kernel_normal_draws_are_standard_normal(dtype):
    """Distributional checks of the draws (reference: jax.random.normal at lqg/system.py:102-105): moments, a
    Kolmogorov-Smirnov test, independence across steps / trials / components / eps-eta, seed sensitivity."""
    from scipy import stats
    T, n = 64, 4096
    m = _identity_noise_system(2, T, dtype)
    x, xh, y, u = m.simulate(1234, n=n, return_all=True)
    eps = x[:, 1:].double().cpu().numpy()                    # [n, T, 2]
    eta = (y - x[:, 1:]).double().cpu().numpy()
    for z in (eps, eta):
        flat = z.reshape(-1)
        assert abs(flat.mean()) < 5e-3 and abs(flat.var() - 1) < 1e-2
        assert abs(stats.skew(flat)) < 2e-2 and abs(stats.kurtosis(flat)) < 5e-2
        assert stats.kstest(flat[:200000], "norm").pvalue > 1e-3
        assert np.abs(z).max() < 7.0 and np.isfinite(z).all()
    c = lambda a, b: abs(np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1])
    assert c(eps[:, :-1], eps[:, 1:]) < 6e-3                 # consecutive steps
    assert c(eps[:-1], eps[1:]) < 6e-3                       # neighbouring trials
    assert c(eps[..., 0], eps[..., 1]) < 6e-3                # components of one block
    assert c(eps, eta) < 6e-3                                # process vs observation noise
    x2 = m.simulate(1235, n=n)
    assert c(x[:, 1:].double().cpu().numpy(), x2[:, 1:].double().cpu().numpy()) < 6e-3   # another seed: another stream
    assert torch.equal(m.simulate(1234, n=n), x)             # same seed: same trajectories


def test_in_kernel_draws_do_not_depend_on_batch_kernel_or_dtype():
    """A trajectory is a pure function of (seed, system index, trial index): the first trials of a larger batch, the lane
    kernels and the run-time-dims cooperative kernel, fp32 and fp64 all see the same normals; with B > 1 systems the draws of
    system s do not change when the number of trials or of systems does (advisor, round 3: they did, the counter was
    system * n_trials + trial)."""
    m = _identity_noise_system(2, 40, torch.float64)
    a = m.simulate(7, n=300)
    b = m.simulate(7, n=64)
    assert torch.equal(a[:64], b)
    m32 = _identity_noise_system(2, 40, torch.float32)
    assert torch.equal(m32.simulate(7, n=64).double(), b)     # (draws are made in fp32 and widened)
    # a shape with no lane kernel (x = b = 5, y = 4): k_coop_simulate<RNG>; its first two components' stream equals the x = 2 one
    from lqg_amd import _abi
    assert not _abi.load().lqg_kernel_supported(_abi.FAM_SIMULATE, _abi._dims_struct(dict(x=5, b=5, u=1, y=4, d=5)))
    m5 = _identity_noise_system(5, 40, torch.float64, ydim=4)
    c = m5.simulate(7, n=64)
    assert c.shape == (64, 41, 5) and torch.equal(c[..., :2], b)
    # many systems: counter = (trial, system, step, block) — system 0 of a batch is the single system; every system's
    # trajectories are invariant to n and to B; different systems draw different streams
    ms = _identity_noise_system(2, 40, torch.float64, n_sys=5)
    d = ms.simulate(7, n=60)                                  # [5, 60, 41, 2]
    assert torch.equal(d[0], a[:60])
    d17 = ms.simulate(7, n=17)
    assert torch.equal(d17, d[:, :17])                        # fewer trials: the same draws for every system
    m3 = _identity_noise_system(2, 40, torch.float64, n_sys=3)
    assert torch.equal(m3.simulate(7, n=60), d[:3])           # fewer systems: the same draws
    cc = np.corrcoef(d[1, :, 1:].reshape(-1).cpu().numpy(), d[2, :, 1:].reshape(-1).cpu().numpy())[0, 1]
    assert abs(cc) < 0.05 and not torch.equal(d[1], d[2])
    m5s = _identity_noise_system(5, 40, torch.float64, n_sys=3, ydim=4)      # the cooperative kernel agrees on systems too
    assert torch.equal(m5s.simulate(7, n=17)[..., :2], d[:3, :17])


def test_bounded_equals_subjective_under_a_shared_seed():
    """lqg/tests/lqg_test.py:69-93 with the in-kernel draws: both models have x = y = 2, hence the same counters."""
    import lqg_amd
    kw = dict(process_noise=1.0, sigma_target=6.0, action_cost=0.1, action_variability=0.5, sigma_cursor=3.0, T=500,
              device="cuda", dtype=torch.float64)
    x_b = lqg_amd.BoundedActor(**kw).simulate(rng_key=0, n=20)
    x_s = lqg_amd.SubjectiveActor(subj_noise=1.0, subj_vel_noise=0.0, **kw).simulate(rng_key=0, n=20)
    assert torch.allclose(x_b, x_s, rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------
# ABI 2: the MIXED problem LQG_F32_SYS64 straight through the C ABI (generic dense kernels: time-varying specs and affine cost
# terms included), against the fp64 C oracle and against the all-fp32 problem on the same inputs
@pytest.mark.parametrize("case", [0, 1, 2, 3, 7])
def test_mixed_precision_entry_on_random_dense_systems(oracle_lib, case, monkeypatch):
    import ctypes as C
    import oracle as OC
    from lqg_amd import _abi, _hip
    monkeypatch.setenv("LQG_NO_SPECIALIZE", "1")
    rng = np.random.default_rng(1000 + case)          # (the systems of test_random_systems_match_the_oracle: known well-posed)
    x, b, u, y = SHAPES[case % len(SHAPES)]
    T, n = int(rng.integers(5, 60)), 5
    tv, affine = bool(case & 1), bool(case & 2)
    actor, dyn = random_system(rng, x, b, u, y, T, tv, affine)
    rd = lambda spec: {k: v.astype(np.float32).astype(np.float64) for k, v in spec.items()}      # fp32-representable inputs
    actor, dyn = rd(actor), rd(dyn)
    X, _, _, _ = OC.simulate(actor, dyn, rng.standard_normal((n, T, x)), rng.standard_normal((n, T, y)))
    xs = X.astype(np.float32)
    ref = OC.log_likelihood(actor, dyn, xs.astype(np.float64))
    a64, d64 = to_spec(actor, torch.float64), to_spec(dyn, torch.float64)
    x32 = torch.as_tensor(xs, device="cuda")
    ln = _hip.Launch(a64, d64, d=x, n_trials=n, traj_dtype=torch.float32)
    lib = ln.require_gpu()
    assert ln.p.dtype == _abi.F32_SYS64
    ll = torch.empty(n, dtype=torch.float32, device="cuda")
    nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
    _abi.check(lib.lqg_log_likelihood(C.byref(ln.p), ln.traj(x32, False), C.c_void_p(ll.data_ptr()), 0, 1,
                                      C.c_void_p(ws.data_ptr()), nbytes, ln.stream()), "lqg_log_likelihood (mixed)")
    err_mixed = np.abs(np_(ll) - ref).max() / np.abs(ref).max()
    # the all-fp32 problem on the same inputs, same entry point
    monkeypatch.setenv("LQG_MIXED", "0")
    sys32 = lqg_amd.System(actor=to_spec(actor, torch.float32), dynamics=to_spec(dyn, torch.float32))
    err_f32 = np.abs(np_(sys32.log_likelihood(x32)) - ref).max() / np.abs(ref).max()
    assert err_mixed < 1e-6, (err_mixed, err_f32)
    assert err_mixed < 2.0 * err_f32 + 2e-7          # never meaningfully worse than the fp32 recursions
