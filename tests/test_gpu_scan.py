"""-m gpu: the TIME-PARALLEL system sweeps (lqg_amd/csrc/lqg_scan.hpp: Riccati, Kalman and moment recursions as associative
scans over the time axis) against the golden vectors of the reference's own source and against the sequential kernels.

LQG_SCAN=1 selects the scan path wherever it is defined (no affine cost terms, eigenvalue floor provably inactive,
u, y, d <= 4); by default it serves few systems with many steps.  The scans run in fp64 whatever the problem dtype, so the
fp32 tolerances are those of the fp32 per-trial sweep alone."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, relerr
from gpu_common import np_, system_from_golden

pytestmark = pytest.mark.gpu

TOL = {torch.float64: dict(ll=1e-10, mat=1e-9), torch.float32: dict(ll=1e-6, mat=2e-5)}
AFFINE = {"timevarying_T30"}                    # q, r, P, qf non-zero: not a scan case (falls back to the sequential kernels)
ILL = {"pointmass_d4_T50"}


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("name", [n for n in golden_names() if n not in AFFINE])
def test_scan_sweeps_match_golden(name, dtype, monkeypatch):
    from lqg_amd import _hip
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_SCAN", "1")
    monkeypatch.setenv("LQG_NO_DECOUPLE", "1")           # the joint problem, as the reference solves it
    g, actor, dyn = load_golden(name)
    tol = dict(TOL[dtype])
    if name in ILL:
        # (fp32: the plan evaluates an ill-conditioned fp32 problem over its fp64 image on the forced scan route too, plan.f32_needs_wide)
        tol = dict(ll=5e-6, mat=1e-6) if dtype == torch.float64 else dict(ll=2e-5, mat=2e-4)      # cond 5.6e8 of the observed noise block costs the scans ~4 digits more than the sequential
                                            # recursion; the default rule keeps such systems off the scan (plan.SCAN_MAX_COND)
    sys_ = system_from_golden(actor, dyn, dtype)
    S0 = torch.as_tensor(g["Sigma0"], dtype=dtype, device="cuda") if "Sigma0" in g else None
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    plan = LogLikelihoodPlan(sys_, x, Sigma0=S0)
    assert all(wk["scan"] for wk in plan.work), plan.description
    ll = plan.run().clone()
    assert np.abs(np_(ll) / g["ll"] - 1).max() < tol["ll"]
    ll1 = LogLikelihoodPlan(sys_, x[:1], Sigma0=S0).run().clone()
    assert np.abs(np_(ll1) / g["ll"][:1] - 1).max() < tol["ll"]
    mu, Sig = _hip.conditional_moments(sys_.actor, sys_.dynamics, x, Sigma0=S0)
    assert relerr(np_(mu), g["mu"]) < tol["mat"] and relerr(np_(Sig), g["Sigma"][0]) < tol["mat"]


@pytest.mark.parametrize("model", ["pointmass", "hand2d", "bounded_candidates", "subjective2d"])
def test_scan_and_sequential_sweeps_agree_at_full_horizon(model, monkeypatch):
    """T = 500 / 1000, the shapes of BASELINE configs 2 and 4 and a handful of candidates: scan vs sequential, fp64 1e-10;
    for an fp32 problem the scan path (fp64 operators) is at least as close to fp64 as the sequential fp32 sweeps."""
    import lqg_amd
    from lqg_amd.plan import LogLikelihoodPlan
    dev = torch.device("cuda")
    if model == "pointmass":
        m = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=torch.float64)
        d, n = 2, 512
    elif model == "hand2d":
        import os, sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench_configs import hand2d_system
        m = hand2d_system(1000, dev, torch.float64)
        d, n = 4, 256
    elif model == "bounded_candidates":
        sig = torch.tensor([4.0, 9.0, 15.0, 30.0], dtype=torch.float64, device=dev)
        m = lqg_amd.BoundedActor(T=700, sigma_target=sig, action_cost=0.2, device=dev, dtype=torch.float64)
        d, n = 2, 64
    else:
        m = lqg_amd.SubjectiveActor(dim=2, T=300, device=dev, dtype=torch.float64)
        d, n = 4, 128
    with torch.no_grad():
        x = m.simulate(4, n=n)[..., :d].contiguous()
    monkeypatch.setenv("LQG_SCAN", "0")
    ref = m.log_likelihood(x).clone()
    ref32 = m.to(torch.float32).log_likelihood(x.float()).clone()
    monkeypatch.setenv("LQG_SCAN", "1")
    p = LogLikelihoodPlan(m, x)
    assert all(wk["scan"] for wk in p.work)
    got = p.run().clone()
    assert got.shape == ref.shape and float((got / ref - 1).abs().max()) < 1e-10
    got32 = LogLikelihoodPlan(m.to(torch.float32), x.float()).run().clone()
    e_scan = float((got32.double() / ref - 1).abs().max())
    e_seq = float((ref32.double() / ref - 1).abs().max())
    assert e_scan < max(2e-6, 1.5 * e_seq), (e_scan, e_seq)


@pytest.mark.parametrize("order,T", [("", 500), ("1", 500), ("2", 500), ("0", 500), ("1", 37), ("1", 64), ("1", 129), ("2", 129), ("1", 301)],
                         ids=["rule-T500", "work_efficient-T500", "brent_kung-T500", "hillis_steele-T500", "work_efficient-T37", "work_efficient-T64",
                              "work_efficient-T129", "brent_kung-T129", "work_efficient-T301"])
def test_delay_model_windows_of_64_against_the_sequential_kernels(monkeypatch, order, T):
    """The reference's largest model (DelayedSubjectiveActor, lqg/tracking/delay.py:44-51: x = 26, b = 39) at T = 500:
    windows of 39 (Riccati, Kalman) and 63 (moment recursion) on k_scan_level_rt against the cooperative sequential
    sweeps, one system and a handful of candidates — in every order of the levels (Hillis-Steele ping-pong; Brent-Kung in place;
    Brent-Kung around a ping-pong scan of the block totals, the default: lqg_scan_inst.hip run_scan) and, for the strided index
    maps of the latter two, at horizons that are not powers of two."""
    from lqg_amd.plan import LogLikelihoodPlan
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    dev = torch.device("cuda")
    monkeypatch.setenv("LQG_SCAN_ORDER", order)
    if T != 500:
        monkeypatch.setenv("LQG_SCAN_MIN_STEPS", "8")
    for kw in (dict(), dict(sigma_target=torch.tensor([3.0, 6.0, 11.0], dtype=torch.float64, device=dev))):
        m = DelayedSubjectiveActor(T=T, device=dev, dtype=torch.float64, **kw)
        with torch.no_grad():
            x = m.simulate(5, n=6)[..., :2].contiguous()
        monkeypatch.setenv("LQG_SCAN", "0")
        p0 = LogLikelihoodPlan(m, x)
        assert not any(wk["scan"] for wk in p0.work)
        ref = p0.run().clone()
        monkeypatch.delenv("LQG_SCAN")
        p1 = LogLikelihoodPlan(m, x)
        assert all(wk["scan"] for wk in p1.work), p1.description          # (the default for few long systems)
        got = p1.run().clone()
        assert got.shape == ref.shape and float((got / ref - 1).abs().max()) < 1e-10
        got32 = LogLikelihoodPlan(m.to(torch.float32), x.float()).run().clone()
        assert float((got32.double() / ref - 1).abs().max()) < 1e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-6)], ids=["f64", "f32"])
def test_delay_model_evaluation_replays_as_one_graph(dtype, tol):
    """The whole evaluation of the delay model — time-parallel sweeps on register-resident windows, time-chunked row-parallel
    per-trial sweep — captured as one hipGraph by the inference loops' evaluator and replayed at other parameters."""
    from lqg_amd.infer import graphed
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    T = 150
    with torch.no_grad():
        x = DelayedSubjectiveActor(T=T, device="cuda", dtype=dtype).simulate(8, n=6)[..., :2].contiguous()
    names = ["c", "action_variability", "sigma_target", "sigma_cursor"]
    gl = graphed.GraphedLogLik(x, DelayedSubjectiveActor, names, 1)
    assert gl.capture()
    for th in ([0.5, 0.5, 6.0, 3.0], [0.3, 0.7, 9.0, 2.0]):
        out = gl(torch.tensor([th], dtype=dtype, device="cuda")).clone()
        m = DelayedSubjectiveActor(T=T, device="cuda", dtype=dtype, **dict(zip(names, th)))
        eager = m.log_likelihood(x).double().sum()
        assert float((out[0] / eager - 1).abs()) < tol


def test_scan_is_the_default_for_one_long_system_and_not_for_batches():
    import lqg_amd
    from lqg_amd.plan import LogLikelihoodPlan
    m = lqg_amd.PointMassBoundedActor(T=500, device="cuda", dtype=torch.float32)
    x = m.simulate(1, n=300)[..., :2].contiguous()
    assert all(wk["scan"] for wk in LogLikelihoodPlan(m, x).work)
    short = lqg_amd.PointMassBoundedActor(T=40, device="cuda", dtype=torch.float32)
    assert not any(wk["scan"] for wk in LogLikelihoodPlan(short, short.simulate(1, n=300)[..., :2].contiguous()).work)
    sig = torch.linspace(3.0, 30.0, 64, device="cuda")
    many = lqg_amd.BoundedActor(T=500, sigma_target=sig, device="cuda")
    assert not any(wk["scan"] for wk in LogLikelihoodPlan(many, many.simulate(1, n=4)[0]).work)
    # an ill-conditioned observed block (all four point-mass states observed, cond((V V')[:4, :4]) = 5.6e8) stays on the
    # sequential sweeps by default (plan.SCAN_MAX_COND)
    ill = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device="cuda", dtype=torch.float64)
    assert not any(wk["scan"] for wk in LogLikelihoodPlan(ill, ill.simulate(1, n=8).contiguous()).work)


@pytest.mark.parametrize("case", range(8))
def test_time_parallel_path_on_random_dense_systems(oracle_lib, case, monkeypatch):
    """Random dense systems (no structure), time-invariant and genuinely time-varying, partial observation, custom Sigma0,
    horizons that do not divide into the chunks: scans + time-chunked per-trial sweep against the fp64 C oracle."""
    import lqg_amd
    from gpu_common import to_spec
    from lqg_amd.plan import LogLikelihoodPlan
    from test_gpu_random import SHAPES, random_system
    rng = np.random.default_rng(7000 + case)
    x, b, u, y = SHAPES[case % len(SHAPES)]
    T = int(rng.integers(70, 260))
    tv = bool(case & 1)
    actor, dyn = random_system(rng, x, b, u, y, T, tv, affine=False)
    for spec in (actor, dyn):            # open-loop stable: over 70-260 steps a spectral radius > 1 makes the problem itself
        rho = max(np.abs(np.linalg.eigvals(spec["A"][t])).max() for t in range(0, T, 7))    # ill-conditioned (every path,
        spec["A"] = spec["A"] * min(1.0, 0.97 / rho)                                         # sequential included, then
    d = x if case % 3 else (2 if x == 4 else x)                                              # agrees with the oracle to 1e-3 only)
    S0 = None
    if case % 4 == 3:
        M = rng.standard_normal((b, b))
        S0 = M @ M.T / b + 0.3 * np.eye(b)
    n = 5
    X, _, _, _ = oracle_lib.simulate(actor, dyn, rng.standard_normal((n, T, x)), rng.standard_normal((n, T, y)), Sigma0=S0)
    xs = X[..., :d]
    ref_ll = oracle_lib.log_likelihood(actor, dyn, xs, S0)
    ref_mu, ref_Sig = oracle_lib.conditional_moments(actor, dyn, xs, S0)
    sys_ = lqg_amd.System(actor=to_spec(actor, torch.float64), dynamics=to_spec(dyn, torch.float64))
    S0t = None if S0 is None else torch.as_tensor(S0, dtype=torch.float64, device="cuda")
    xt = torch.as_tensor(xs, dtype=torch.float64, device="cuda")
    monkeypatch.setenv("LQG_SCAN", "1")
    monkeypatch.setenv("LQG_FUSE_TRIALS_MAX", "0")
    for chunks in ("", "3", "11"):
        if chunks:
            monkeypatch.setenv("LQG_TRIAL_CHUNKS", chunks)
        plan = LogLikelihoodPlan(sys_, xt, Sigma0=S0t)
        assert all(wk["scan"] for wk in plan.work), plan.description
        ll = np_(plan.run().clone())
        assert np.abs(ll - ref_ll).max() < 1e-9 * np.abs(ref_ll).max(), (chunks, np.abs(ll - ref_ll).max())
    from lqg_amd import _hip
    mu, Sig = _hip.conditional_moments(sys_.actor, sys_.dynamics, xt, Sigma0=S0t)
    assert np.abs(np_(mu) - ref_mu).max() < 1e-8 * max(1.0, np.abs(ref_mu).max())
    assert np.abs(np_(Sig) - ref_Sig).max() < 1e-8 * max(1.0, np.abs(ref_Sig).max())


@pytest.mark.parametrize("case", range(7))
def test_windows_of_25_to_64_on_random_dense_systems(oracle_lib, case, monkeypatch):
    """Dense random systems whose scan windows exceed LDS (k_scan_level_rt: elimination in registers, run-time n), sizes
    that are not multiples of the 8 rows a wave owns, time-varying specs, partial observation — against the fp64 C oracle."""
    import lqg_amd
    from gpu_common import to_spec
    from lqg_amd.plan import LogLikelihoodPlan
    from test_gpu_random import random_system
    rng = np.random.default_rng(7100 + case)
    # (the last three: windows that fill their 16 x 16 tiles exactly — 32, 48 and 64 — and the largest b the C oracle holds)
    x, b, u, y, d = [(9, 25, 2, 3, 3), (12, 31, 1, 2, 1), (26, 39, 1, 1, 2), (18, 33, 3, 4, 4),
                     (9, 25, 1, 2, 2), (18, 32, 2, 2, 2), (26, 40, 1, 1, 2)][case]
    T = int(rng.integers(40, 90))
    actor, dyn = random_system(rng, x, b, u, y, T, bool(case & 1), affine=False)
    for spec in (actor, dyn):
        rho = max(np.abs(np.linalg.eigvals(spec["A"][t])).max() for t in range(0, T, 7))
        spec["A"] = spec["A"] * min(1.0, 0.97 / rho)
    n = 3
    X, _, _, _ = oracle_lib.simulate(actor, dyn, rng.standard_normal((n, T, x)), rng.standard_normal((n, T, y)))
    xs = X[..., :d]
    ref_ll = oracle_lib.log_likelihood(actor, dyn, xs, None)
    ref_mu, ref_Sig = oracle_lib.conditional_moments(actor, dyn, xs, None)
    sys_ = lqg_amd.System(actor=to_spec(actor, torch.float64), dynamics=to_spec(dyn, torch.float64))
    xt = torch.as_tensor(xs, dtype=torch.float64, device="cuda")
    monkeypatch.setenv("LQG_SCAN", "1")
    plan = LogLikelihoodPlan(sys_, xt)
    assert all(wk["scan"] for wk in plan.work), plan.description
    ll = np_(plan.run().clone())
    assert np.abs(ll - ref_ll).max() < 1e-9 * np.abs(ref_ll).max(), np.abs(ll - ref_ll).max()
    from lqg_amd import _hip
    mu, Sig = _hip.conditional_moments(sys_.actor, sys_.dynamics, xt)
    assert np.abs(np_(mu) - ref_mu).max() < 1e-8 * max(1.0, np.abs(ref_mu).max())
    assert np.abs(np_(Sig) - ref_Sig).max() < 1e-8 * max(1.0, np.abs(ref_Sig).max())


def test_point_mass_setup_kernel_equals_the_torch_construction(monkeypatch):
    """csrc/lqg_setup.hip (`lqg_point_mass_setup`: matrix exponentials, Van Loan block, eigenvalue clipping, upper Cholesky in
    one kernel) against the torch.linalg construction of lqg_amd/tracking/point_mass.py (the reference's own sequence of
    operations, point_mass.py:50-144) over a wide parameter range; and the no-grad / grad routes give the same model."""
    import lqg_amd
    g = torch.Generator(device="cpu").manual_seed(5)
    B = 300
    u = lambda lo, hi: torch.exp(torch.rand(B, generator=g, dtype=torch.float64) * (np.log(hi) - np.log(lo)) + np.log(lo)).cuda()
    kw = dict(damping=u(0.02, 5.0), m=u(0.2, 5.0), tau=u(0.0015, 0.5), action_variability=u(1e-3, 2.0),
              sigma_target=u(1.0, 30.0), sigma_cursor=u(0.5, 10.0), action_cost=u(1e-3, 1.0))
    a = lqg_amd.PointMassBoundedActor(T=10, device="cuda", dtype=torch.float64, **kw)
    monkeypatch.setenv("LQG_SETUP_KERNEL", "0")
    b = lqg_amd.PointMassBoundedActor(T=10, device="cuda", dtype=torch.float64, **kw)
    for f in ("A", "B", "V"):
        x, y = getattr(a.actor, f)[:, 0], getattr(b.actor, f)[:, 0]
        # (the Van Loan block passes through exp(+dt / tau) ~ 6e4 at tau = 0.0015: both routes carry ~1e-11 of cancellation)
        scale = y.abs().amax((-1, -2), keepdim=True).clamp_min(1e-300)          # per candidate
        assert float(((x - y).abs() / scale).max()) < 1e-10, f
    # V V' is what the path consumes: relative agreement of the whole covariance
    VVa, VVb = a.actor.V @ a.actor.V.transpose(-1, -2), b.actor.V @ b.actor.V.transpose(-1, -2)
    assert float(((VVa - VVb).abs().amax((-1, -2)) / VVb.abs().amax((-1, -2))).max()) < 1e-10


@pytest.mark.parametrize("model", ["bounded", "pointmass", "subjective1d"])
def test_scan_levels_packed_sub_wave_for_many_candidates(model, monkeypatch):
    """24 candidates x 500 steps: a level holds > 8192 elements, so the 2x2 .. 6x6 windows run several per wave
    (lqg_scan_inst.hip launch_level): same result as the sequential sweeps."""
    import lqg_amd
    from lqg_amd.plan import LogLikelihoodPlan
    dev = torch.device("cuda")
    sig = torch.linspace(3.0, 30.0, 24, dtype=torch.float64, device=dev)
    if model == "bounded":
        m = lqg_amd.BoundedActor(T=500, sigma_target=sig, device=dev, dtype=torch.float64)
    elif model == "pointmass":
        m = lqg_amd.PointMassBoundedActor(T=500, sigma_target=sig, action_variability=0.5, device=dev, dtype=torch.float64)
    else:
        m = lqg_amd.SubjectiveActor(T=500, sigma_target=sig, device=dev, dtype=torch.float64)
    with torch.no_grad():
        x = m.simulate(2, n=12)[0][..., :2].contiguous()
    monkeypatch.setenv("LQG_SCAN", "0")
    ref = LogLikelihoodPlan(m, x).run().clone()
    monkeypatch.setenv("LQG_SCAN", "1")
    p = LogLikelihoodPlan(m, x)
    assert all(wk["scan"] for wk in p.work)
    got = p.run().clone()
    assert got.shape == (24, 12) and float((got / ref - 1).abs().max()) < 1e-10


def test_default_rule_against_sequential_on_random_zoo_models():
    """scripts/fuzz_time_parallel.py in 'default' mode: 80 random (class, parameters over wide log-uniform ranges, horizon,
    trials, candidates, chunk count) cases; wherever the default rule takes the time-parallel path the result equals the
    sequential kernels to 1e-9 — the ill-conditioned fully observed point mass must have been kept on the sequential sweeps."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LQG_")}
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_time_parallel.py"), "11", "80", "default"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("mode default")][-1]
    assert last.endswith("failures []"), last
    assert int(last.split("scan path used in")[1].split()[0]) >= 20, last
