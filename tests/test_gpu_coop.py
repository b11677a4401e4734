"""-m gpu: the COOPERATIVE kernels (lqg_amd/csrc/lqg_coop.hpp: one workgroup per system, run-time dims, LDS-staged)
against the golden vectors of the reference's own source and against the lane-per-system kernels.

LQG_COOP=1 forces the cooperative path for every shape (the library reads the variable per call); shapes without lane
kernels — the reference's DelayedSubjectiveActor, x=26 b=39 (lqg/tracking/delay.py:44-51) — take it by themselves.
Tolerances as tests/test_gpu_parity.py: fp64 ll 1e-10, gains / moments 1e-9; fp32 ll 1e-6, gains / moments 2e-5.
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, relerr
from gpu_common import np_, system_from_golden

pytestmark = pytest.mark.gpu

TOL = {torch.float64: dict(ll=1e-10, mat=1e-9), torch.float32: dict(ll=1e-6, mat=2e-5)}
ILL = {"pointmass_d4_T50"}          # observed block with condition number ~1e12 (tests/test_gpu_parity.py)


def _time_invariant_twin(actor, dyn, dtype):
    """The same system with stride-0 time axes (what the model constructors build) when every slice is equal, else None."""
    import lqg_amd
    from lqg_amd.system import Actor, Dynamics
    for spec in (actor, dyn):
        for f in ("A", "B", "F", "V", "W", "Q", "R"):
            if not np.array_equal(spec[f], np.broadcast_to(spec[f][:1], spec[f].shape)):
                return None
        if any(np.any(spec[f]) for f in ("q", "qf", "P", "r")) or not np.array_equal(spec["Qf"], spec["Q"][-1]):
            return None
    T = actor["A"].shape[0]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a[0]), dtype=dtype, device="cuda")
    a = Actor(A=t(actor["A"]), B=t(actor["B"]), F=t(actor["F"]), V=t(actor["V"]), W=t(actor["W"]), Q=t(actor["Q"]),
              R=t(actor["R"]), T=T)
    d = Dynamics(A=t(dyn["A"]), B=t(dyn["B"]), F=t(dyn["F"]), V=t(dyn["V"]), W=t(dyn["W"]), T=T)
    return lqg_amd.System(actor=a, dynamics=d)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("name", golden_names())
def test_cooperative_kernels_match_golden(name, dtype, monkeypatch):
    from lqg_amd import _abi, _hip
    from lqg_amd.belief import kf
    from lqg_amd.control import lqr

    monkeypatch.setenv("LQG_COOP", "1")
    monkeypatch.setenv("LQG_NO_DECOUPLE", "1")            # the joint problem, as the reference solves it
    g, actor, dyn = load_golden(name)
    tol = dict(TOL[dtype])
    if name in ILL:
        # (fp32: a caller who forces the cooperative kernels still gets the fp64 image of an ill-conditioned problem — moments and
        # log-likelihood route WIDE, plan.f32_needs_wide — never 1e-3-wrong numbers or NaN; compared with the golden vector at the
        # tolerance the ROUNDING OF THE INPUTS to fp32 leaves: 4.4e-6 on this model's log-likelihood, test_gpu_parity)
        tol = dict(ll=1e-7, mat=1e-6) if dtype == torch.float64 else dict(ll=2e-5, mat=2e-4)
    S0 = torch.as_tensor(g["Sigma0"], dtype=dtype, device="cuda") if "Sigma0" in g else None
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    n = x.shape[0]
    systems = [system_from_golden(actor, dyn, dtype)]      # [T, ...] stacks: the time-varying code path
    twin = _time_invariant_twin(actor, dyn, dtype)         # stride-0 time axes: the pipelined time-invariant path
    if twin is not None:
        systems.append(twin)
    for sys_ in systems:
        ln = _hip.Launch(sys_.actor, sys_.dynamics, d=x.shape[-1], n_trials=n, Sigma0=S0)
        import ctypes as C
        assert _abi.load().lqg_strategy(C.byref(ln.p)) == _abi.STRATEGY_COOP
        gains = lqr.backward(sys_.actor)
        K = kf.forward(sys_.actor, S0)
        assert relerr(np_(gains.L), g["L"]) < tol["mat"] and relerr(np_(gains.H), g["H"]) < tol["mat"]
        assert relerr(np_(K), g["K"]) < tol["mat"]
        if np.abs(g["l"]).max() > 0:
            assert relerr(np_(gains.l), g["l"]) < tol["mat"]
        else:
            assert np.abs(np_(gains.l)).max() == 0.0
        mu, Sig = _hip.conditional_moments(sys_.actor, sys_.dynamics, x, Sigma0=S0)
        assert relerr(np_(mu), g["mu"]) < tol["mat"] and relerr(np_(Sig), g["Sigma"][0]) < tol["mat"]
        ll = sys_.log_likelihood(x, Sigma0=S0)
        assert np.abs(np_(ll) / g["ll"] - 1).max() < tol["ll"]
        ll1 = sys_.log_likelihood(x[:1], Sigma0=S0)
        assert np.abs(np_(ll1) / g["ll"][:1] - 1).max() < tol["ll"]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_delayed_subjective_actor_runs_by_itself(dtype):
    """lqg_amd's DelayedSubjectiveActor (the reference's constructor lines, lqg/tracking/delay.py:44-51; delay 12 ->
    m = 65) needs no flag: it has no lane kernels, the cooperative ones take it.  Against the golden vector made from
    the reference's own classes; candidates batched on the leading axis as everywhere else."""
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    g, _, _ = load_golden("delay12_subjective1d_T30")
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    m = DelayedSubjectiveActor(T=30, device="cuda", dtype=dtype)
    assert (m.xdim, m.bdim) == (26, 39)
    ll = m.log_likelihood(x)
    assert np.abs(np_(ll) / g["ll"] - 1).max() < TOL[dtype]["ll"]
    bt = m.belief_tracking_distribution(x)
    assert relerr(np_(bt.loc), g["mu"][:, :, 26:]) < TOL[dtype]["mat"]
    # three candidates of sigma_target on the system axis; the middle one is the golden's
    sig = torch.tensor([4.0, 6.0, 9.0], dtype=dtype, device="cuda")
    mb = DelayedSubjectiveActor(T=30, sigma_target=sig, device="cuda", dtype=dtype)
    llb = mb.log_likelihood(x)
    assert llb.shape == (3, x.shape[0])
    assert np.abs(np_(llb[1]) / g["ll"] - 1).max() < TOL[dtype]["ll"]
    assert not np.allclose(np_(llb[0]), np_(llb[1]))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-6)], ids=["f64", "f32"])
def test_time_chunked_row_parallel_sweep_equals_the_one_pass_sweep(dtype, tol, monkeypatch):
    """The per-trial sweep of shapes without lane kernels (k_coop_trial_rows) cut along time (zero-state pass with the unit
    vectors of the transition matrices, boundary walk, density pass; csrc/lqg_coop.hpp): equal to the one-pass sweep to
    rounding for chunk counts that do and do not divide T, for 1 .. 40 trials, on the sequential and the time-parallel system
    sweeps; the default rule chunks few trials of few systems only; the golden vector (T = 30 < 64) is never chunked."""
    from lqg_amd.plan import LogLikelihoodPlan
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    T = 137
    sig = torch.tensor([5.0, 8.0], dtype=dtype, device="cuda")
    m = DelayedSubjectiveActor(T=T, sigma_target=sig, device="cuda", dtype=dtype)
    with torch.no_grad():
        x = m.simulate(3, n=40)[..., :2].contiguous()
    for scan in ("0", "1"):
        monkeypatch.setenv("LQG_SCAN", scan)
        monkeypatch.setenv("LQG_COOP_TRIAL_CHUNKS", "0")
        ref = m.log_likelihood(x).clone()
        for chunks in ("2", "5", "13", "34", None):
            if chunks is None:
                monkeypatch.delenv("LQG_COOP_TRIAL_CHUNKS")
            else:
                monkeypatch.setenv("LQG_COOP_TRIAL_CHUNKS", chunks)
            for n in (1, 2, 7, 40):
                got = LogLikelihoodPlan(m, x[:, :n].contiguous()).run().clone()
                assert got.shape == (2, n)
                assert float((got / ref[:, :n] - 1).abs().max()) < tol, (scan, chunks, n)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-6)], ids=["f64", "f32"])
def test_row_lists_of_the_mean_update_equal_the_dense_rows(dtype, tol, monkeypatch):
    """k_coop_trial_rows walks the run-time row lists of the operator's mean-update block (k_coop_trial_lists: the union of the
    non-zero columns over the horizon; 9 % of m^2 for the delay models): equal to the dense rows (LQG_COOP_SPARSE=0 — the terms
    left out are exact zeros) in one pass, cut along time, with many trials per workgroup, and for the materialised means."""
    from lqg_amd.plan import LogLikelihoodPlan
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    T = 150
    sig = torch.tensor([5.0, 8.0, 13.0], dtype=dtype, device="cuda")
    m = DelayedSubjectiveActor(T=T, sigma_target=sig, device="cuda", dtype=dtype)
    with torch.no_grad():
        x = m.simulate(3, n=70)[..., :2].contiguous()
    monkeypatch.setenv("LQG_SCAN", "0")
    for chunks, n in (("0", 70), ("0", 3), ("9", 5), ("9", 1)):
        monkeypatch.setenv("LQG_COOP_TRIAL_CHUNKS", chunks)
        xs = x[:, :n].contiguous()
        monkeypatch.setenv("LQG_COOP_SPARSE", "0")
        dense = LogLikelihoodPlan(m, xs).run().clone()
        monkeypatch.setenv("LQG_COOP_SPARSE", "1")
        got = LogLikelihoodPlan(m, xs).run().clone()
        assert torch.isfinite(got).all() and float((got / dense - 1).abs().max()) < tol, (chunks, n)
    monkeypatch.setenv("LQG_COOP_SPARSE", "0")
    mu0, _ = m.conditional_moments(x[:, 0].contiguous())
    monkeypatch.setenv("LQG_COOP_SPARSE", "1")
    mu1, _ = m.conditional_moments(x[:, 0].contiguous())
    assert float((mu1 - mu0).abs().max()) <= tol * float(mu0.abs().max())


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-6)], ids=["f64", "f32"])
@pytest.mark.parametrize("model", ["pointmass", "hand1d", "subjective2d"])
def test_cooperative_and_lane_kernels_agree_on_a_batch(model, dtype, tol, monkeypatch):
    """Same inputs through both strategies: B candidates x n trials (the few-systems regime the cooperative kernels are for)."""
    import lqg_amd
    from lqg_amd import workload
    dev = torch.device("cuda")
    if model == "pointmass":
        av = torch.linspace(0.2, 1.5, 7, dtype=dtype, device=dev)
        m = lqg_amd.PointMassBoundedActor(T=120, action_variability=av, device=dev, dtype=dtype)
        d = 2
    elif model == "hand1d":
        import sys, os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench_configs import hand2d_system
        m = hand2d_system(150, dev, dtype)
        d = 4
    else:
        m, _ = workload.headline_system(5, 200, seed=3, device=dev, dtype=dtype)
        d = 4
    with torch.no_grad():
        x = m.simulate(4, n=33)[..., :d].contiguous()
    monkeypatch.setenv("LQG_COOP", "0")
    ref = m.log_likelihood(x)
    monkeypatch.setenv("LQG_COOP", "1")
    got = m.log_likelihood(x)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert float(((got.double() / ref.double()) - 1).abs().max()) < tol
