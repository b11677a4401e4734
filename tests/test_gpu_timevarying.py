"""`System.log_likelihood` on specs that VARY IN TIME, served by the pattern libraries (VERDICT r05 task 4).

The reference's data model is (T, ...)-stacked (lqg/spec.py:5-19, lqg/utils.py:10-35; per-step arrays come from user code): until
round 5 such a model dropped to the dense generic kernels (fp64: 3.1 M solves/s, 87x wasted traffic).  Now `lqg_log_likelihood_sp`
runs k_riccati_tv_sp -> k_forward_tv_sp (csrc/lqg_sp_entry.hpp: run_sp_tv) whenever the specs keep one sparsity pattern over
systems and steps: one trial in-lane; several trials and the mixed mode through the operator stream; the cross cost P in the
Riccati step; q, qf, r ignored (they only move the affine gain l, which the likelihood does not read: lqg/system.py:169-181)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_common import np_, system_from_golden

pytestmark = pytest.mark.gpu


def _plan_ll(system, x, Sigma0=None):
    """log-likelihood through a LogLikelihoodPlan + which entry served every work item AFTER the run (a pattern library that
    refuses a problem is replaced by the generic entry on the first call: `specialised` then reads False)."""
    from lqg_amd.plan import LogLikelihoodPlan
    plan = LogLikelihoodPlan(system, x, Sigma0=Sigma0)
    ll = plan.run().clone()
    torch.cuda.synchronize()
    return ll, [bool(wk["specialised"]) for wk in plan.work], [bool(wk["mixed"]) for wk in plan.work]


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-6)], ids=["f64", "f32"])
def test_golden_timevarying_T30_runs_on_its_pattern_library(dtype, tol):
    """Golden case timevarying_T30 (every field time-varying, q / qf / P / r non-zero, custom Sigma0, one system x 3 trials;
    reference-generated: oracle/gen_golden.py): all three trials (operator stream; fp32: the mixed mode), two trials, and one
    trial (in-lane) — served by the pattern library, equal to the reference's log-likelihood."""
    g, actor, dyn = load_golden("timevarying_T30")
    m = system_from_golden(actor, dyn, dtype)
    S0 = torch.as_tensor(g["Sigma0"], dtype=dtype, device="cuda")
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    for sel in (slice(0, 3), slice(0, 2), slice(1, 2)):
        ll, spec, mixed = _plan_ll(m, x[sel], S0)
        assert all(spec), (sel, spec)
        assert np.abs(np_(ll) / g["ll"][sel] - 1).max() < tol, sel
        if dtype == torch.float32 and sel == slice(0, 3):
            assert all(mixed)                                        # fp64 system sweeps, operators rounded once, fp32 per-trial sweep
    # the public call takes the same route
    assert np.abs(np_(m.log_likelihood(x, Sigma0=S0)) / g["ll"] - 1).max() < tol


def _tv_batch(B, T, dtype, dev="cuda", psd=False):
    import bench_m2
    system, base = bench_m2.m2_system(torch.device(dev), dtype, B, T, psd=psd)
    from lqg_amd import workload
    x = workload.pack_trials(workload.simulate_one_trial_each(base, seed=5))          # [B, 1, T+1, 4]
    return system, x


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-6)], ids=["f64", "f32"])
def test_time_varying_batch_of_2p17_systems_through_the_specialised_entry(dtype, tol, oracle_lib, monkeypatch):
    """2^17 systems whose every non-zero spec entry moves in time and over the systems ([T][element][system] storage, the M2
    workload of bench_m2.py at T = 40): the plan serves them from the pattern library; sampled systems against the C oracle,
    a block of systems against the dense generic kernels."""
    from lqg_amd import workload
    B, T = 1 << 17, 40
    system, x = _tv_batch(B, T, dtype)
    ll, spec, _ = _plan_ll(system, x)
    assert all(spec) and ll.shape == (B, 1) and bool(torch.isfinite(ll).all())
    rng = np.random.default_rng(0)
    one_of = lambda sp, j: {f: (getattr(sp, f)[j] if getattr(sp, f).dim() == workload._batched_ndim(f)
                                else getattr(sp, f)).double().cpu().numpy() for f in sp._fields}
    for j in sorted({int(v) for v in rng.integers(0, B, size=12)}):
        ref = oracle_lib.log_likelihood(one_of(system.actor, j), one_of(system.dynamics, j), x[j].double().cpu().numpy(), None)
        assert abs(float(ll[j, 0]) / float(ref[0]) - 1) < tol, j
    # the dense generic kernels on the first 4096 systems
    sub = workload.slice_system(system, 0, 4096)
    monkeypatch.setenv("LQG_NO_SPECIALIZE", "1")
    ll_d, spec_d, _ = _plan_ll(sub, x[:4096])
    monkeypatch.delenv("LQG_NO_SPECIALIZE")
    assert not any(spec_d)
    assert float((ll_d / ll[:4096] - 1).abs().max()) < (1e-11 if dtype == torch.float64 else 2e-6)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-6)], ids=["f64", "f32"])
def test_time_varying_model_with_costs_that_stay_costs_decouples(dtype, tol, oracle_lib):
    """The same batch with Q_t, R_t moved by a congruence (positive semi-definite at every step, as the costs of a model whose
    parameters move in time are): the eigenvalue floor is provably inactive, the two 1-D components decouple exactly and each
    runs on the (x, b, u, y, d) = (2, 3, 1, 2, 2) time-varying pattern; [T][element][system] storage survives the split."""
    from lqg_amd import workload
    from lqg_amd.plan import LogLikelihoodPlan
    B, T = 1 << 17, 40
    system, x = _tv_batch(B, T, dtype, psd=True)
    plan = LogLikelihoodPlan(system, x)
    ll = plan.run().clone()
    torch.cuda.synchronize()
    assert len(plan.work) == 2 and all(wk["specialised"] for wk in plan.work)
    assert all(wk["dims"] == (2, 3, 1, 2, 2) for wk in plan.work)
    for wk in plan.work:                                              # system index fastest in every time-varying field of a component
        assert wk["ln"].p.actor.A.sb == 1 and wk["ln"].p.dynamics.A.sb == 1 and wk["ln"].p.actor.A.st != 0
    rng = np.random.default_rng(1)
    one_of = lambda sp, j: {f: (getattr(sp, f)[j] if getattr(sp, f).dim() == workload._batched_ndim(f)
                                else getattr(sp, f)).double().cpu().numpy() for f in sp._fields}
    for j in sorted({int(v) for v in rng.integers(0, B, size=12)}):
        ref = oracle_lib.log_likelihood(one_of(system.actor, j), one_of(system.dynamics, j), x[j].double().cpu().numpy(), None)
        assert abs(float(ll[j, 0]) / float(ref[0]) - 1) < tol, j


def test_time_varying_candidates_with_several_trials_each():
    """Time-varying specs with a candidate axis AND several trials per candidate (operator stream written by k_forward_tv_sp<FUSED =
    false>, per-trial sweep of the pattern library): equal to one single-trial evaluation per trial, fp64 to rounding."""
    B, T, n = 300, 50, 7
    system, x1 = _tv_batch(B, T, torch.float64)
    xs = torch.cat([x1 * (1.0 + 0.01 * k) for k in range(n)], dim=1).contiguous()       # [B, n, T+1, 4]
    ll, spec, _ = _plan_ll(system, xs)
    assert all(spec) and ll.shape == (B, n)
    for k in (0, 3, 6):
        ll1, spec1, _ = _plan_ll(system, xs[:, k:k + 1].contiguous())
        assert all(spec1)
        assert float((ll[:, k] / ll1[:, 0] - 1).abs().max()) < 1e-12
