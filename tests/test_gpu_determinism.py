"""-m gpu: repeat-run BITWISE determinism of the kernels whose lanes share LDS across waves with hand-placed barriers —
k_coop_riccati / k_coop_forward / k_coop_trial_rows (+ k_coop_trial_fix), k_scan_level / k_scan_level_rt / k_scan_lane and
the per-step builders / finalisers around them, k_trial_zs / k_trial_fix / k_trial_ll / k_trial_sum.

The lane kernels are race-free by construction (one system per lane, no shared memory); these are not: a missing barrier
or an LDS buffer reused one stage early shows up as a result that depends on the wave scheduling of a particular run.
Each case builds its plan once and replays it REPS times on the same inputs; every replay must equal the first one bit
for bit (compared on the device, one read-back per case).  No GPU AddressSanitizer / XNACK exists on this pool — the CPU
restatement runs under ASan + UBSan instead (tests/test_oracle.py)."""
import pytest
import torch

from conftest import load_golden
from gpu_common import system_from_golden

pytestmark = pytest.mark.gpu
REPS = 200


def _replay(plan, reps=REPS):
    first = plan.run().clone()
    differs = torch.zeros((), dtype=torch.bool, device=first.device)
    for _ in range(reps - 1):
        out = plan.run()
        differs |= (out.view(torch.int32 if out.dtype == torch.float32 else torch.int64)
                    != first.view(torch.int32 if first.dtype == torch.float32 else torch.int64)).any()
    assert torch.isfinite(first).all()
    assert not bool(differs), "a replay differed bitwise from the first run"
    return first


def _delay12(dtype, T):
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    m = DelayedSubjectiveActor(T=T, device="cuda", dtype=dtype)
    with torch.no_grad():
        x = m.simulate(5, n=6)[..., :2].contiguous()
    return m, x


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("path", ["coop_one_pass", "coop_chunked", "scan_rt", "scan_rt_in_place"])
def test_delay12_m65_replays_are_bitwise_identical(path, dtype, monkeypatch):
    """The reference's largest model (x = 26, b = 39, m = 65): 1024-lane workgroups, hybrid LDS / L2 arena, run-time sparsity
    lists (sequential cooperative sweeps); windows of 39 / 63 in registers of 16 waves, DPP pivot search, fp64 MFMA products
    (k_scan_level_rt), its levels ping-pong (Hillis-Steele) and IN PLACE (Brent-Kung: a workgroup overwrites one of its own
    operands); the row-parallel per-trial sweep in one pass and cut along time."""
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_SCAN", "1" if path.startswith("scan_rt") else "0")
    monkeypatch.setenv("LQG_SCAN_ORDER", "1" if path == "scan_rt_in_place" else "0")
    monkeypatch.setenv("LQG_COOP_TRIAL_CHUNKS", "0" if path == "coop_one_pass" else "7")
    m, x = _delay12(dtype, 120)
    plan = LogLikelihoodPlan(m, x)
    assert all(wk["scan"] == path.startswith("scan_rt") for wk in plan.work), plan.description
    _replay(plan, REPS if path.startswith("scan_rt") else 60)          # (the sequential sweeps of m = 65 cost ~5 ms per replay)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("name", ["tutorial_lqg_T100", "subjective2d_T60", "pointmass_d2_T50", "hand2d_T40", "timevarying_T30"])
def test_cooperative_kernels_replay_bitwise(name, dtype, monkeypatch):
    """k_coop_riccati / k_coop_forward (fixed-dims and run-time-dims instantiations, LDS arena) + k_coop_trial on the golden
    systems (joint problem, several trials)."""
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_COOP", "1")
    monkeypatch.setenv("LQG_SCAN", "0")
    monkeypatch.setenv("LQG_NO_DECOUPLE", "1")
    monkeypatch.setenv("LQG_F32_WIDE", "0")
    g, actor, dyn = load_golden(name)
    sys_ = system_from_golden(actor, dyn, dtype)
    x = torch.as_tensor(g["x"], dtype=dtype, device="cuda")
    x = torch.cat([x] * 3, dim=0)
    plan = LogLikelihoodPlan(sys_, x)
    assert all(wk["coop"] for wk in plan.work), plan.description
    _replay(plan)


@pytest.mark.parametrize("model", ["bounded", "subjective2d", "pointmass", "hand2d", "bounded_9_candidates"])
def test_time_parallel_path_replays_bitwise(model, monkeypatch):
    """k_scan_lane (windows of 2 / 3: one launch for all levels), k_scan_level<N> (windows of 4 .. 10, sub-wave packing with
    several candidates), their builders / finalisers, and the time-chunked per-trial sweep k_trial_zs -> k_trial_fix ->
    k_trial_ll -> k_trial_sum."""
    import os
    import sys
    import lqg_amd
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_SCAN", "1")
    monkeypatch.setenv("LQG_TRIAL_CHUNKS", "6")
    dev, dt = torch.device("cuda"), torch.float32
    if model == "bounded":
        m, d, n = lqg_amd.BoundedActor(T=500, device=dev, dtype=dt), 2, 300
    elif model == "bounded_9_candidates":
        c = torch.linspace(10.0, 30.0, 9, device=dev, dtype=dt)
        m, d, n = lqg_amd.BoundedActor(T=500, sigma_target=c, device=dev, dtype=dt), 2, 50
    elif model == "subjective2d":
        m, d, n = lqg_amd.SubjectiveActor(dim=2, T=500, device=dev, dtype=dt), 4, 300
    elif model == "pointmass":
        m, d, n = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=dt), 2, 4096
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench_configs import hand2d_system
        m, d, n = hand2d_system(400, dev, dt), 4, 2048
    with torch.no_grad():
        x = m.simulate(3, n=n)
        x = (x[0] if x.dim() == 4 else x)[..., :d].contiguous()
    plan = LogLikelihoodPlan(m, x)
    assert all(wk["scan"] for wk in plan.work), plan.description
    _replay(plan)


def test_moment_scans_replay_bitwise(monkeypatch):
    """lqg_conditional_moments_scan: mu, Sigma of every step from the prefix scans (materialising finalisers)."""
    import lqg_amd
    from lqg_amd import _hip
    monkeypatch.setenv("LQG_SCAN", "1")
    m = lqg_amd.PointMassBoundedActor(T=300, action_variability=0.5, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        x = m.simulate(4, n=8)[..., :2].contiguous()
    mu0, S0 = _hip.conditional_moments(m.actor, m.dynamics, x, system=m)
    for _ in range(50):
        mu, S = _hip.conditional_moments(m.actor, m.dynamics, x, system=m)
        assert torch.equal(mu.view(torch.int64), mu0.view(torch.int64)) and torch.equal(S.view(torch.int64), S0.view(torch.int64))
