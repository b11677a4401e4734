"""CPU: host-side mirror of the reference's interface — shapes, views, strides, model constructors."""
import numpy as np
import pytest
import torch

import lqg_amd
from lqg_amd import _abi, _hip, workload
from lqg_amd.utils import time_stack_spec
from conftest import load_golden


def test_time_stack_spec_matches_reference_shapes_without_copies():
    """lqg/utils.py:10-35: same shapes/values, but the T copies are a stride-0 view."""
    A, B = torch.eye(3), torch.ones(3, 1)
    F, V, W = torch.ones(2, 3), torch.eye(3), torch.eye(2)
    Q, R = torch.eye(3), torch.eye(1)
    s = time_stack_spec(A, B, F, V, W, Q, R, T=7)
    assert s.A.shape == (7, 3, 3) and s.B.shape == (7, 3, 1) and s.F.shape == (7, 2, 3)
    assert s.q.shape == (7, 3) and s.P.shape == (7, 1, 3) and s.r.shape == (7, 1)
    assert s.Qf.shape == (3, 3) and s.qf.shape == (3,)
    assert s.A.stride(0) == 0 and float(s.q.abs().sum()) == 0.0
    assert tuple(lqg_amd.LQGSpec._fields) == ("Q", "q", "Qf", "qf", "P", "R", "r", "A", "B", "V", "F", "W")


@pytest.mark.parametrize("name,ctor,kw,d", [
    ("bounded_T100", lqg_amd.BoundedActor, dict(T=100, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05,
                                                action_variability=0.5), 2),
    ("optimal_T30", lqg_amd.OptimalActor, dict(T=30), 2),
    ("relobs_T40", lqg_amd.RelativeObservationBoundedActor, dict(T=40), 2),
    ("subjective1d_T50", lqg_amd.SubjectiveActor, dict(dim=1, T=50), 2),
    ("subjective2d_T60", lqg_amd.SubjectiveActor, dict(dim=2, T=60, action_cost=0.5, sigma_cursor=3.0,
                                                       subj_noise=1.3, subj_vel_noise=0.7), 4),
    ("bounded2d_T40", lqg_amd.BoundedActor, dict(dim=2, T=40, action_cost=0.2), 4),
    ("pointmass_d2_T50", lqg_amd.PointMassBoundedActor, dict(T=50, action_variability=0.5), 2),
])
def test_model_zoo_builds_the_reference_matrices(name, ctor, kw, d):
    """Constructors reproduce the matrices the reference built (stored in the golden files)."""
    g, actor, dyn = load_golden(name)
    m = ctor(device="cpu", dtype=torch.float64, **kw)
    for f in lqg_amd.LQGSpec._fields:
        a, dd = getattr(m.actor, f).numpy(), getattr(m.dynamics, f).numpy()
        assert a.shape == actor[f].shape, f
        assert np.allclose(a, actor[f], atol=1e-9, rtol=1e-9), f
        assert np.allclose(dd, dyn[f], atol=1e-9, rtol=1e-9), f
    assert (m.xdim, m.bdim, m.udim, m.ydim, m.T) == (dyn["A"].shape[1], actor["A"].shape[1], dyn["B"].shape[2],
                                                     dyn["F"].shape[1], actor["A"].shape[0])


def test_candidate_axis_builds_batched_specs():
    sig = torch.linspace(5.0, 50.0, 7)
    m = lqg_amd.BoundedActor(sigma_target=sig, T=30, device="cpu")
    assert m.n_systems == 7 and m.actor.W.shape == (7, 30, 2, 2) and m.actor.W.stride(1) == 0
    assert torch.allclose(m.actor.W[:, 0, 0, 0], sig)
    one = lqg_amd.BoundedActor(sigma_target=float(sig[3]), T=30, device="cpu")
    assert torch.allclose(m.actor.W[3], one.actor.W)


def test_launch_views_and_dims():
    m = lqg_amd.SubjectiveActor(dim=2, T=50, sigma_target=torch.tensor([3.0, 4.0, 5.0]), device="cpu")
    ln = _hip.Launch(m.actor, m.dynamics, d=4, n_trials=9)
    p = ln.p
    assert (p.n_sys, p.n_trials, p.T, p.dtype) == (3, 9, 50, _abi.F32)
    dm = p.dims
    assert (dm.x, dm.b, dm.u, dm.y, dm.d, dm.nva, dm.nwa, dm.nvd, dm.nwd) == (4, 6, 2, 4, 4, 6, 4, 4, 4)
    assert p.actor.A.st == 0 and p.actor.A.sb == 36 and p.actor.A.sr == 6 and p.actor.A.sc == 1
    assert p.actor.q.ptr is None and p.actor.P.ptr is None          # known-zero affine terms -> NULL
    assert p.dynamics.Q.ptr is None                                  # dynamics cost is never read
    assert p.Sigma0.ptr is None
    x = torch.zeros(9, 51, 4)
    xx, xb = _hip._prep_x(ln, x)
    tv = ln.traj(xx, xb)
    assert (tv.sb, tv.sn, tv.st, tv.sd) == (0, 51 * 4, 4, 1)        # shared trials: system stride 0
    xp = workload.pack_trials(x)
    assert xp.shape == x.shape and torch.equal(xp, x)
    tv = ln.traj(xp, False)
    assert (tv.sn, tv.st, tv.sd) == (1, 4 * 9, 9)                   # trial index fastest in memory


def test_wrong_trajectory_length_is_rejected():
    m = lqg_amd.BoundedActor(T=20, device="cpu")
    ln = _hip.Launch(m.actor, m.dynamics, d=2, n_trials=3)
    with pytest.raises(_abi.LqgHipError, match="T\\+1"):
        _hip._prep_x(ln, torch.zeros(3, 20, 2))


def test_workload_generators_are_seeded():
    a, pa = workload.headline_system(16, 10, seed=3, device="cpu", dtype=torch.float64)
    b, pb = workload.headline_system(16, 10, seed=3, device="cpu", dtype=torch.float64)
    c, pc = workload.headline_system(16, 10, seed=4, device="cpu", dtype=torch.float64)
    assert all(torch.equal(pa[k], pb[k]) for k in pa) and not torch.equal(pa["sigma_target"], pc["sigma_target"])
    for k, (lo, hi) in workload.RANGES.items():
        assert float(pa[k].min()) >= lo and float(pa[k].max()) <= hi
    assert (a.xdim, a.bdim, a.udim, a.ydim, a.n_systems) == (4, 6, 2, 4, 16)


def test_decoupling_plan_of_the_model_zoo():
    """dim=2 models split into two 1-D components; coupled models do not."""
    from lqg_amd import decouple
    m = lqg_amd.SubjectiveActor(dim=2, T=5, device="cpu", dtype=torch.float64)
    plan = decouple.plan(m, 4)
    assert [(s.xdim, s.bdim, s.udim, s.ydim, cols, bs) for s, cols, bs in plan] == \
        [(2, 3, 1, 2, [0, 1], [0, 1, 4]), (2, 3, 1, 2, [2, 3], [2, 3, 5])]
    one = lqg_amd.SubjectiveActor(dim=1, T=5, device="cpu", dtype=torch.float64)
    for f in ("A", "B", "F", "V", "W", "Q", "R"):
        assert torch.equal(getattr(plan[0][0].actor, f), getattr(one.actor, f)), f
        if f in ("A", "B", "F", "V", "W"):
            assert torch.equal(getattr(plan[1][0].dynamics, f), getattr(one.dynamics, f)), f
    assert decouple.plan(lqg_amd.BoundedActor(dim=1, T=5, device="cpu"), 2) is None
    assert decouple.plan(lqg_amd.PointMassBoundedActor(T=5, device="cpu"), 2) is None     # coupled through the cost
    # a full Sigma0 couples the beliefs of the two axes: no decoupling
    assert decouple.plan(m, 4, Sigma0=torch.ones(6, 6, dtype=torch.float64)) is None


def test_pattern_extraction_of_the_model_zoo():
    from lqg_amd import specialize
    dims, masks, key = specialize.class_pattern(lqg_amd.SubjectiveActor, 4, dim=2)
    assert dims == dict(x=4, b=6, u=2, y=4, d=4) and len(key) == 16
    assert not masks["DB"].any()                              # Fd Bd - Fa Ba cancels exactly
    assert masks["Aa"].sum() == 8 and masks["Fa"].sum() == 4 and masks["N3"].sum() == 4
    src = specialize.generate_source(key, dims, masks)
    assert "lqg_log_likelihood_sp" in src and "lqg::Mask<6, 6> Aa" in src
    # a parameter that happens to be zero must not narrow the cached class mask
    z = lqg_amd.SubjectiveActor(dim=2, T=3, subj_vel_noise=0.0, device="cpu")
    assert specialize.system_pattern(z, 4)[2] == key


def test_time_parallel_rules():
    """The host-side rules that pick the time-parallel path (lqg_amd/plan.py) and the chunk count of the per-trial sweep
    (csrc/lqg_trial_chunk.hpp through lqg_workspace_bytes): monotone in the joint dimension, overridable by environment."""
    from lqg_amd import plan
    assert plan.scan_max_systems(4) == 8 and plan.scan_max_systems(8) == 32 and plan.scan_max_systems(20) == 64
    assert plan.scan_min_steps(4) == 375 and plan.scan_min_steps(8) == 93 and plan.scan_min_steps(20) == 64
    assert all(plan.scan_max_systems(m) <= plan.scan_max_systems(m + 1) for m in range(2, 24))
    # windows that live in registers (m > 24: the delay-augmented models) are taken for a handful of systems only
    assert plan.scan_max_systems(65) == 10 and plan.scan_max_systems(65, fp64=True) == 24
    assert all(plan.scan_min_steps(m) >= plan.scan_min_steps(m + 1) for m in range(2, 30))


def test_trial_chunk_scratch_grows_with_the_chunk_count(monkeypatch):
    import ctypes as C
    import torch
    import lqg_amd
    from lqg_amd import _abi, _hip
    from lqg_amd import build
    build.build(verbose=False)
    lib = _abi.load()
    m = lqg_amd.BoundedActor(T=400, device="cpu")
    sizes = []
    for chunks in ("0", "2", "8", "1000"):
        monkeypatch.setenv("LQG_TRIAL_CHUNKS", chunks)
        ln = _hip.Launch(m.actor, m.dynamics, d=2, n_trials=64)
        sizes.append(lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD))
    assert sizes[0] < sizes[1] < sizes[2] < sizes[3]
    # forced counts are clamped to chunks of at least 4 steps: 1000 -> 100 chunks of 4
    al = lambda v: (v + 255) // 256 * 256
    assert sizes[3] - sizes[0] == al(99 * 4 * 64 * 4) + al(99 * 4 * 4 * 4) + al(100 * 64 * 8)
    # one or two trials run in-lane (no operator stream, no chunking); a short horizon is never chunked
    monkeypatch.delenv("LQG_TRIAL_CHUNKS")
    short = lqg_amd.BoundedActor(T=12, device="cpu")
    a = lib.lqg_workspace_bytes(C.byref(_hip.Launch(short.actor, short.dynamics, d=2, n_trials=64).p), _abi.OP_LOG_LIKELIHOOD)
    monkeypatch.setenv("LQG_TRIAL_CHUNKS", "0")
    b = lib.lqg_workspace_bytes(C.byref(_hip.Launch(short.actor, short.dynamics, d=2, n_trials=64).p), _abi.OP_LOG_LIKELIHOOD)
    assert a == b


def test_time_varying_pattern_is_the_union_over_time_and_conservative():
    """specialize.pattern_of_time_varying (the masks mode M2's pattern library is compiled for): raw-field masks are the union
    over systems and steps, derived masks are boolean products — supersets of the numerically observed ones — and the pattern
    does not depend on the batch size or on the values of the moving entries."""
    import numpy as np
    import lqg_amd
    from lqg_amd import specialize
    torch.manual_seed(0)

    def moving(B, T, scale):
        base = lqg_amd.SubjectiveActor(dim=2, T=T, sigma_target=torch.linspace(4.0, 9.0, B, dtype=torch.float64),
                                       device="cpu", dtype=torch.float64)
        vary = lambda t: (t.expand(B, T, *t.shape[-2:]) * (1.0 + scale * torch.randn(B, T, *t.shape[-2:], dtype=torch.float64))).contiguous()
        a0, d0 = base.actor, base.dynamics
        actor = a0._replace(A=vary(a0.A), B=vary(a0.B), F=vary(a0.F), V=vary(a0.V), W=vary(a0.W), Q=vary(a0.Q), R=vary(a0.R))
        dyn = d0._replace(A=vary(d0.A), B=vary(d0.B), F=vary(d0.F), V=vary(d0.V), W=vary(d0.W))
        return lqg_amd.System(actor=actor, dynamics=dyn), base

    sys_a, base = moving(3, 5, 1e-3)
    sys_b, _ = moving(7, 4, 5e-2)
    dims_a, masks_a = specialize.pattern_of_time_varying(sys_a, 4)
    dims_b, masks_b = specialize.pattern_of_time_varying(sys_b, 4)
    assert dims_a == dims_b == dict(x=4, b=6, u=2, y=4, d=4)
    assert specialize.pattern_key(dims_a, masks_a) == specialize.pattern_key(dims_b, masks_b)
    # raw fields: exactly the base model's non-zeros (multiplicative motion keeps zeros zero)
    _, base_masks = specialize.pattern_of(base, 4)
    for k in ("Aa", "Ba", "Fa", "Ad", "Bd", "Fd"):
        assert np.array_equal(masks_a[k], base_masks[k]), k
    # derived masks: supersets of the numeric ones of the time-varying system itself (which may cancel by accident) and of the
    # base model's (whose F_d B_d - F_a B_a cancels exactly)
    _, numeric = specialize.pattern_of(sys_a, 4)
    for k in masks_a:
        assert not np.any(numeric[k] & ~masks_a[k]), k
        assert not np.any(base_masks[k] & ~masks_a[k]), k
    assert masks_a["DB"].sum() > base_masks["DB"].sum()            # (the cancellation is not assumed)
    # a structural zero that becomes non-zero at ONE (system, step) enters the union
    sys_a.actor.A[2, 3, 0, 5] = 0.25
    _, masks_c = specialize.pattern_of_time_varying(sys_a, 4)
    assert masks_c["Aa"][0, 5] and not masks_a["Aa"][0, 5]


def test_m2_benchmark_inputs_are_lane_contiguous():
    """Mode M2's time-varying specs must reach the kernels in [T][row][col][system] order (system stride 1): an elementwise
    product with a permuted operand silently produced [T][system][row][col] until the end of round 3."""
    import bench_m2
    B, T = 64, 5
    system, _ = bench_m2.m2_system(torch.device("cpu"), torch.float32, B, T)
    for spec, fields in ((system.actor, "ABFVWQR"), (system.dynamics, "ABFVW")):
        for f in fields:
            t = getattr(spec, f)
            r, c = t.shape[-2:]
            assert t.shape[:2] == (B, T) and t.stride() == (1, r * c * B, c * B, B), (f, t.shape, t.stride())


def test_system_major_time_varying_specs_get_one_layout_hint_and_pack_systems_fixes_it():
    import warnings
    import bench_m2
    from lqg_amd import _hip, workload
    system, _ = bench_m2.m2_system(torch.device("cpu"), torch.float32, 4096, 3)
    _hip._layout_warned = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _hip.Launch(system.actor, system.dynamics, d=4)                     # [T][r][c][B]: nothing to say
        assert not [x for x in w if issubclass(x.category, _hip.LqgLayoutWarning)]
        major = system.actor._replace(A=system.actor.A.contiguous())        # [B][T][r][c]
        _hip.Launch(major, system.dynamics, d=4)
        _hip.Launch(major, system.dynamics, d=4)                            # (once per process)
        hints = [x for x in w if issubclass(x.category, _hip.LqgLayoutWarning)]
        assert len(hints) == 1 and "pack_systems" in str(hints[0].message)
    packed = workload.pack_systems(major.A)
    assert packed.stride(0) == 1 and torch.equal(packed, major.A)
    _hip._layout_warned = False


def test_scan_level_schedule_restated_on_symbolic_elements():
    """scripts/scan_schedule_check.py restates the level schedule of run_scan (csrc/lqg_scan_inst.hip: up-sweep to blocks of B,
    ping-pong scan of the block totals, down-sweep — and plain Brent-Kung) on elements that are step RANGES: every combine joins
    adjacent ranges, no level reads what nobody wrote, every prefix comes out as [0, k]; the rounds of one delay-12 system are the
    ones DESIGN.md §3c quotes.  (The GPU tests check the kernels' numbers in each order at several horizons.)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scan_schedule_check", os.path.join(root, "scripts", "scan_schedule_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for T in list(range(2, 70)) + [129, 301, 500, 1067]:
        for n_sys in (1, 3, 13, 400):
            for order in (0, 2):
                mod.run(T + 1, T, n_sys, order)
                mod.run(T, 0, n_sys, order)
    assert mod.run(501, 500, 1)[1:] == (11, 13) and mod.run(500, 0, 1)[1:] == (10, 10)
    assert mod.run(501, 500, 1, order=2)[1:] == (16, 18) and mod.run(500, 0, 1, order=2)[1:] == (16, 16)


def test_adjoint_patterns_take_structure_from_constructors_and_booleans_not_values():
    """lqg_amd.specialize.adjoint_pattern (round 5): (i) for a System whose fields require grad the masks of the hoisted products
    follow from boolean algebra — Fd Bd - Fa Ba cancels numerically for a shared spec (and for matrices of ones) although
    d ll / d Bd != 0, and A - I has a zero diagonal for A = ones; (ii) zoo classes keep the constructor's structure and report which
    fields ANY parameter moves (for the tracking models only V, W, R); a field that requires grad is always live."""
    import numpy as np
    import torch
    import lqg_amd
    from lqg_amd import specialize
    m = lqg_amd.BoundedActor(T=3, device="cpu", dtype=torch.float64)
    leaf = lambda t: t[:1].clone().requires_grad_(True).expand_as(t)
    a = m.actor._replace(**{f: leaf(getattr(m.actor, f)) for f in ("A", "B", "F", "V", "W", "Q", "R")})
    s = lqg_amd.System(actor=a, dynamics=a)
    dims, masks, key, live = specialize.adjoint_pattern(s, 2)
    assert all(live.values())
    assert masks["DB"].all() and masks["AdmI"].all() and masks["FAa"].all() and masks["N3"].all()
    # the same spec WITHOUT grad: values decide (the products cancel, the diagonal of A - I vanishes)
    _, masks0 = specialize.pattern_of(lqg_amd.System(actor=m.actor, dynamics=m.actor), 2)
    assert not masks0["DB"].any() and not masks0["AdmI"].diagonal().any()
    # zoo class: class-level structure + live fields
    dz, mz, kz, lz = specialize.adjoint_pattern(m, 2)
    assert (dz, kz) == specialize.class_pattern(lqg_amd.BoundedActor, 2, dim=1)[::2]
    assert {k for k, v in lz.items() if v} == {"Va", "Wa", "R", "Vd", "Wd"}
    pm = specialize.zoo_live(lqg_amd.PointMassBoundedActor, (), 2)
    assert pm["Aa"] and pm["Bd"] and not pm["Fa"]
    assert specialize.adjoint_key("abc", specialize.ALL_LIVE) == "abc" and specialize.adjoint_key("abc", lz) != "abc"
    # a constructor argument that requires grad makes its fields live on top of the class-level set
    sig = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    mg = lqg_amd.BoundedActor(T=3, sigma_target=sig, device="cpu", dtype=torch.float64)
    _, _, _, lg = specialize.adjoint_pattern(mg, 2)
    assert lg["Wa"] and lg["Wd"] and not lg["Aa"]
