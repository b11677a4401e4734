"""-m gpu: size-independent properties at BASELINE.json's full sizes (where the CPU oracle cannot follow) plus the
edge cases of the boundary (empty batches, T = 1, one trial, shared vs per-system trials)."""
import numpy as np
import pytest
import torch

import lqg_amd
from lqg_amd import _hip, workload
from gpu_common import np_

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def headline():
    B, T = 1 << 18, 500                                   # BASELINE headline / config 5 shape, full batch
    system, _ = workload.headline_system(B, T, seed=1234, device=DEV, dtype=torch.float32)
    x = workload.pack_trials(workload.simulate_one_trial_each(system, seed=99))
    return system, x


def test_full_batch_all_paths_agree(headline, monkeypatch):
    """2^18 solves: every fp32 path (decoupled + specialised, specialised joint, generic dense joint) against the fp64
    path (itself within 1e-14 of the oracle): EVERY system within the north-star 1e-6 (measured: median 3e-8,
    p99.9 1.9e-7, max 5.7e-7 with the deviation-form innovation; 1.6e-6 without it); the three fp64 paths agree to
    1e-11."""
    system, x = headline
    s64, x64 = system.to(torch.float64), x.double()

    def check32(ll, ref):
        r = (ll.double() / ref - 1).abs().flatten()
        assert float(torch.quantile(r[: 1 << 18], 0.999)) < 5e-7 and float(r.max()) < 1e-6

    ref = s64.log_likelihood(x64).clone()                 # decoupled + specialised, fp64
    check32(system.log_likelihood(x), ref)
    monkeypatch.setenv("LQG_NO_DECOUPLE", "1")
    ref_joint = s64.log_likelihood(x64).clone()
    check32(system.log_likelihood(x), ref)
    monkeypatch.setenv("LQG_NO_SPECIALIZE", "1")
    ref_gen = s64.log_likelihood(x64).clone()
    check32(system.log_likelihood(x), ref)
    assert ref.shape == (1 << 18, 1) and torch.isfinite(ref).all()
    assert float((ref_joint / ref - 1).abs().max()) < 1e-11 and float((ref_gen / ref - 1).abs().max()) < 1e-11
    # checksum of checksums: the fp64 objectives of the extreme paths agree to 1e-12 rel
    assert abs(float(_hip.sum_trials(ref.view(1, -1)) / _hip.sum_trials(ref_gen.view(1, -1))) - 1) < 1e-12


def test_full_batch_permutation_and_sharding_invariance(headline):
    """Systems are independent: evaluating a shard gives exactly the shard of the full result; the objective is the
    sum of shard objectives (what the multi-GPU path relies on)."""
    system, x = headline
    B = system.n_systems
    ll = system.log_likelihood(x).clone()
    lo, hi = B // 4, B // 2
    sub = workload.slice_system(system, lo, hi)
    ll_sub = sub.log_likelihood(x[lo:hi])
    assert torch.equal(ll_sub, ll[lo:hi])                 # bitwise: a lane's arithmetic does not depend on its neighbours
    total = _hip.sum_trials(ll.view(1, -1))
    parts = sum(_hip.sum_trials(ll[a:b].reshape(1, -1)) for a, b in ((0, lo), (lo, hi), (hi, B)))
    assert abs(float(parts / total) - 1) < 1e-12


def test_config3_shape_objective_is_sum_of_trials(monkeypatch):
    """4096 candidates x 1024 shared trials (config 3 shape, shortened horizon): [B, n] result, fp64 objective equals
    the sum over trials, shared x == explicitly replicated x."""
    Bc, n, T = 4096, 1024, 120
    m, _ = workload.bounded_system(Bc, T, seed=5, device=DEV, dtype=torch.float32)
    truth = lqg_amd.BoundedActor(T=T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, device=DEV)
    x = truth.simulate(13, n=n)
    ll = m.log_likelihood(x)
    assert ll.shape == (Bc, n) and torch.isfinite(ll).all()
    obj = _hip.sum_trials(ll)
    assert obj.dtype == torch.float64 and obj.shape == (Bc,)
    assert float(((obj - ll.double().sum(-1)).abs() / obj.abs()).max()) < 1e-10
    xf = x[:3].unsqueeze(0).expand(Bc, 3, T + 1, 2).contiguous()                       # per-system copies of the trials
    few = m.log_likelihood(xf).clone()            # (3 trials per system: the per-trial sweep runs time-chunked)
    assert float((few.double() / ll[:, :3].double() - 1).abs().max()) < 2e-6
    monkeypatch.setenv("LQG_TRIAL_CHUNKS", "0")   # same sweep geometry: bitwise
    assert torch.equal(m.log_likelihood(xf), ll[:, :3])


def test_metamorphic_bounded_equals_subjective_at_scale():
    """Likelihood analogue of the reference's tests/lqg_test.py:69-93 over 4096 candidates."""
    B, T = 4096, 200
    p = workload.sample_params(("action_variability", "sigma_target", "sigma_cursor", "action_cost"), B, 3, DEV, torch.float64)
    b = lqg_amd.BoundedActor(T=T, device=DEV, dtype=torch.float64, **p)
    s = lqg_amd.SubjectiveActor(T=T, subj_noise=1.0, subj_vel_noise=0.0, device=DEV, dtype=torch.float64, **p)
    x = workload.simulate_one_trial_each(b, seed=1)
    assert float((b.log_likelihood(x) / s.log_likelihood(x) - 1).abs().max()) < 1e-9


def test_edge_cases_of_the_boundary():
    m = lqg_amd.BoundedActor(T=1, device=DEV, dtype=torch.float64)                 # single step
    x = m.simulate(0, n=5)
    assert x.shape == (5, 2, 2) and torch.isfinite(m.log_likelihood(x)).all()
    mu, Sig = m.conditional_moments(x[0])
    assert mu.shape == (1, 4) and Sig.shape == (1, 4, 4)
    m = lqg_amd.BoundedActor(T=50, device=DEV)
    assert m.log_likelihood(torch.zeros(0, 51, 2, device=DEV)).shape == (0,)        # no trials: nothing launched
    with pytest.raises(lqg_amd._abi.LqgHipError, match="T\\+1"):
        m.log_likelihood(torch.zeros(3, 50, 2, device=DEV))                        # wrong number of rows



@pytest.mark.parametrize("big_batch", [False, True], ids=["cooperative", "compiled"])
def test_unlisted_model_shape(oracle_lib, monkeypatch, big_batch):
    """A hand-built model whose shape (x=b=3, u=2, y=2) is not in lqg_dims.def.  A small batch runs on the run-time-dims
    cooperative kernels of the main library — nothing is compiled; from JIT_MIN_SYSTEMS systems per call lqg_amd compiles
    an auxiliary lane-kernel library for the shape on first use (lqg_amd.build.build_dims_library; forced here by
    lowering the threshold).  Every entry point works and matches the oracle either way; what has neither kernels nor
    a compiler fails loudly instead of falling back."""
    from lqg_amd import _abi, build
    from lqg_amd.belief import kf
    from lqg_amd.control import lqr
    if big_batch:
        if not __import__("os").path.exists(build.HIPCC):
            pytest.skip("no hipcc on this box")
        monkeypatch.setattr(_abi, "JIT_MIN_SYSTEMS", 1)
    before = set(_abi._dims_libs)
    rng = np.random.default_rng(11)
    A = np.eye(3) + 0.05 * rng.standard_normal((3, 3))
    B = 0.1 * rng.standard_normal((3, 2))
    F = rng.standard_normal((2, 3))
    V = np.diag([1.0, 0.5, 0.7]) + 0.05 * rng.standard_normal((3, 3))
    W = np.diag([2.0, 3.0])
    Q = np.diag([1.0, 0.5, 0.2])
    R = 0.3 * np.eye(2)
    T = 40
    import lqg_np as O
    spec = O.time_stack_spec(A, B, F, V, W, Q, R, T)
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64, device=DEV)
    m = lqg_amd.LQG(t64(A), t64(B), t64(F), t64(V), t64(W), t64(Q), t64(R), T=T)
    assert not _abi.load().lqg_dims_supported(_abi.F64, _abi.C.byref(_abi.Dims(3, 3, 2, 2, 3, 3, 2, 3, 2)))
    x = m.simulate(2, n=5)
    assert x.shape == (5, T + 1, 3)
    xn = x.cpu().numpy()
    L, _, H = oracle_lib.riccati_backward(spec)
    K = oracle_lib.kalman_forward(spec)
    g = lqr.backward(m.actor)
    assert np.abs(np_(g.L) - L).max() < 1e-11 and np.abs(np_(kf.forward(m.actor, None)) - K).max() < 1e-11
    for d in (3, 2):                                                      # full and partial observation
        ref = oracle_lib.log_likelihood(spec, spec, xn[..., :d])
        assert np.abs(np_(m.log_likelihood(x[..., :d])) / ref - 1).max() < 1e-10          # several trials
        assert abs(float(m.log_likelihood(x[:1, :, :d])[0]) / ref[0] - 1) < 1e-10          # one trial (specialised)
    mu, Sig = m.conditional_moments(x[0])
    mu_r, Sig_r = oracle_lib.conditional_moments(spec, spec, xn[:1])
    assert np.abs(np_(mu) - mu_r[0]).max() < 1e-10 and np.abs(np_(Sig) - Sig_r).max() < 1e-10
    assert (set(_abi._dims_libs) != before) == big_batch                  # compiled only in the big-batch mode
    # without a compiler: the likelihood AND (round 4: cooperative reverse-mode sweep) the gradient of an unknown shape are
    # served by the main library; a shape beyond every kernel family fails loudly, never a fallback
    monkeypatch.setattr(build, "HIPCC", "/nonexistent/hipcc")
    assert _abi.library_for(dict(x=3, b=4, u=1, y=2, d=3), family=_abi.FAM_ADJOINT) is _abi.load()
    assert _abi.library_for(dict(x=3, b=4, u=1, y=2, d=3)) is _abi.load()
    with pytest.raises(_abi.LqgHipError):
        _abi.library_for(dict(x=3, b=4, u=5, y=2, d=3), family=_abi.FAM_ADJOINT)
    sig = torch.tensor(2.0, dtype=torch.float64, device=DEV, requires_grad=True)          # ... and the gradient is right
    Wg = torch.diag(torch.stack([sig, torch.tensor(3.0, dtype=torch.float64, device=DEV)]))
    mg = lqg_amd.LQG(t64(A), t64(B), t64(F), t64(V), Wg, t64(Q), t64(R), T=T)
    mg.log_likelihood(x).sum().backward()
    h = 1e-6
    with torch.no_grad():
        f = lambda v: float(lqg_amd.LQG(t64(A), t64(B), t64(F), t64(V), t64(np.diag([v, 3.0])), t64(Q), t64(R), T=T)
                            .log_likelihood(x).sum())
        fd = (f(2.0 + h) - f(2.0 - h)) / (2 * h)
    assert abs(float(sig.grad) - fd) < 1e-6 * max(1.0, abs(fd))


def _run_bench(*flags):
    """bench.py as a CHILD process (never re-exec a process that touched the GPU); returns rank 0's JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu_headline_weak_scaling():
    """The N > 1 bench flow WITH the HIP kernels on the one GPU there is: `bench.py --gpus 2 --share-gpu` starts two
    ranks (torch.distributed.run, child processes), both compute on device 0, the objective is all-reduced (gloo, host
    tensor).  Rank r's data is seeded 1234 + r: rank 0's partial objective equals the single-rank run's, and the
    all-reduced objective is the sum of the two partials."""
    common = ["--log2-batch", "14", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
    one = _run_bench(*common)
    two = _run_bench("--gpus", "2", "--share-gpu", *common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["world_size"] == 2 and two["share_gpu"]
    assert len(two["per_rank_solves_per_s"]) == 2 and len(two["per_rank_objective"]) == 2
    assert two["scaling"] == "weak" and two["all_finite"]
    assert abs(two["per_rank_objective"][0] / one["objective_sum"] - 1) < 1e-12
    assert abs(sum(two["per_rank_objective"]) / two["objective_sum"] - 1) < 1e-12
    assert two["per_rank_objective"][0] != two["per_rank_objective"][1]          # different shards
    assert two["value"] > 0 and two["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-6


def test_two_ranks_share_the_gpu_config3_trial_split():
    """`--config 3` (4096 candidates x 1024 trials, T = 1067) with the TRIAL axis split over two ranks: the all-reduced
    [4096] objective equals the single-rank one to fp64 rounding (strong scaling: same work, same answer)."""
    common = ["--config", "3", "--steps", "2", "--warmup", "1"]
    one = _run_bench(*common)
    two = _run_bench("--gpus", "2", "--share-gpu", *common)
    assert two["n_gpus"] == 2 and two["world_size"] == 2 and two["share_gpu"] and two["scaling"] == "strong"
    assert len(two["per_rank_s"]) == 2 and two["config"]["trials_per_rank"] == 512
    assert two["best_candidate"] == one["best_candidate"]
    assert abs(two["objective_checksum"] / one["objective_checksum"] - 1) < 1e-12
    assert abs(two["objective_max"] / one["objective_max"] - 1) < 1e-12


def test_eight_ranks_share_the_gpu_headline():
    """The 8-rank flow the driver's scaling run will take (VERDICT r05 task 5), on the one GPU there is: eight ranks, eight
    different shards (seed 1234 + rank), eight distinct partial objectives whose sum is the all-reduced objective; rank 0's
    partial equals the single-rank run's."""
    common = ["--log2-batch", "14", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
    one = _run_bench(*common)
    eight = _run_bench("--gpus", "8", "--share-gpu", *common)
    assert eight["n_gpus"] == 8 and eight["world_size"] == 8 and eight["collective"]["world_size"] == 8 and eight["share_gpu"]
    pr = eight["per_rank_objective"]
    assert len(pr) == 8 and len(set(pr)) == 8 and len(eight["per_rank_solves_per_s"]) == 8
    assert abs(pr[0] / one["objective_sum"] - 1) < 1e-12
    assert abs(sum(pr) / eight["objective_sum"] - 1) < 1e-12
    assert eight["scaling"] == "weak" and eight["all_finite"] and eight["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-6
    assert abs(eight["value"] * eight["ms_per_step"] * 1e-3 / (8 * 2 ** 14) - 1) < 1e-9     # whole-job aggregate over all ranks


def test_eight_ranks_share_the_gpu_config3_128_trials_per_rank():
    """`--config 3` over 8 ranks: every rank holds all 4096 candidates and 128 of the 1024 trials; the all-reduced [4096]
    objective equals the single-rank one to fp64 rounding; eight distinct partial objectives."""
    common = ["--config", "3", "--steps", "2", "--warmup", "1"]
    one = _run_bench(*common)
    eight = _run_bench("--gpus", "8", "--share-gpu", *common)
    assert eight["n_gpus"] == 8 and eight["world_size"] == 8 and eight["scaling"] == "strong"
    assert eight["config"]["trials_per_rank"] == 128 and len(eight["per_rank_s"]) == 8
    pr = eight["per_rank_objective"]
    assert len(pr) == 8 and len(set(pr)) == 8
    assert abs(sum(pr) / eight["objective_checksum"] - 1) < 1e-12
    assert eight["best_candidate"] == one["best_candidate"]
    assert abs(eight["objective_checksum"] / one["objective_checksum"] - 1) < 1e-12
    assert abs(eight["objective_max"] / one["objective_max"] - 1) < 1e-12


def test_eight_ranks_share_the_gpu_config4_32768_trials_per_rank():
    """`--config 4` literally: 262 144 trials of the 2-D hand model, 32 768 per rank over 8 ranks.  The data set is defined in
    8 seeded blocks, so the 1-rank run (all 8 blocks) and the 8-rank run (one block each) score the SAME trials.  In fp64 the
    objective agrees to 1e-12.  In fp32 a shard of 32 768 trials takes the time-chunked per-trial sweep (more waves in flight)
    where 262 144 trials run one pass: the same trials in a different fp32 evaluation order, 1e-7 per log-likelihood (both
    within 1e-6 of the oracle) and 7e-11 on the sum of 262 144 of them — asserted at 1e-9."""
    for dtype, tol in (("f32", 1e-9), ("f64", 1e-12)):
        common = ["--config", "4", "--steps", "2", "--warmup", "1", "--dtype", dtype]
        one = _run_bench(*common)
        eight = _run_bench("--gpus", "8", "--share-gpu", *common)
        assert one["config"]["trials_per_rank"] == 262144 and eight["config"]["trials_per_rank"] == 32768
        assert eight["n_gpus"] == 8 and eight["world_size"] == 8 and eight["scaling"] == "strong"
        pr = eight["per_rank_objective"]
        assert len(pr) == 8 and len(set(pr)) == 8
        assert abs(sum(pr) / eight["objective_sum"] - 1) < 1e-12
        assert abs(eight["objective_sum"] / one["objective_sum"] - 1) < tol, dtype
        assert eight["max_rel_err_vs_fp64_oracle"] < 1e-6 and one["max_rel_err_vs_fp64_oracle"] < 1e-6


def test_driver_launch_line_on_rccl_with_one_rank():
    """The launch line the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` — at N = 1 on the one GPU there is: the process group comes up on the `nccl`
    backend (= RCCL), the objective goes through a real all-reduce, and the line carries the collective fields the first
    multi-GPU run will be read by (backend, ranks seen, all-reduce latency percentiles)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--log2-batch", "14", "--steps", "3",
                        "--warmup", "1", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["all_finite"] and out["value"] > 0
    assert out["collective"]["backend"].startswith("nccl") and out["collective"]["rccl_ranks_seen"] == 1
    assert out["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-6
    one = _run_bench("--log2-batch", "14", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline")
    assert abs(out["objective_sum"] / one["objective_sum"] - 1) < 1e-12


def test_more_than_65535_systems_with_several_trials_each():
    """the per-trial sweeps put the system index in grid.y.  More than 65 535 systems with three or more trials each
    (forward: operator stream + k_trial_sp; reverse: k_asp_trial_rev) must launch and agree with a small batch of the same candidates —
    verified on gfx950 / ROCm 7.2 at 70 000 and 140 000 systems (round 6: scripts/r06_gridy_probe.py)."""
    dev = torch.device("cuda")
    for B, n, dt, tol in ((70000, 3, torch.float32, 1e-5), (140000, 5, torch.float64, 1e-12)):
        sig = torch.linspace(3.0, 40.0, B, device=dev, dtype=dt)
        x = lqg_amd.BoundedActor(T=40, device=dev, dtype=dt).simulate(5, n=n).contiguous()
        ll = lqg_amd.BoundedActor(T=40, sigma_target=sig, device=dev, dtype=dt).log_likelihood(x)
        idx = torch.tensor([0, 65535, 65536, B - 1], device=dev)
        ref = lqg_amd.BoundedActor(T=40, sigma_target=sig[idx], device=dev, dtype=dt).log_likelihood(x)
        assert ll.shape == (B, n) and bool(torch.isfinite(ll).all())
        assert float(((ll[idx] - ref) / ref).abs().max()) < tol
        s2 = sig.clone().requires_grad_(True)
        lqg_amd.BoundedActor(T=40, sigma_target=s2, device=dev, dtype=dt).log_likelihood(x).sum().backward()
        s3 = sig[idx].clone().requires_grad_(True)
        lqg_amd.BoundedActor(T=40, sigma_target=s3, device=dev, dtype=dt).log_likelihood(x).sum().backward()
        assert bool(torch.isfinite(s2.grad).all())
        assert float(((s2.grad[idx] - s3.grad) / s3.grad).abs().max()) < (1e-3 if dt == torch.float32 else 1e-9)
