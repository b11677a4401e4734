#!/usr/bin/env python3
"""bench_configs.py — the five BASELINE.json configs on ONE MI355X (evidence beside bench.py's headline line).

Prints one JSON line per config: solves/s, trial-evals/s, per-phase kernel times (hipEvents recorded by the library),
parity against the fp64 CPU oracle on a sample, and for config 5 the fp32-vs-fp64 tolerance sweep.
Units: solve = one system + one trajectory end to end; trial-eval = one more trajectory for an
already-solved system.   usage: python bench_configs.py [--configs 1,2,3,4,5] [--reps 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import lqg_amd
from lqg_amd import _hip, workload
from lqg_amd.system import Actor, System
from bench import PEAK_FP32_TFLOPS, PEAK_FP64_TFLOPS, PEAK_HBM_GBS, algorithmic_bytes_per_solve, algorithmic_flops_per_step


def hand2d_system(T, device, dtype, cursor_noise=0.1):
    """2-D hand model of notebooks/HandModel.ipynb (BASELINE config 4), observed dims first, position noise added
    so that the observed block of V V^T is non-singular — same matrices as
    oracle/gen_golden.py::hand2d."""
    dt, m, tau = 1.0 / 60.0, 1.0, 0.04
    A1 = torch.zeros(5, 5, dtype=torch.float64)
    A1[0, 0] = 1.0
    A1[1:, 1:] = torch.tensor([[1.0, dt, 0.0, 0.0], [0.0, 1.0, dt / m, 0.0], [0.0, 0.0, 1.0 - dt / tau, dt / tau],
                               [0.0, 0.0, 0.0, 1.0 - dt / tau]], dtype=torch.float64)
    B1 = dt / tau * torch.tensor([[0.0], [0.0], [0.0], [0.0], [1.0]], dtype=torch.float64)
    F1 = torch.eye(2, 5, dtype=torch.float64)
    V1 = torch.diag(torch.tensor([1.0, cursor_noise, 0.0, 0.0, 0.5], dtype=torch.float64))
    W1 = torch.diag(torch.tensor([6.0, 6.0], dtype=torch.float64))
    Q1 = torch.zeros(5, 5, dtype=torch.float64)
    Q1[:2, :2] = torch.tensor([[1.0, -1.0], [-1.0, 1.0]])
    bdg = torch.block_diag
    A, B, F, V, W, Q = bdg(A1, A1), bdg(B1, B1), bdg(F1, F1), bdg(V1, V1), bdg(W1, W1), bdg(Q1, Q1)
    R = torch.eye(2, dtype=torch.float64)
    perm = [0, 1, 5, 6, 2, 3, 4, 7, 8, 9]
    A, B, V, F, Q = A[perm][:, perm], B[perm], V[perm], F[:, perm], Q[perm][:, perm]
    cv = lambda t: t.to(device=device, dtype=dtype).contiguous()
    spec = Actor(A=cv(A), B=cv(B), F=cv(F), V=cv(V), W=cv(W), Q=cv(Q), R=cv(R), T=T)
    return System(actor=spec, dynamics=spec)


def timed_loglik(system, x, reps):
    """Run the log-likelihood plan `reps` times; return ll, median per-phase milliseconds and the path taken."""
    from lqg_amd.plan import LogLikelihoodPlan
    plan = LogLikelihoodPlan(system, x, events=True)
    ph, wall = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(reps + 1):
        e0.record()
        ll = plan.run()
        e1.record()
        p3 = plan.phase_ms()
        e1.synchronize()
        if r:
            ph.append(list(p3) + [sum(p3)])
            wall.append(e0.elapsed_time(e1))
    ph = np.median(np.array(ph), axis=0)
    # total_ms = sum of the kernel phases over the components (attribution); wall_ms = the evaluation as the caller sees
    # it (independent components of a small-batch problem run concurrently on side streams)
    return ll.clone(), dict(riccati_ms=float(ph[0]), forward_ms=float(ph[1]), trial_ms=float(ph[2]), total_ms=float(ph[3]),
                            wall_ms=float(np.median(wall)), concurrent_components=bool(plan.side),
                            workspace_MB=sum(wk["nbytes"] for wk in plan.work) / 1e6, path=plan.description)


def host_spec(spec, sel=None):
    import oracle  # noqa
    out = {}
    for f in lqg_amd.LQGSpec._fields:
        t = getattr(spec, f)
        if sel is not None and t.dim() == workload._batched_ndim(f):
            t = t[torch.as_tensor(sel, device=t.device)]
        elif sel is not None:
            t = t.expand(len(sel), *t.shape)
        out[f] = t.double().cpu().numpy()
    return out


def oracle_check(system, x, ll, n_samples=16):
    """max rel err of ll against the fp64 oracle on a sample of (system, trial) pairs."""
    import oracle as OC
    batched = system.n_systems is not None
    n = x.shape[-3]
    rng = np.random.default_rng(0)
    if batched:
        B = system.n_systems
        sel = np.unique(rng.integers(0, B, size=min(n_samples, B)))
        tr = np.unique(rng.integers(0, n, size=min(4, n)))
        a, d = host_spec(system.actor, sel), host_spec(system.dynamics, sel)
        xs = x[..., tr, :, :]
        xs = (xs[torch.as_tensor(sel, device=x.device)] if x.dim() == 4 else xs.expand(len(sel), *xs.shape)).double().cpu().numpy()
        ref = OC.log_likelihood(a, d, xs)
        got = ll[torch.as_tensor(sel, device=ll.device)][:, torch.as_tensor(tr, device=ll.device)].double().cpu().numpy()
    else:
        tr = np.unique(rng.integers(0, n, size=min(n_samples, n)))
        a, d = host_spec(system.actor), host_spec(system.dynamics)
        ref = OC.log_likelihood(a, d, x[torch.as_tensor(tr, device=x.device)].double().cpu().numpy())
        got = ll[torch.as_tensor(tr, device=ll.device)].double().cpu().numpy()
    return float(np.abs(got / ref - 1).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,4,5")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink batch sizes (debug)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    want = {int(c) for c in args.configs.split(",")}
    sc = lambda v: max(1, int(v * args.scale))

    def emit(cfg, name, system, x, dtype_name, extra=None):
        ll, ph = timed_loglik(system, x, args.reps)
        B = system.n_systems or 1
        n = x.shape[-3]
        dm = dict(x=system.xdim, b=system.bdim, u=system.udim, y=system.ydim, d=x.shape[-1])
        fl = algorithmic_flops_per_step(**dm)                       # SURVEY.md §8(d), reference formulation
        T_ = system.T
        alg_flops = B * T_ * (fl["total"] - fl["mean"] - fl["logprob"]) + B * n * T_ * (fl["mean"] + fl["logprob"])
        w_ = 4 if dtype_name == "f32" else 8
        alg_bytes = B * (algorithmic_bytes_per_solve(T=T_, w=w_, **dm) - w_ * (T_ + 1) * dm["d"]) + \
            (1 if x.dim() == 3 else B) * n * w_ * (T_ + 1) * dm["d"] + B * n * w_
        peak = PEAK_FP32_TFLOPS if dtype_name == "f32" else PEAK_FP64_TFLOPS
        out = dict(config=cfg, workload=name, dtype=dtype_name, systems=B, trials_per_system=n, T=system.T,
                   dims=dm, **ph,
                   algorithmic=dict(tflops=alg_flops / (ph["total_ms"] * 1e-3) / 1e12,
                                    frac_vector_peak=alg_flops / (ph["total_ms"] * 1e-3) / 1e12 / peak,
                                    hbm_gbs=alg_bytes / (ph["total_ms"] * 1e-3) / 1e9,
                                    frac_hbm_peak=alg_bytes / (ph["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    note="M1 accounting of SURVEY.md 8(d): reference-formulation flops, inputs-once bytes"),
                   solves_per_s=B / (ph["total_ms"] * 1e-3),
                   trial_evals_per_s=B * n / (ph["total_ms"] * 1e-3),
                   all_finite=bool(torch.isfinite(ll).all()),
                   max_rel_err_vs_fp64_oracle=oracle_check(system, x, ll))
        if extra:
            out.update(extra)
        print(json.dumps(out), flush=True)
        return ll

    if 1 in want:   # Tutorial LQG, state dim 2, T=100, 32 trials (plumbing)
        for dt_, nm in ((torch.float32, "f32"), (torch.float64, "f64")):
            m = lqg_amd.BoundedActor(T=100, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5,
                                     device=dev, dtype=dt_)
            x = m.simulate(11, n=32)
            emit(1, "Tutorial LQG (BoundedActor n=2), T=100, 32 trials, one system", m, x, nm)
    if 2 in want:   # PointMass n=4, T=500, 65536 trials
        m = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=torch.float32)
        x = workload.pack_trials(m.simulate(12, n=sc(65536))[..., :2].contiguous())
        emit(2, "PointMassBoundedActor n=4, T=500, 65536 trials of (target, cursor), one system", m, x, "f32")
    if 3 in want:   # data.mat-shaped: 1068 rows, 4096 candidates x 1024 trials
        Bc, Nt, T = sc(4096), sc(1024), 1067
        m, _ = workload.bounded_system(Bc, T, seed=5, device=dev, dtype=torch.float32)
        truth = lqg_amd.BoundedActor(T=T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, action_variability=0.5,
                                     device=dev, dtype=torch.float32)
        x = workload.pack_trials(truth.simulate(13, n=Nt))
        ll = emit(3, "BoundedActor n=2, data.mat-shaped trials (1068 rows), 4096 candidates x 1024 shared trials", m, x, "f32")
        obj = _hip.sum_trials(ll)
        print(json.dumps(dict(config=3, note="objective = sum over trials per candidate (lqg_sum_trials, fp64)",
                              best_candidate=int(obj.argmax()), objective_max=float(obj.max()))), flush=True)
    if 4 in want:   # hand model 2-D, n=10, T=1000, 262144 trials over 8 GPUs -> 32768 per GPU
        m = hand2d_system(1000, dev, torch.float32)
        x = workload.pack_trials(m.simulate(14, n=sc(32768))[..., :4].contiguous())
        emit(4, "2-D hand model n=10 (m=20), T=1000, 32768 trials per GPU (262144 over 8), one system", m, x, "f32")
    if 5 in want:   # n=6, T=500, 1048576 trials, fp32 vs fp64
        n5 = sc(1 << 20)
        m64 = lqg_amd.SubjectiveActor(dim=2, T=500, device=dev, dtype=torch.float64)
        x64 = m64.simulate(15, n=n5)
        ll64 = emit(5, "SubjectiveActor(dim=2) n=6, T=500, 1048576 trials, one system (additive noise)", m64,
                    workload.pack_trials(x64), "f64")
        m32 = m64.to(torch.float32)
        x32 = workload.pack_trials(x64.float())
        ll32, _ = timed_loglik(m32, x32, 1)
        rel = (ll32.double() / ll64 - 1).abs()
        emit(5, "SubjectiveActor(dim=2) n=6, T=500, 1048576 trials, one system (additive noise)", m32, x32, "f32",
             extra=dict(fp32_vs_fp64=dict(max_rel=float(rel.max()), p99_rel=float(rel.quantile(0.99)),
                                          mean_rel=float(rel.mean()))))


if __name__ == "__main__":
    main()
