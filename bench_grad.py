"""Gradient bench (SURVEY.md §8f rank 1 / BASELINE.md's derived "≈54 value+grad evaluations/s" row): value AND
reverse-mode gradient of the summed log-likelihood.

  python bench_grad.py [--steps K] [--warmup W] [--dtype f64|f32]

Two workloads, one JSON line each:
  eval   the reference's inference inner loop: BoundedActor, T=500, 50 trials, 4 parameters, ONE parameter vector —
         `jax.value_and_grad(ll)` per NUTS leapfrog / Adam step (lqg/infer/utils.py:18, lqg/optim.py:142-147).
         Reported for the adjoint sweep and for the batched finite-difference sweep (2P+1 candidates).
  sweep  B independent (system, trial) pairs (x=b=2, T=500): solves+adjoint per second on the split sweep of round 5 (per-kernel
         times from HIP events) and on the round-1 lane kernels.
"""
import argparse
import json
import time

import torch

import lqg_amd
from lqg_amd import grad as G
from lqg_amd.infer import gradient


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--log2-lanes", type=int, default=16)
    ap.add_argument("--sweep-only", action="store_true", help="skip the single-parameter-vector workload (profiling)")
    args = ap.parse_args()
    dt = torch.float64 if args.dtype == "f64" else torch.float32
    dev = torch.device("cuda", 0)

    # ---- eval: one parameter vector, 50 trials of 500 steps
    if not args.sweep_only:
        eval_workload(args, dev)
    sweep_workload(args, dev, dt)


def eval_workload(args, dev):
    true = dict(sigma_target=25.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5)
    with torch.no_grad():
        x = lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64, **true).simulate(0, n=50)
    p0 = dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4)
    va, ga = gradient.value_and_grad(x, lqg_amd.BoundedActor, p0, method="adjoint")
    vf, gf = gradient.value_and_grad(x, lqg_amd.BoundedActor, p0, method="fd")
    err = max(abs(ga[k] - gf[k]) / max(1.0, abs(gf[k])) for k in p0)
    import os
    # "fd": the default — batched central differences, the whole evaluation replayed as one hipGraph
    # (lqg_amd/infer/graphed.py); "fd_eager": the same arithmetic launched from Python (LQG_GRAPH=0)
    for label in ("adjoint", "fd", "fd_eager"):
        method = "fd" if label.startswith("fd") else label
        if label == "fd_eager":
            os.environ["LQG_GRAPH"] = "0"
        # (a different parameter vector every call: nothing may be cached across evaluations)
        it = iter(range(10 ** 9))
        sec = timed(lambda: gradient.value_and_grad(x, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 1e-3 * next(it)),
                                                    method=method), args.steps, args.warmup)
        os.environ.pop("LQG_GRAPH", None)
        method = label
        print(json.dumps({"metric": "value_and_grad_evals_per_s", "value": 1.0 / sec, "unit": "evals/s", "method": method,
                          "ms_per_eval": sec * 1e3, "dtype": "f64", "n_gpus": 1,
                          "config": {"workload": "BoundedActor T=500 x=b=2, 50 trials, 4 parameters, one parameter vector"},
                          "grad_vs_fd_max_rel": err, "baseline_note": "BASELINE.md derived ~54 evals/s (JAX CPU, not comparable)"}))



def sweep_workload(args, dev, dt):
    """B candidates, one (system, trial) pair each (BoundedActor, x=b=2, T=500): solves + reverse-mode gradient per second on a
    persistent lqg_amd.grad.GradPlan — the split sweep of round 5 (csrc/lqg_adjoint_sp.hpp) — and, beside it, the round-1 lane
    kernels (LQG_ADJOINT_SP=0).  The headline and config-3 shapes are legs of bench.py (`extra.value_and_grad_*`)."""
    from lqg_amd import options, workload
    B = 1 << args.log2_lanes
    sig = torch.linspace(5.0, 50.0, B, device=dev, dtype=dt)
    model = lqg_amd.BoundedActor(T=500, sigma_target=sig, device=dev, dtype=dt)
    with torch.no_grad():
        xs = lqg_amd.BoundedActor(T=500, sigma_target=25.0, device=dev, dtype=dt).simulate(1, n=1)          # [1, 501, 2]
    xs = workload.pack_trials(xs.expand(B, 1, 501, 2).contiguous())
    for sp in (1, 0):
        with options.override(ADJOINT_SP=sp):
            gp = G.GradPlan(model, xs, events=bool(sp))
            sec = timed(lambda: gp.run(), args.steps, args.warmup)
            rec = {"metric": "solves_with_gradient_per_s", "value": B / sec, "unit": "solves+adjoint/s", "ms_per_step": sec * 1e3,
                   "dtype": args.dtype, "n_gpus": 1, "adjoint_sp": sp, "path": gp.description,
                   "config": {"workload": f"{B} candidates x 1 trial, BoundedActor x=b=2, T=500"}}
            if sp:
                rec["kernel_ms"] = gp.phase_ms()
            print(json.dumps(rec))


if __name__ == "__main__":
    main()
