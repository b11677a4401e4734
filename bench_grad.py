"""Gradient bench (SURVEY.md §8f rank 1 / BASELINE.md's derived "≈54 value+grad evaluations/s" row): value AND
reverse-mode gradient of the summed log-likelihood.

  python bench_grad.py [--steps K] [--warmup W] [--dtype f64|f32]

Two workloads, one JSON line each:
  eval   the reference's inference inner loop: BoundedActor, T=500, 50 trials, 4 parameters, ONE parameter vector —
         `jax.value_and_grad(ll)` per NUTS leapfrog / Adam step (lqg/infer/utils.py:18, lqg/optim.py:142-147).
         Reported for the adjoint sweep and for the batched finite-difference sweep (2P+1 candidates).
  sweep  B independent (system, trial) pairs, one lane each (x=b=2, T=500): solves+adjoint per second, with the
         per-kernel times from HIP events.
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
    # ---- sweep: B lanes, one (system, trial) each
    B = 1 << args.log2_lanes
    sig = torch.linspace(5.0, 50.0, B, device=dev, dtype=dt)
    model = lqg_amd.BoundedActor(T=500, sigma_target=sig, device=dev, dtype=dt)
    with torch.no_grad():
        xs = lqg_amd.BoundedActor(T=500, sigma_target=25.0, device=dev, dtype=dt).simulate(1, n=1)          # [1, 501, 2]
    xs = xs.expand(B, 1, 501, 2).contiguous()
    fn = lambda: G.raw_grad(model.actor, model.dynamics, xs, g=None, want_value=True)
    sec = timed(fn, args.steps, args.warmup)
    # per-lane step traffic of the kept state: S 3 + L 2 + P 3 + Sigma 10 + mu 4 reals written once, read once (+L rewritten)
    reals = 3 + 2 + 3 + 10 + 4
    esz = 8 if dt == torch.float64 else 4
    bytes_ = B * 500 * (2 * reals + 2 * 2) * esz
    print(json.dumps({"metric": "solves_with_gradient_per_s", "value": B / sec, "unit": "solves+adjoint/s",
                      "ms_per_step": sec * 1e3, "dtype": args.dtype, "n_gpus": 1,
                      "config": {"workload": f"{B} (system, trial) lanes, BoundedActor x=b=2, T=500, four adjoint sweeps"},
                      "scratch_GB_per_step": bytes_ / 1e9, "scratch_GBps": bytes_ / sec / 1e9}))


if __name__ == "__main__":
    main()
