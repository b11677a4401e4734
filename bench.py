#!/usr/bin/env python3
"""bench.py — LQG solves/sec (Riccati + Kalman + log-likelihood), n=6, T=500, on N MI355X.

One "step" = one pass of the hot path (lqg_log_likelihood through the C ABI: k_riccati -> k_forward fused
with the per-trial density) over one batch of B independent (candidate, trajectory) pairs per GPU; inputs are
resident in HBM before the timed region.  Workload = BASELINE.json headline / config 5 shape:
SubjectiveActor(dim=2) (x=4, b=6, u=2, y=4, d=4), T=500, synthetic candidates, trajectories
simulated from the model.  Weak scaling: every rank owns B solves; the only collective is the all-reduce of
the summed log-likelihood (the objective of lqg.infer / lqg.optim), issued once per step.

    python bench.py [--gpus N --steps K --warmup W --log2-batch 20 --dtype f32|f64]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 matrix (MFMA f32) peak
PEAK_FP64_TFLOPS = 78.6    # datasheet FP64 vector == FP64 matrix peak
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_flops_per_step(x, b, u, y, d):
    """flops of ONE time step of the reference's formulation (2mnk per matmul, no symmetry
    exploitation, no hoisting) — the 'algorithmic' figure the roofline uses."""
    m, o = x + b, d
    ric = (4 * u * b * b + 2 * u * u * b + 2 * u * b + 10.67 * u ** 3 + 2 * u * u * (b + 1) + 4 * b ** 3
           + 2 * u * u * b + 4 * u * b * b + 2 * b * b + 6 * u * b + 2 * u * u)
    kal = 6 * b ** 3 + 2 * b ** 3 + 2 * y * b * b + 2 * y * y * b + 2 * y ** 3 + 2 * y ** 3 + 2 * b * y * y + 2 * b * b * y
    joint = (2 * x * u * b + 2 * b * y * x + 2 * b * x * x + 2 * b * b * u + 2 * b * b * y + 2 * b ** 3 + 2 * y * x * u
             + 2 * y * b * u + 2 * b * y * u + 2 * b * b * u + 2 * b * x * x + 2 * b * y * y)
    sig = 4 * m ** 3 + 2 * m * m * (x + y) + o ** 3 / 3 + 2 * o * o * m + 2 * m * m * o
    mean = 2 * m * m + 2 * m * o + o
    lp = d ** 3 / 3 + d * d + 3 * d
    return dict(riccati=ric, kalman=kal, joint=joint, sigma=sig, mean=mean, logprob=lp,
                total=ric + kal + joint + sig + mean + lp)


def algorithmic_bytes_per_solve(x, b, u, y, d, T, w):
    """SURVEY.md §8(d) mode M1 (time-invariant specs in, one scalar out)."""
    return w * ((T + 1) * d + (3 * b * b + b * u + y * b + y * y + u * u) + (2 * x * x + x * u + y * x + y * y) + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2-batch", type=int, default=20,
                    help="solves per GPU per step = 2**this (2^20: 14.7 GB resident; 2^18 fills each SIMD with exactly 4 waves "
                         "and runs ~12 %% slower per solve)")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--layout", default="packed", choices=["packed", "reference"],
                    help="trajectory layout in HBM: packed = [T+1][d][B] (batch fastest), reference = [B][T+1][d]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stack", action="store_true",
                    help="launch decoupled components separately instead of stacked into one launch")
    ap.add_argument("--cpu-sample", type=int, default=0, help="solves in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torchrun the collective path is exercised even at world == 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import lqg_amd
    from lqg_amd import _abi, _hip, _hipev, workload

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    w = 4 if args.dtype == "f32" else 8
    B, T = 1 << args.log2_batch, args.T
    lib = _abi.load()

    # ---- synthetic workload, resident in HBM
    system, params = workload.headline_system(B, T, seed=1234 + rank, device=dev, dtype=dtype)
    dm = dict(x=system.xdim, b=system.bdim, u=system.udim, y=system.ydim, d=system.xdim)
    x_ref = workload.simulate_one_trial_each(system, seed=99 + 7919 * rank)            # [B,1,T+1,d]
    x = workload.pack_trials(x_ref) if args.layout == "packed" else x_ref
    torch.cuda.synchronize()

    # The hot path exactly as lqg_amd.System.log_likelihood runs it (lqg_amd/plan.py), decided once:
    # (1) if the model's interaction graph splits into independent components (every dim=2 zoo model is two 1-D
    #     models) each component is solved on its own and the log-likelihoods add (lqg_amd/decouple.py;
    #     LQG_NO_DECOUPLE=1 disables); (2) each solve uses the structure-specialised library of its sparsity pattern
    #     when one exists (lqg_amd/specialize.py; LQG_NO_SPECIALIZE=1 forces the generic dense kernels).
    # Both are exact and both are derived from the spec DATA, not from the model's name.
    from lqg_amd.plan import LogLikelihoodPlan
    plan = LogLikelihoodPlan(system, x, events=True, stack=not args.no_stack)
    ll = plan.ll
    fwd_name = plan.description
    sp_all = all(wk["specialised"] for wk in plan.work)
    total = torch.zeros((), dtype=torch.float64, device=dev)

    def step():
        plan.run()
        s = _hip.sum_trials(ll.view(1, B))                 # objective = sum of log-likelihoods (fp64)
        if dist is not None:
            dist.all_reduce(s)                             # the one collective of the path (RCCL over xGMI)
        return s

    for _ in range(args.warmup):
        total = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # one set of phase events per timed step (up to 64), so that per-kernel durations cover the timed region without
    # any host synchronisation inside it
    n_ev = min(args.steps, 64)
    ev_sets = [[[_hipev.Event() for _ in range(4)] for _ in plan.work] for _ in range(n_ev)]
    t0 = time.perf_counter()
    for it in range(args.steps):
        if it < n_ev:
            plan.use_events(ev_sets[it])
        total = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ric_ms = [sum(e[0].elapsed_ms(e[1]) for e in es) for es in ev_sets]
    fwd_ms = [sum(e[1].elapsed_ms(e[2]) for e in es) for es in ev_sets]

    ll_host = ll[:, 0].double().cpu().numpy()
    finite = bool(np.isfinite(ll_host).all())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    solves = float(B) * world * args.steps
    value = solves / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    fl = algorithmic_flops_per_step(**dm)
    flops_solve = fl["total"] * T
    bytes_solve = algorithmic_bytes_per_solve(T=T, w=w, **dm)
    fwd_avg_ms = float(np.mean(fwd_ms))
    ric_avg_ms = float(np.mean(ric_ms))
    peak = PEAK_FP32_TFLOPS if args.dtype == "f32" else PEAK_FP64_TFLOPS
    # dominant kernel = k_forward: Kalman + joint + Sigma recursion + mean + log-density for B solves per launch
    fwd_flops_launch = (fl["total"] - fl["riccati"]) * T * B
    achieved_tflops = fwd_flops_launch / (fwd_avg_ms * 1e-3) / 1e12
    hbm_gbs = bytes_solve * B / ((fwd_avg_ms + ric_avg_ms) * 1e-3) / 1e9
    # PMC-derived figures of the dominant kernel, collected in separate rocprofv3 --pmc passes of this same command and
    # committed under profiles/ (FETCH_SIZE x2 gfx950 correction, calibrated there): HBM bytes and VALU instructions
    traffic = executed = None
    n_launch = len(plan.work)                       # forward-kernel launches per step
    alg_gbs = bytes_solve * B / (fwd_avg_ms * n_launch * 1e-3) / 1e9     # algorithmic bytes of one step / forward-kernel time
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pj = json.load(open(pmc_path))
            key = f"{'k_forward_sp' if sp_all else 'k_forward'}_x{n_launch}_{args.dtype}_log2B{args.log2_batch}"
            if plan.n_stacked > 1:
                key = f"k_forward_sp_stacked{plan.n_stacked}_{args.dtype}_log2B{args.log2_batch}"
            if plan.merged and max(plan.merged) > 1:
                key = f"k_forward_sp_merged{max(plan.merged)}_{args.dtype}_log2B{args.log2_batch}"
            elif n_launch == 1 and key not in pj and not sp_all:
                key = f"k_forward_{args.dtype}_log2B{args.log2_batch}"
            rec = pj.get(key, {})
            traffic = rec.get("hbm_bytes_per_launch")
            if "valu_wave_insts_per_launch" in rec:
                rate = rec["valu_wave_insts_per_launch"] * n_launch / (fwd_avg_ms * 1e-3)
                executed = {"valu_wave_insts_per_launch": rec["valu_wave_insts_per_launch"],
                            "valu_issue_frac": rate / (1024 * 2.4e9 / 2),
                            "note": "wave64 VALU instructions issued per second / (1024 SIMDs x 2.4 GHz / 2 cycles)"}
        except Exception:
            traffic = None
    if traffic is not None:
        hbm_gbs = traffic * n_launch / (fwd_avg_ms * 1e-3) / 1e9    # measured bytes of the forward launches / their time

    # ---- parity spot check against the CPU oracle (not timed)
    parity = None
    cpu = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as OC
        OC.build()
        ns = 64
        idx = np.linspace(0, B - 1, ns).astype(np.int64)
        sub = workload.slice_system  # noqa
        actor_np = {f: getattr(system.actor, f) for f in lqg_amd.LQGSpec._fields}
        dyn_np = {f: getattr(system.dynamics, f) for f in lqg_amd.LQGSpec._fields}

        def host(spec_t, sel, np_dt=np.float64):
            """Selected systems as NumPy arrays; a time-invariant (stride-0) time axis stays a stride-0 broadcast."""
            out = {}
            index = torch.as_tensor(sel)
            for f, t in spec_t.items():
                nd = workload._batched_ndim(f)
                has_t = f not in ("Qf", "qf")
                tax = -(2 if f in ("q", "r") else 3)
                ti = has_t and (t.stride(tax) == 0 or t.shape[tax] == 1)
                base = t.select(tax, 0) if ti else t
                nd_b = nd - (1 if ti else 0)
                base = base[index.to(base.device)] if base.dim() == nd_b else base.expand(len(sel), *base.shape)
                a = base.cpu().numpy().astype(np_dt)
                if ti:
                    k = a.ndim + tax + 1
                    a = np.broadcast_to(np.expand_dims(a, k), a.shape[:k] + (t.shape[tax],) + a.shape[k:])
                out[f] = a
            return out

        a64, d64 = host(actor_np, idx), host(dyn_np, idx)
        x64 = x_ref[torch.as_tensor(idx, device=dev)].double().cpu().numpy()
        ref = OC.log_likelihood(a64, d64, x64, dtype=np.float64)[:, 0]
        got = ll_host[idx]
        parity = dict(samples=int(ns), max_rel_err_vs_fp64_oracle=float(np.abs(got / ref - 1).max()))
        if not args.no_cpu_baseline and world == 1:
            ncpu = os.cpu_count() or 1
            OC.lib().lqg_oracle_set_threads(ncpu)
            np_dt = np.float32 if args.dtype == "f32" else np.float64
            nsamp = args.cpu_sample or 16384          # 64 solves per thread on a 256-thread host
            sel = np.arange(nsamp) % B
            a_s, d_s = host(actor_np, sel, np_dt), host(dyn_np, sel, np_dt)
            x_s = x_ref[torch.as_tensor(sel, device=dev)].cpu().numpy().astype(np_dt)
            OC.log_likelihood({k: v[:8] for k, v in a_s.items()}, {k: v[:8] for k, v in d_s.items()}, x_s[:8], dtype=np_dt)
            tc = time.perf_counter()
            OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
            tc1 = time.perf_counter() - tc
            # repeat the sample so that the CPU leg does ~10-20 s of work
            if tc1 < 8.0 and not args.cpu_sample:
                rep = int(min(16, max(1, 12.0 / max(tc1, 1e-3))))
                tc = time.perf_counter()
                for _ in range(rep):
                    OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
                tc1 = (time.perf_counter() - tc) / rep
                nrep = rep
            else:
                nrep = 1
            model = ""
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            cpu = dict(value=nsamp / tc1, unit="solves/s", cores=OC.lib().lqg_oracle_max_threads(), kind="port",
                       sample=f"{nsamp} solves of the same workload ({args.dtype}, T={T}) x {nrep} repetitions, "
                              f"oracle/lqg_oracle.c with OpenMP over systems", cpu_model=model,
                       host_cpu_count=ncpu)
    except Exception as e:  # the oracle is a checker: its absence must not break the measurement
        parity = dict(error=repr(e))

    out = {
        "metric": "LQG solves/sec (Riccati+Kalman+loglik), n=6 T=500",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SubjectiveActor(dim=2) x=4 b=6 u=2 y=4 d=4, T={T}, {B} independent "
                               f"(candidate, trajectory) solves per GPU per step (BASELINE config 5 / headline shape)",
                   "solves_per_gpu": B, "T": T, "trajectory_layout": args.layout,
                   "path": fwd_name,
                   "parallelism": f"candidate-sharded x{world}, all-reduce of the summed log-likelihood"},
        # Contract form: bound in {hbm, mfma}; achieved = ALGORITHMIC bytes per launch (mode M1: trajectory
        # in, specs in, one scalar out) / the dominant kernel's HIP-event time.  The kernel is VALU-issue-bound (SURVEY 8d
        # says so by construction for M1), so the honest reading is in `valu` (flop view, executed-instruction issue rate)
        # and `hbm_measured` (PMC bytes / time: how busy HBM really is, incl. the gain stream L_t between the two sweeps).
        "roofline": {"bound": "hbm", "achieved": alg_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": alg_gbs / PEAK_HBM_GBS, "traffic": traffic,
                     "kernel": "forward sweep (Kalman + joint system + Sigma recursion + mean + log-density) of: " + fwd_name,
                     "kernel_ms": fwd_avg_ms, "riccati_kernel_ms": ric_avg_ms,
                     "algorithmic_bytes_per_solve": bytes_solve, "algorithmic_bytes_per_launch": bytes_solve * B,
                     "kernel_launches_per_step": n_launch,
                     "hbm_measured": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": hbm_gbs / PEAK_HBM_GBS,
                                      "basis": "PMC traffic of the forward launches / their time" if traffic is not None
                                      else "algorithmic bytes (M1) / (riccati + forward time)"},
                     "valu": {"achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tflops / peak,
                              "algorithmic_flops_per_solve": flops_solve, "executed": executed,
                              "note": "ALGORITHMIC flops of the reference formulation (SURVEY.md 8d: dense, no symmetry, "
                                      "no hoisting) / kernel time; the structure-specialised + decoupled kernels execute "
                                      "~20x fewer, so this frac exceeds 1 - `executed.valu_issue_frac` is the utilisation"},
                     "note": "M1 is VALU-issue-bound, not HBM- or MFMA-bound (MFMA deliberately unused: contractions are "
                             "<= 6x6 per lane). traffic > algorithmic bytes because the control gains L_t travel from the "
                             "backward to the forward sweep through HBM (6 kB/solve), which M1's figure does not count"},
        "cpu_baseline": cpu, "parity": parity, "all_finite": finite,
        "objective_sum": float(total.item()),
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
