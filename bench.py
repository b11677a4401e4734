#!/usr/bin/env python3
"""bench.py — LQG solves/sec (Riccati + Kalman + log-likelihood), n=6, T=500, on N MI355X.

One "step" = one pass of the hot path (lqg_log_likelihood through the C ABI: Riccati sweep -> forward sweep fused
with the per-trial density) over one batch of B independent (candidate, trajectory) pairs per GPU; inputs are
resident in HBM before the timed region.  Workload = BASELINE.json headline / config 5 shape:
SubjectiveActor(dim=2) (x=4, b=6, u=2, y=4, d=4), T=500, synthetic candidates, trajectories
simulated from the model.  Weak scaling: every rank owns B solves; the only collective is the all-reduce of
the summed log-likelihood (the objective of lqg.infer / lqg.optim), issued once per step.

    python bench.py [--gpus N --steps K --warmup W --log2-batch 20 --dtype f32|f64]
    python bench.py --gpus 8                 # bare: starts 8 ranks itself (torch.distributed.run), relays rank 0's line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config 3 [--gpus N]    # BASELINE config 3: 4096 candidates x 1024 trials, trial axis split
                                             # over the ranks, all-reduce of the [4096] fp64 objective (strong scaling)
    python bench.py --gpus 2 --share-gpu     # the same N-rank flow on ONE GPU: every rank computes on device 0 and the
                                             # reduce goes through gloo on a host tensor (multi-rank evidence without a
                                             # multi-GPU node; the per-GPU rates are NOT a scaling measurement)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 matrix (MFMA f32) peak
PEAK_FP64_TFLOPS = 78.6    # datasheet FP64 vector == FP64 matrix peak
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COPY_HBM_GBS = 6290.0      # MI355X_MICROARCH.md: measured device copy rate
VALU_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 2    # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")


def algorithmic_flops_per_step(x, b, u, y, d):
    """flops of ONE time step of the reference's formulation (2mnk per matmul, no symmetry
    exploitation, no hoisting) — the 'algorithmic' figure of the flop view."""
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=5, choices=[3, 5],
                    help="5 = the headline (BASELINE metric); 3 = candidate search, trial axis split over the ranks")
    ap.add_argument("--log2-batch", type=int, default=20,
                    help="solves per GPU per step = 2**this (2^20: 14.7 GB resident; 2^18 fills each SIMD with exactly 4 waves "
                         "and runs ~12 %% slower per solve)")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--layout", default="packed", choices=["packed", "reference"],
                    help="trajectory layout in HBM: packed = [T+1][d][B] (batch fastest), reference = [B][T+1][d]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (fp64 headline, dense generic n=6)")
    ap.add_argument("--no-stack", action="store_true",
                    help="launch decoupled components separately instead of stacked into one launch")
    ap.add_argument("--cpu-sample", type=int, default=0, help="solves in the CPU baseline sample (0 = auto)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks compute on device 0 (backend gloo, the objective reduced on a host tensor): runs the N>1 "
                         "flow with the HIP kernels on a single-GPU box")
    return ap.parse_args()


class HostReduce:
    """--share-gpu: the three collectives of this file (all_reduce of the fp64 objective, barrier, all_gather of one
    timing) over gloo on HOST tensors, behind the torch.distributed calls the RCCL path makes on device tensors."""

    def __init__(self, torch, dist):
        self._t, self._d = torch, dist

    def get_world_size(self):
        return self._d.get_world_size()

    def barrier(self):
        self._d.barrier()

    def all_reduce(self, t):
        h = t.detach().cpu()                      # (stream-ordered copy: waits for the partial sums)
        self._d.all_reduce(h)
        t.copy_(h)

    def all_gather(self, outs, t):
        hs = [self._t.zeros_like(t, device="cpu") for _ in outs]
        self._d.all_gather(hs, t.detach().cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)

    def destroy_process_group(self):
        self._d.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` run bare: start N ranks as CHILD processes (one per GPU, torch.distributed.run) and
    relay their output.  Nothing in this process has touched the GPU (no HIP call, no torch.cuda.is_available()), and it
    never execs: it waits for the children and exits with their code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def pmc_record(kernel, pattern_key, dtype, log2_batch):
    """The PMC-derived HBM bytes / VALU instructions of `kernel`, but only if the committed profile was taken on
    exactly these kernels: same library source hash, same sparsity-pattern library, dtype and batch."""
    try:
        from lqg_amd import build, specialize
        pj = json.load(open(PMC_FILE))
        for rec in pj.get("records", []):
            if (rec.get("kernel") == kernel and rec.get("dtype") == dtype and rec.get("log2_batch") == log2_batch
                    and rec.get("pattern_key") == pattern_key and rec.get("source_hash") == build.source_hash()
                    and rec.get("sp_headers_hash") == specialize._headers_hash()):
                return rec
    except Exception:
        pass
    return None


class Timer:
    """Per-step durations on the launch stream without host synchronisation: one event pair per step."""

    def __init__(self, torch, n):
        self.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]

    def ms(self):
        return [a.elapsed_time(b) for a, b in self.ev]


def timed_steps(torch, dist, step, steps, warmup, before_step=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    out = None
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    tm = Timer(torch, steps)
    t0 = time.perf_counter()
    for it in range(steps):
        if before_step is not None:
            before_step(it)
        tm.ev[it][0].record()
        out = step()
        tm.ev[it][1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    elapsed, per_rank = mine, [mine]
    if dist is not None:
        t = torch.tensor([mine], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        per_rank = [float(v.item()) for v in allt]
        elapsed = max(per_rank)
    return out, elapsed, per_rank, tm.ms()


def allreduce_us(torch, dist, n, reps=50):
    """Mean latency of the path's one collective: all-reduce of n fp64 values (RCCL over xGMI)."""
    if dist is None:
        return None
    v = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(5):
        dist.all_reduce(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.all_reduce(v)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def host_specs(torch, np, workload, LQGSpec, system, sel, np_dt):
    """Selected systems of both specs as NumPy arrays; a time-invariant (stride-0) time axis stays a stride-0 broadcast."""
    def host(spec):
        out = {}
        index = torch.as_tensor(sel)
        for f in LQGSpec._fields:
            t = getattr(spec, f)
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
    return host(system.actor), host(system.dynamics)


def headline_leg(torch, dist, args, dev, rank, world, dtype_name, log2_batch, steps, warmup, env=None, cpu=False):
    """One measurement of the headline workload; returns the dict rank 0 prints (None on other ranks)."""
    import numpy as np
    import lqg_amd
    from lqg_amd import _hip, _hipev, workload
    from lqg_amd.plan import LogLikelihoodPlan

    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        dtype = torch.float32 if dtype_name == "f32" else torch.float64
        w = 4 if dtype_name == "f32" else 8
        B, T = 1 << log2_batch, args.T
        system, _ = workload.headline_system(B, T, seed=1234 + rank, device=dev, dtype=dtype)
        dm = dict(x=system.xdim, b=system.bdim, u=system.udim, y=system.ydim, d=system.xdim)
        x_ref = workload.simulate_one_trial_each(system, seed=99 + 7919 * rank)            # [B,1,T+1,d]
        x = workload.pack_trials(x_ref) if args.layout == "packed" else x_ref
        torch.cuda.synchronize()
        # The hot path exactly as lqg_amd.System.log_likelihood runs it (lqg_amd/plan.py), decided once:
        # (1) a model whose interaction graph splits into independent components is solved per component (every dim=2
        #     zoo model is two 1-D models; identical components become trials of ONE system), LQG_NO_DECOUPLE=1 disables;
        # (2) each solve uses the structure-specialised library of its sparsity pattern (LQG_NO_SPECIALIZE=1 forces the
        #     generic dense kernels).  Both are exact and derived from the spec DATA, not from the model's name.
        plan = LogLikelihoodPlan(system, x, events=True, stack=not args.no_stack)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ll = plan.ll
    sp_all = all(wk["specialised"] for wk in plan.work)

    def step():
        plan.run()
        s = _hip.sum_trials(ll.view(1, B))                 # objective = sum of log-likelihoods (fp64)
        if dist is not None:
            dist.all_reduce(s)                             # the one collective of the path (RCCL over xGMI)
        return s

    n_ev = min(steps, 64)
    ev_sets = [[[_hipev.Event() for _ in range(4)] for _ in plan.work] for _ in range(n_ev)]
    total, elapsed, per_rank, step_ms = timed_steps(
        torch, dist, step, steps, warmup, before_step=lambda it: plan.use_events(ev_sets[it]) if it < n_ev else None)
    ric_ms = [sum(e[0].elapsed_ms(e[1]) for e in es) for es in ev_sets]
    fwd_ms = [sum(e[1].elapsed_ms(e[2]) for e in es) for es in ev_sets]
    ll_host = ll[:, 0].double().cpu().numpy()
    ar_us = allreduce_us(torch, dist, 1)
    local_obj = _hip.sum_trials(ll.view(1, B)).reshape(1)      # this rank's partial objective (what it fed the all-reduce)
    per_rank_obj = [float(local_obj.item())]
    if dist is not None:
        allo = [torch.zeros_like(local_obj) for _ in range(dist.get_world_size())]
        dist.all_gather(allo, local_obj)
        per_rank_obj = [float(v.item()) for v in allo]
    if rank != 0:
        return None

    fl = algorithmic_flops_per_step(**dm)
    bytes_solve = algorithmic_bytes_per_solve(T=T, w=w, **dm)
    fwd_avg_ms, ric_avg_ms = float(np.mean(fwd_ms)), float(np.mean(ric_ms))
    n_launch = len(plan.work)                       # forward-kernel launches per step
    alg_gbs = bytes_solve * B / (fwd_avg_ms * 1e-3) / 1e9     # algorithmic bytes of one step / forward-kernel time
    kernel = ("k_forward_sp" if sp_all else "k_forward") + \
        (f"<merged{max(plan.merged)}>" if plan.merged and max(plan.merged) > 1 else "") + f"x{n_launch}"
    pkey = plan.work[0].get("pattern_key") if sp_all else "generic"
    rec = pmc_record(kernel, pkey, dtype_name, log2_batch)
    traffic = rec["hbm_bytes_per_launch"] if rec else None
    limits = {"note": "which side binds the dominant kernel: measured HBM bytes/s against the guide's achievable copy rate "
                      "vs wave64 VALU instructions issued/s against 1024 SIMDs x 2.4 GHz / 2 cycles"}
    if rec:
        limits["hbm_frac_of_copy_rate"] = traffic / (fwd_avg_ms * 1e-3) / 1e9 / COPY_HBM_GBS
        limits["hbm_frac_of_peak"] = traffic / (fwd_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        if rec.get("valu_wave_insts_per_launch"):
            limits["valu_issue_frac"] = rec["valu_wave_insts_per_launch"] / (fwd_avg_ms * 1e-3) / VALU_WAVE_INSTS_PER_S
            limits["valu_insts_per_step_per_wave"] = rec.get("valu_insts_per_step_per_wave")
        limits["profile"] = rec.get("profile")
    solves = float(B) * world * steps
    out = {
        "metric": "LQG solves/sec (Riccati+Kalman+loglik), n=6 T=500",
        "value": solves / elapsed, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "ms_per_step_min": float(np.min(step_ms)),
        "ms_per_step_median": float(np.median(step_ms)), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"SubjectiveActor(dim=2) x=4 b=6 u=2 y=4 d=4, T={T}, {B} independent "
                               f"(candidate, trajectory) solves per GPU per step (BASELINE config 5 / headline shape)",
                   "solves_per_gpu": B, "T": T, "trajectory_layout": args.layout, "path": plan.description,
                   "parallelism": f"candidate-sharded x{world}, all-reduce of the summed log-likelihood"},
        "world_size": world if dist is None else dist.get_world_size(),
        "per_rank_solves_per_s": [float(B) * steps / t for t in per_rank], "allreduce_us": ar_us,
        "share_gpu": bool(args.share_gpu),
        # Contract form: achieved = ALGORITHMIC bytes per launch (mode M1: trajectory in, specs in, one
        # scalar out) / the dominant kernel's HIP-event time; traffic = PMC bytes of that launch (null unless the
        # committed profile was taken on exactly this build).
        "roofline": {"bound": "hbm", "achieved": alg_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": alg_gbs / PEAK_HBM_GBS, "traffic": traffic,
                     "kernel": "forward sweep (Kalman + joint system + Sigma recursion + mean + log-density) of: "
                               + plan.description,
                     "kernel_ms": fwd_avg_ms, "riccati_kernel_ms": ric_avg_ms,
                     "algorithmic_bytes_per_solve": bytes_solve, "algorithmic_bytes_per_launch": bytes_solve * B,
                     "kernel_launches_per_step": n_launch, "limits": limits,
                     "algorithmic_flops_per_solve": fl["total"] * T},
        "all_finite": bool(np.isfinite(ll_host).all()), "objective_sum": float(total.item()),
        "per_rank_objective": per_rank_obj,
    }
    # ---- parity spot check against the CPU oracle (not timed) and, on the main leg, the CPU baseline
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as OC
        OC.build()
        ns = 64
        idx = np.linspace(0, B - 1, ns).astype(np.int64)
        a64, d64 = host_specs(torch, np, workload, lqg_amd.LQGSpec, system, idx, np.float64)
        x64 = x_ref[torch.as_tensor(idx, device=dev)].double().cpu().numpy()
        ref = OC.log_likelihood(a64, d64, x64, dtype=np.float64)[:, 0]
        out["parity"] = dict(samples=int(ns), max_rel_err_vs_fp64_oracle=float(np.abs(ll_host[idx] / ref - 1).max()))
        if cpu:
            ncpu = os.cpu_count() or 1
            OC.lib().lqg_oracle_set_threads(ncpu)
            np_dt = np.float32 if dtype_name == "f32" else np.float64
            nsamp = args.cpu_sample or 16384          # 64 solves per thread on a 256-thread host
            sel = np.arange(nsamp) % B
            a_s, d_s = host_specs(torch, np, workload, lqg_amd.LQGSpec, system, sel, np_dt)
            x_s = x_ref[torch.as_tensor(sel, device=dev)].cpu().numpy().astype(np_dt)
            OC.log_likelihood({k: v[:8] for k, v in a_s.items()}, {k: v[:8] for k, v in d_s.items()}, x_s[:8], dtype=np_dt)
            tc = time.perf_counter()
            OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
            tc1 = time.perf_counter() - tc
            nrep = 1
            if tc1 < 8.0 and not args.cpu_sample:     # repeat the sample so that the CPU leg does ~10-20 s of work
                nrep = int(min(16, max(1, 12.0 / max(tc1, 1e-3))))
                tc = time.perf_counter()
                for _ in range(nrep):
                    OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
                tc1 = (time.perf_counter() - tc) / nrep
            model = ""
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            out["cpu_baseline"] = dict(
                value=nsamp / tc1, unit="solves/s", cores=OC.lib().lqg_oracle_max_threads(), kind="port",
                sample=f"{nsamp} solves of the same workload ({dtype_name}, T={T}) x {nrep} repetitions, "
                       f"oracle/lqg_oracle.c (literal dense restatement) with OpenMP over systems",
                cpu_model=model, host_cpu_count=ncpu)
    except Exception as e:  # the oracle is a checker: its absence must not break the measurement
        out["parity"] = dict(error=repr(e))
    return out


def config3(torch, dist, args, dev, rank, world):
    """BASELINE config 3 / SURVEY §8(e): 4096 parameter candidates x 1024 data.mat-shaped trials (1068 rows).  The TRIAL
    axis is split over the ranks (every rank holds all candidates and 1024/N trials: x is read once per rank, the
    4096 per-system sweeps are replicated); one all-reduce of the [4096] fp64 objective per step.  Strong scaling."""
    import numpy as np
    import lqg_amd
    from lqg_amd import _hip, workload
    from lqg_amd.plan import LogLikelihoodPlan

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    Bc, Nt, T = 4096, 1024, 1067
    if Nt % world:
        raise SystemExit(f"--config 3: {Nt} trials do not split over {world} ranks")
    system, _ = workload.bounded_system(Bc, T, seed=5, device=dev, dtype=dtype)       # the same candidates on every rank
    truth = lqg_amd.BoundedActor(T=T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, action_variability=0.5,
                                 device=dev, dtype=dtype)
    x_all = truth.simulate(13, n=Nt)                                                  # same seed: same data on every rank
    lo = rank * (Nt // world)
    x = workload.pack_trials(x_all[lo:lo + Nt // world].contiguous())
    plan = LogLikelihoodPlan(system, x, events=True)

    def step():
        obj = _hip.sum_trials(plan.run())            # [4096] fp64 partial objective of this rank's trials
        if dist is not None:
            dist.all_reduce(obj)
        return obj

    obj, elapsed, per_rank, step_ms = timed_steps(torch, dist, step, args.steps, args.warmup)
    ph = plan.phase_ms()
    ar_us = allreduce_us(torch, dist, Bc)
    if rank != 0:
        return None
    w = 4 if args.dtype == "f32" else 8
    evals = float(Bc) * Nt * args.steps
    bytes_launch = (Nt // world) * (T + 1) * 2 * w            # the data rows; operators come from the scalar cache
    return {
        "metric": "trial-evals/sec (candidate search: 4096 candidates x 1024 trials, T=1067)",
        "value": evals / elapsed, "unit": "trial-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_min": float(np.min(step_ms)),
        "ms_per_step_median": float(np.median(step_ms)), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE config 3: BoundedActor x=b=2 u=1 y=2 d=2, data.mat-shaped trials (1068 rows), "
                               f"{Bc} candidates x {Nt} shared trials; objective = sum over trials per candidate",
                   "candidates": Bc, "trials": Nt, "trials_per_rank": Nt // world, "T": T, "path": plan.description,
                   "parallelism": f"trial-split x{world}, all-reduce of the [{Bc}] fp64 objective"},
        "world_size": world if dist is None else dist.get_world_size(),
        "per_rank_s": per_rank, "allreduce_us": ar_us, "share_gpu": bool(args.share_gpu),
        "objective_checksum": float(obj.sum()),
        "phase_ms": {"riccati": ph[0], "forward": ph[1], "trial": ph[2]},
        "roofline": {"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "kernel": "k_trial (per-trial mean recursion + density over the operator stream)",
                     "note": "the per-trial sweep reads each data row once per CANDIDATE BLOCK from L2/HBM and its operators "
                             f"from the scalar cache; algorithmic bytes of the data alone = {bytes_launch} B per rank — "
                             "the sweep is VALU-bound (bench_configs.py carries the per-config accounting)"},
        "best_candidate": int(obj.argmax()), "objective_max": float(obj.max()),
    }


def extra_legs(torch, args, dev):
    """Secondary legs, outside the headline's timed region (same workload generator, same timing protocol): the other
    dtype of the headline, and the generic dense joint n=6 kernels (no specialisation, no decoupling) in both dtypes."""
    extra = {}
    other = "f64" if args.dtype == "f32" else "f32"
    torch.cuda.empty_cache()
    leg = headline_leg(torch, None, args, dev, 0, 1, other, args.log2_batch, min(args.steps, 10), 2)
    extra[f"headline_{other}"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "dtype", "parity", "roofline")}
    extra[f"headline_{other}"]["path"] = leg["config"]["path"]
    torch.cuda.empty_cache()
    for dn in ("f32", "f64"):
        leg = headline_leg(torch, None, args, dev, 0, 1, dn, 17, min(args.steps, 5), 1,
                           env={"LQG_NO_SPECIALIZE": "1", "LQG_NO_DECOUPLE": "1"})
        r = leg["roofline"]
        # dense joint n=6 (m=10) kernel: VALU-bound; executed-instruction figures come from the ISA of
        # k_forward<R,4,6,2,4,4,TI,FUSED> (profiles/README.md), the algorithmic flop rate from SURVEY 8(d)
        fl_rate = r["algorithmic_flops_per_solve"] * leg["value"] / 1e12
        extra[f"dense_generic_{dn}"] = {
            "value": leg["value"], "unit": leg["unit"], "ms_per_step": leg["ms_per_step"], "dtype": dn,
            "solves_per_gpu": 1 << 17, "path": leg["config"]["path"], "parity": leg.get("parity"),
            "env": "LQG_NO_SPECIALIZE=1 LQG_NO_DECOUPLE=1",
            "roofline": {"bound": "valu", "achieved": fl_rate, "unit": "TFLOP/s (algorithmic, SURVEY 8d)",
                         "peak": PEAK_FP32_TFLOPS if dn == "f32" else PEAK_FP64_TFLOPS,
                         "frac": fl_rate / (PEAK_FP32_TFLOPS if dn == "f32" else PEAK_FP64_TFLOPS),
                         "kernel_ms": r["kernel_ms"], "riccati_kernel_ms": r["riccati_kernel_ms"]}}
        torch.cuda.empty_cache()
    # one system x many trials (BASELINE configs 2 and 4, the inner loop of MLE / NUTS): the time-parallel path (scans over
    # the time axis + time-chunked per-trial sweep) against the sequential lane kernels, wall ms per evaluation
    import bench_configs as bc
    import lqg_amd
    from lqg_amd import workload
    for cfg in (2, 4):
        if cfg == 2:
            m = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=torch.float32)
            x = workload.pack_trials(m.simulate(12, n=65536)[..., :2].contiguous())
        else:
            m = bc.hand2d_system(1000, dev, torch.float32)
            x = workload.pack_trials(m.simulate(14, n=32768)[..., :4].contiguous())
        leg = {"systems": 1, "trials": int(x.shape[-3]), "T": int(m.T), "dtype": "f32"}
        for name, env in (("time_parallel", {}), ("sequential", {"LQG_SCAN": "0", "LQG_TRIAL_CHUNKS": "0"})):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                ll, ph = bc.timed_loglik(m, x, 10)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            leg[name] = {"wall_ms": ph["wall_ms"], "system_sweeps_ms": ph["riccati_ms"] + ph["forward_ms"],
                         "per_trial_sweep_ms": ph["trial_ms"], "path": ph["path"],
                         "max_rel_err_vs_fp64_oracle": bc.oracle_check(m, x, ll, n_samples=4)}
        leg["speedup"] = leg["sequential"]["wall_ms"] / leg["time_parallel"]["wall_ms"]
        extra[f"config{cfg}_one_system"] = leg
        del x
        torch.cuda.empty_cache()
    # the inner loop of the reference's inference drivers (lqg/infer/mle.py:17-23, NUTS): value + gradient of ONE parameter
    # vector on 50 trials, T = 500, fp64 — batched central differences replayed as one hipGraph (lqg_amd/infer/graphed.py),
    # the same launched from Python, and the reverse-mode sweep; BASELINE.md derives ~54 evaluations/s for the reference
    import time
    from lqg_amd.infer import gradient
    true = dict(sigma_target=25.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5)
    with torch.no_grad():
        xg = lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64, **true).simulate(0, n=50)
    xg = torch.cat([xg, xg[:, -1:]], dim=1)                       # lqg_model's convention: T rows = T - 1 steps
    p0 = dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4)
    leg = {"workload": "BoundedActor T=500, 50 trials, 4 parameters, one parameter vector, fp64", "unit": "ms per value+gradient"}
    for name, method, env, reps in (("fd_graph", "fd", {}, 300), ("fd_eager", "fd", {"LQG_GRAPH": "0"}, 100), ("adjoint", "adjoint", {}, 50)):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            for i in range(5):
                gradient.value_and_grad(xg, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 0.01 * i), method=method)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):          # (a different vector every call: nothing is cached across evaluations)
                gradient.value_and_grad(xg, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 1e-3 * i), method=method)
            torch.cuda.synchronize()
            leg[name] = (time.perf_counter() - t0) / reps * 1e3
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    leg["evaluations_per_s"] = 1e3 / leg["fd_graph"]
    extra["one_vector_value_and_grad"] = leg
    torch.cuda.empty_cache()
    # ---- the measurement claims of DESIGN.md §6 that used to be builder-run only (round-2 review item 4) -------------------
    # mode M2 of SURVEY 8(d): genuinely time-varying specs in, L, H, K, mu, Sigma out, [T][element][system] storage
    try:
        import bench_m2
        extra["m2_f32"] = bench_m2.run(dev, "f32", 17, 500, 3)
    except Exception as e:
        extra["m2_f32"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # BASELINE config 3 at its literal shape (4096 candidates x 1024 trials, T = 1067): trial-evals/s, k_trial_sp time
    try:
        class A3:
            dtype, steps, warmup, share_gpu = "f32", 10, 2, False
        c3 = config3(torch, None, A3, dev, 0, 1)
        w3 = 4
        tb = 4096 * 1024 * (1067 + 1) * 2 * w3                 # one data row per (candidate, trial, step) as the sweep reads it
        c3["roofline"] = {"bound": "valu", "kernel": "k_trial_sp (per-trial mean recursion + density over the operator stream)",
                          "kernel_ms": c3["phase_ms"]["trial"], "system_sweeps_ms": c3["phase_ms"]["riccati"] + c3["phase_ms"]["forward"],
                          "achieved": tb / (c3["phase_ms"]["trial"] * 1e-3) / 1e9, "unit": "GB/s of data rows through L2 (x is 8.7 MB: "
                          "HBM sees it once per pass)", "peak": None, "frac": None, "traffic": None,
                          "trial_steps_per_s": 4096 * 1024 * 1067 / (c3["phase_ms"]["trial"] * 1e-3)}
        extra["config3"] = {k: c3[k] for k in ("metric", "value", "unit", "ms_per_step", "phase_ms", "roofline", "config",
                                                "best_candidate")}
    except Exception as e:
        extra["config3"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # BASELINE config 5 in its literal form: ONE system x 2^20 trials (the per-trial sweep streams x once: HBM-bound)
    try:
        m5 = lqg_amd.SubjectiveActor(dim=2, T=500, device=dev, dtype=torch.float32)
        x5 = workload.pack_trials(m5.simulate(15, n=1 << 20))
        ll5, ph5 = bc.timed_loglik(m5, x5, 10)
        b5 = (1 << 20) * 501 * 4 * 4 + (1 << 20) * 4
        extra["config5_one_system"] = {
            "systems": 1, "trials": 1 << 20, "T": 500, "dtype": "f32", "wall_ms": ph5["wall_ms"], "path": ph5["path"],
            "value": (1 << 20) / (ph5["wall_ms"] * 1e-3), "unit": "trial-evals/s",
            "roofline": {"bound": "hbm", "kernel": "per-trial sweep (k_trial_sp) over 2^20 trajectories", "kernel_ms": ph5["trial_ms"],
                         "system_sweeps_ms": ph5["riccati_ms"] + ph5["forward_ms"],
                         "achieved": b5 / (ph5["trial_ms"] * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": b5 / (ph5["trial_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": b5},
            "max_rel_err_vs_fp64_oracle": bc.oracle_check(m5, x5, ll5, n_samples=4)}
        del x5
    except Exception as e:
        extra["config5_one_system"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # the reference's largest model: DelayedSubjectiveActor (delay 12: x = 26, b = 39, m = 65): time-parallel system sweeps on
    # windows of 39 / 63 (k_scan_level_rt) + the row-parallel per-trial sweep
    try:
        from lqg_amd.tracking.delay import DelayedSubjectiveActor
        md = DelayedSubjectiveActor(T=500, device=dev, dtype=torch.float32)
        legd = {"model": "DelayedSubjectiveActor (lqg/tracking/delay.py:44-51): x=26 b=39 m=65, T=500", "dtype": "f32",
                "unit": "ms per log-likelihood evaluation"}
        xd_all = md.simulate(21, n=256)[..., :2].contiguous()
        for nt in (1, 256):
            xd = xd_all[:nt].contiguous()
            lld, phd = bc.timed_loglik(md, xd, 5)
            legd[f"trials_{nt}"] = {"wall_ms": phd["wall_ms"], "riccati_ms": phd["riccati_ms"], "forward_ms": phd["forward_ms"],
                                    "trial_ms": phd["trial_ms"], "path": phd["path"]}
        # the same evaluation on the sequential (workgroup-per-system) sweeps, which the default rule leaves for few long systems
        prev = os.environ.get("LQG_SCAN")
        os.environ["LQG_SCAN"] = "0"
        try:
            _, phs = bc.timed_loglik(md, xd_all[:1].contiguous(), 5)
            legd["trials_1_sequential_sweeps"] = {"wall_ms": phs["wall_ms"], "riccati_ms": phs["riccati_ms"],
                                                  "forward_ms": phs["forward_ms"], "trial_ms": phs["trial_ms"], "path": phs["path"]}
        finally:
            if prev is None:
                os.environ.pop("LQG_SCAN", None)
            else:
                os.environ["LQG_SCAN"] = prev
        t0 = time.perf_counter()
        legd["max_rel_err_vs_fp64_oracle"] = bc.oracle_check(md, xd_all[:2].contiguous(),
                                                             md.log_likelihood(xd_all[:2].contiguous()), n_samples=2)
        # (the same evaluation by the literal dense C port on ONE host core, two trials — context, not a target)
        legd["cpu_port_ms_two_trials_incl_check"] = (time.perf_counter() - t0) * 1e3
        # value + gradient of the same model: central differences over its 6 parameters = 13 systems, each its own
        # workgroup, evaluated concurrently (one launch set) — the gradient route for shapes without adjoint lane kernels
        try:
            from lqg_amd.infer.models import get_model_params
            pd = {k: float(v) for k, v in get_model_params(DelayedSubjectiveActor).items()
                  if k in ("c", "action_variability", "subj_noise", "subj_vel_noise", "sigma_target", "sigma_cursor")}
            xg2 = torch.cat([xd_all[:50], xd_all[:50, -1:]], dim=1).double()
            for i in range(2):
                gradient.value_and_grad(xg2, DelayedSubjectiveActor, pd, method="fd")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5):
                v_, g_ = gradient.value_and_grad(xg2, DelayedSubjectiveActor, dict(pd, sigma_target=pd["sigma_target"] + 1e-3 * i), method="fd")
            torch.cuda.synchronize()
            legd["value_and_grad_fd"] = {"ms": (time.perf_counter() - t0) / 5 * 1e3, "parameters": len(pd), "systems": 2 * len(pd) + 1,
                                         "trials": 50, "dtype": "f64", "finite": bool(all(v == v for v in g_.values()))}
        except Exception as e:
            legd["value_and_grad_fd"] = {"error": repr(e)[:300]}
        extra["delay12"] = legd
    except Exception as e:
        extra["delay12"] = {"error": repr(e)}
    return extra


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))               # before anything touches the GPU

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0                             # every rank computes on device 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torchrun the collective path is exercised even at world == 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:                     # one GPU for all ranks: RCCL needs a device per rank, gloo does not
            dist.init_process_group("gloo")
            dist = HostReduce(torch, dist)
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.config == 3:
        out = config3(torch, dist, args, dev, rank, world)
    else:
        out = headline_leg(torch, dist, args, dev, rank, world, args.dtype, args.log2_batch, args.steps, args.warmup,
                           cpu=(world == 1 and not args.no_cpu_baseline))
        if world == 1 and not args.no_extra and out is not None:
            try:
                out["extra"] = extra_legs(torch, args, dev)
            except Exception as e:          # the secondary legs must never cost the headline line
                out["extra"] = {"error": repr(e)}
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
