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

import bench_line           # noqa: E402  (the contract line: compaction + the side file; no torch, no GPU)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 matrix (MFMA f32) peak
PEAK_FP64_TFLOPS = 78.6    # datasheet FP64 vector == FP64 matrix peak
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COPY_HBM_GBS = 6290.0      # MI355X_MICROARCH.md: measured device copy rate
VALU_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 2    # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")      # scripts/pmc_legs.sh + scripts/pmc_records.py


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
    ap.add_argument("--config", type=int, default=5, choices=[3, 4, 5],
                    help="5 = the headline (BASELINE metric); 3 = candidate search, trial axis split over the ranks; "
                         "4 = BASELINE config 4: one 2-D hand model (n=10, T=1000), 262144 / N trials per rank")
    ap.add_argument("--only", default=None,
                    help="run ONE secondary leg (a key of `extra`) instead of the headline — what the per-leg PMC passes "
                         "profile (scripts/pmc_legs.sh)")
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


def pmc_record(leg, kernel):
    """The PMC-derived HBM bytes / VALU instructions per launch of `kernel` (a family name: "forward", "riccati", "trial")
    inside bench leg `leg`, but only if the committed profile (profiles/r06_pmc.json, collected by scripts/pmc_legs.sh as
    separate --pmc passes of `bench.py --only <leg>`) was taken on exactly this build: same library source hash, same
    pattern-library header hash.  None otherwise — a leg then reports `traffic: null` rather than a stale figure."""
    try:
        from lqg_amd import build, specialize
        pj = json.load(open(PMC_FILE))
        for rec in pj.get("records", []):
            if (rec.get("leg") == leg and rec.get("kernel") == kernel and rec.get("source_hash") == build.source_hash()
                    and rec.get("sp_headers_hash") == specialize._headers_hash()
                    and rec.get("adj_headers_hash", specialize._adj_headers_hash()) == specialize._adj_headers_hash()):
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


def allreduce_us(torch, dist, n, reps=200):
    """Latency of the path's one collective — all-reduce of n fp64 values (RCCL over xGMI) — each call timed on its own:
    {"mean", "p50", "p90", "p99", "max", "back_to_back_mean"} in microseconds (None without a process group)."""
    if dist is None:
        return None
    import numpy as np
    v = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(10):
        dist.all_reduce(v)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        dist.all_reduce(v)
        b.record()
    torch.cuda.synchronize()
    us = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.all_reduce(v)
    e1.record()
    torch.cuda.synchronize()
    return {"elements_fp64": int(n), "calls": int(reps), "mean": float(us.mean()), "p50": float(np.percentile(us, 50)),
            "p90": float(np.percentile(us, 90)), "p99": float(np.percentile(us, 99)), "max": float(us.max()),
            "back_to_back_mean": e0.elapsed_time(e1) / reps * 1e3}


def collective_info(torch, dist, args, world):
    """What the N > 1 line says about its process group, so that the first run on a real multi-GPU node explains itself:
    backend, ranks the backend reports after init (RCCL: `rccl_ranks_seen`), devices."""
    if dist is None:
        return {"backend": None, "world_size": world, "rccl_ranks_seen": None}
    backend = "gloo (host tensors, --share-gpu)" if args.share_gpu else "nccl (RCCL)"
    seen = dist.get_world_size()
    return {"backend": backend, "world_size": int(seen), "rccl_ranks_seen": None if args.share_gpu else int(seen),
            "devices_visible": int(torch.cuda.device_count()), "device": torch.cuda.get_device_name(torch.cuda.current_device())}


def gather_partials(torch, dist, local):
    """Every rank's partial objective (the fp64 scalar it fed the all-reduce), in rank order, on every rank."""
    local = local.reshape(1).double()
    if dist is None:
        return [float(local.item())]
    allo = [torch.zeros_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(allo, local)
    return [float(v.item()) for v in allo]


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


def usable_cpus():
    """(threads worth starting, what limits them): the scheduler affinity of this process and the cgroup CPU quota of the
    container — on a pod with a quota of a few CPUs, 256 OpenMP threads time-slice a handful of cores (round 3 reported
    `cores: 256` for what were ~6 cores' worth of throughput: the one-thread figure gave it away)."""
    n = os.cpu_count() or 1
    why = {"os_cpu_count": n}
    try:
        aff = len(os.sched_getaffinity(0))
        why["sched_affinity"] = aff
        n = min(n, aff)
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                why["cgroup_cpu_quota"] = float(quota) / period
                n = min(n, q)
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), why


def cpu_baseline(torch, np, OC, workload, lqg_amd, system, x_ref, dev, dtype_name, T, B, args):
    """The CPU baseline BASELINE.md 2 planned: the literal dense C restatement (oracle/lqg_oracle.c — a LITERAL PORT of the
    reference's formulas with run-time dims: LU with pivoting, Jacobi eigenvalues, no structure, one heap block per system;
    it is the parity checker, not a tuned CPU solver) timed on the GPU box's host cores on bounded samples of the same
    workload: all cores and ONE thread, in the bench dtype and in the other one.  A stated baseline, not the target."""
    host_cpus = os.cpu_count() or 1
    ncpu, cpu_limit = usable_cpus()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def timed(np_dt, nsamp, threads, budget_s):
        OC.lib().lqg_oracle_set_threads(threads)
        sel = np.arange(nsamp) % B
        a_s, d_s = host_specs(torch, np, workload, lqg_amd.LQGSpec, system, sel, np_dt)
        x_s = x_ref[torch.as_tensor(sel, device=dev)].cpu().numpy().astype(np_dt)
        k = max(1, min(8, nsamp))
        OC.log_likelihood({f: v[:k] for f, v in a_s.items()}, {f: v[:k] for f, v in d_s.items()}, x_s[:k], dtype=np_dt)
        tc = time.perf_counter()
        OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
        t1 = time.perf_counter() - tc
        nrep = 1
        if t1 < 0.5 * budget_s:
            nrep = int(min(16, max(1, budget_s / max(t1, 1e-3))))
            tc = time.perf_counter()
            for _ in range(nrep):
                OC.log_likelihood(a_s, d_s, x_s, dtype=np_dt)
            t1 = (time.perf_counter() - tc) / nrep
        return nsamp / t1, nrep

    np_main = np.float32 if dtype_name == "f32" else np.float64
    np_other, other = (np.float64, "f64") if dtype_name == "f32" else (np.float32, "f32")
    nsamp = args.cpu_sample or 16384                  # 64 solves per thread on a 256-thread host
    v_all, nrep = timed(np_main, nsamp, ncpu, 8.0)
    threads = OC.lib().lqg_oracle_max_threads()
    n1 = max(16, min(512, int(v_all / max(threads, 1) * 3.0)))       # ~3 s of one thread
    v_one, _ = timed(np_main, n1, 1, 3.0)
    v_all_o, _ = timed(np_other, nsamp, ncpu, 4.0)
    v_one_o, _ = timed(np_other, n1, 1, 2.0)
    OC.lib().lqg_oracle_set_threads(ncpu)
    return dict(
        value=v_all, unit="solves/s", cores=threads, kind="port",
        port="literal dense restatement (oracle/lqg_oracle.c: run-time dims, no structure, one heap block per system, "
             "-O3 -march=x86-64-v3, OpenMP over systems) — the parity checker, not a tuned CPU solver",
        sample=f"{nsamp} solves of the same workload ({dtype_name}, T={T}) x {nrep} repetitions on all cores; {n1} solves on "
               f"one thread; the same two samples in {other}",
        single_thread={"value": v_one, "unit": "solves/s", "cores": 1, "dtype": dtype_name, "sample_solves": n1},
        other_dtype={"dtype": other, "value": v_all_o, "cores": threads,
                     "single_thread": {"value": v_one_o, "cores": 1, "sample_solves": n1}},
        cpu_model=model, host_cpu_count=host_cpus, cpu_limit=cpu_limit)


def headline_leg(torch, dist, args, dev, rank, world, dtype_name, log2_batch, steps, warmup, env=None, cpu=False, leg=None,
                 layout=None):
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
        layout = layout or args.layout
        pack_ms = None
        if layout == "packed":
            torch.cuda.synchronize()
            tp = time.perf_counter()
            x = workload.pack_trials(x_ref)                # one-time re-layout [B,1,T+1,d] -> storage [T+1][d][B] (outside the timed region)
            torch.cuda.synchronize()
            pack_ms = (time.perf_counter() - tp) * 1e3
        else:
            x = x_ref
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
    per_rank_obj = gather_partials(torch, dist, _hip.sum_trials(ll.view(1, B)))   # each rank's partial objective (what it fed the all-reduce)
    if rank != 0:
        return None

    fl = algorithmic_flops_per_step(**dm)
    bytes_solve = algorithmic_bytes_per_solve(T=T, w=w, **dm)
    fwd_avg_ms, ric_avg_ms = float(np.mean(fwd_ms)), float(np.mean(ric_ms))
    n_launch = len(plan.work)                       # forward-kernel launches per step
    alg_gbs = bytes_solve * B / (fwd_avg_ms * 1e-3) / 1e9     # algorithmic bytes of one step / forward-kernel time
    rec = pmc_record(leg or f"headline_{dtype_name}", "forward") if (leg or (log2_batch == 20 and not env)) else None
    traffic = rec["hbm_bytes_per_launch"] if rec else None
    limits = {"note": "which side binds the dominant kernel: measured HBM bytes/s against the guide's achievable copy rate "
                      "vs wave64 VALU instructions issued/s against 1024 SIMDs x 2.4 GHz / 2 cycles"}
    if rec:
        limits["hbm_frac_of_copy_rate"] = traffic / (fwd_avg_ms * 1e-3) / 1e9 / COPY_HBM_GBS
        limits["hbm_frac_of_peak"] = traffic / (fwd_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        if rec.get("valu_wave_insts_per_launch"):
            limits["valu_issue_frac"] = rec["valu_wave_insts_per_launch"] / (fwd_avg_ms * 1e-3) / VALU_WAVE_INSTS_PER_S
            limits["valu_insts_per_step_per_wave"] = rec.get("valu_insts_per_step_per_wave")
            if rec.get("sustained_mhz"):
                # the chip does not hold 2.4 GHz under this load (GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the PMC pass): at
                # the clock it sustains the same instruction stream is this fraction of what the SIMDs can issue — the kernel is
                # within ~1.4x of its VALU floor for this formulation, not waiting on HBM
                limits["sustained_mhz"] = rec["sustained_mhz"]
                limits["valu_issue_frac_at_sustained_clock"] = (rec["valu_wave_insts_per_launch"] / (fwd_avg_ms * 1e-3)
                                                                / (1024 * rec["sustained_mhz"] * 1e6 / 2))
                limits["bound"] = "VALU issue (hbm is the contract's word for the algorithmic-bytes view, not the kernel's limiter)"
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
                   "solves_per_gpu": B, "T": T, "trajectory_layout": layout, "pack_trials_ms_one_time": pack_ms,
                   "path": plan.description,
                   "parallelism": f"candidate-sharded x{world}, all-reduce of the summed log-likelihood"},
        "world_size": world if dist is None else dist.get_world_size(),
        "per_rank_solves_per_s": [float(B) * steps / t for t in per_rank],
        "allreduce_us": ar_us["mean"] if ar_us else None, "allreduce_us_percentiles": ar_us,
        "collective": collective_info(torch, dist, args, world), "share_gpu": bool(args.share_gpu),
        # Contract form: achieved = ALGORITHMIC bytes per launch (mode M1: trajectory in, specs in, one
        # scalar out) / the dominant kernel's HIP-event time; traffic = PMC bytes of that launch (null unless the
        # committed profile was taken on exactly this build).
        "roofline": {"bound": "hbm", "achieved": alg_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": alg_gbs / PEAK_HBM_GBS, "traffic": traffic,
                     # the same algorithmic bytes over the WHOLE step (Riccati sweep + forward sweep + objective reduction)
                     "whole_step_frac": bytes_solve * B / (elapsed / steps) / 1e9 / PEAK_HBM_GBS,
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
            out["cpu_baseline"] = cpu_baseline(torch, np, OC, workload, lqg_amd, system, x_ref, dev, dtype_name, T, B, args)
    except Exception as e:  # the oracle is a checker: its absence must not break the measurement
        out["parity"] = dict(error=repr(e))
    return out


def config3_roofline(ph, Bc, n_trials, T, w):
    """Dominant kernel of config 3: k_trial_sp, the per-trial mean recursion + density over the operator stream.  Its data
    (x: 8.7 MB) stays in L2 and its operators come through the scalar cache: VALU-issue-bound.  achieved = wave64 VALU
    instructions issued per second from the committed PMC pass of THIS build (SQ_INSTS_VALU per launch / kernel time), peak =
    1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; traffic = the PMC HBM bytes of one launch.  Without a record
    carrying the running build's hashes: achieved / frac / traffic are null (never a stale number)."""
    rec = pmc_record("config3", "trial")
    ms = ph[2]
    steps_per_s = float(Bc) * n_trials * T / (ms * 1e-3)
    out = {"bound": "valu", "kernel": "k_trial_sp (per-trial mean recursion + density over the operator stream)",
           "kernel_ms": ms, "system_sweeps_ms": ph[0] + ph[1], "unit": "wave64 VALU instructions/s",
           "peak": VALU_WAVE_INSTS_PER_S, "achieved": None, "frac": None, "traffic": None,
           "trial_steps_per_s": steps_per_s,
           "data_rows_GBps_through_L2": float(Bc) * n_trials * (T + 1) * 2 * w / (ms * 1e-3) / 1e9,
           "algorithmic_bytes_per_launch": n_trials * (T + 1) * 2 * w + Bc * n_trials * w}
    if rec and rec.get("valu_wave_insts_per_launch"):
        out["achieved"] = rec["valu_wave_insts_per_launch"] / (ms * 1e-3)
        out["frac"] = out["achieved"] / VALU_WAVE_INSTS_PER_S
        out["valu_insts_per_trial_step"] = rec["valu_wave_insts_per_launch"] * 64.0 / (float(Bc) * n_trials * T)
        out["traffic"] = rec.get("hbm_bytes_per_launch")
        out["profile"] = rec.get("profile")
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
    per_rank_obj = gather_partials(torch, dist, _hip.sum_trials(plan.run()).sum())   # this rank's share, summed over candidates
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
        "per_rank_s": per_rank, "allreduce_us": ar_us["mean"] if ar_us else None, "allreduce_us_percentiles": ar_us,
        "collective": collective_info(torch, dist, args, world), "share_gpu": bool(args.share_gpu),
        "objective_checksum": float(obj.sum()), "per_rank_objective": per_rank_obj,
        "phase_ms": {"riccati": ph[0], "forward": ph[1], "trial": ph[2]},
        "roofline": config3_roofline(ph, Bc, Nt // world, T, w),
        "best_candidate": int(obj.argmax()), "objective_max": float(obj.max()),
    }


def config4(torch, dist, args, dev, rank, world, n_total=262144, steps=None, warmup=None):
    """BASELINE config 4, literally: the 2-D hand model (notebooks/HandModel.ipynb: x = b = 10, u = 2, y = 4, d = 4, T = 1000),
    ONE parameter vector, 262 144 trials sharded over the ranks (262144 / N per rank, contiguous blocks); every
    rank solves the one system itself (a few kB of specs), one all-reduce of the scalar fp64 objective per step.  Strong
    scaling.  The synthetic data set is defined in 8 blocks of n_total / 8 trials (block k simulated from seed 1400 + k);
    rank r of N (N | 8) owns the blocks [8 r / N, 8 (r + 1) / N): the SAME data set at every N, so the all-reduced objective of
    an N-rank run equals the 1-rank run's to fp64 rounding (tests/test_gpu_fullsize.py)."""
    import numpy as np
    import bench_configs as bc
    from lqg_amd import _hip, workload
    from lqg_amd.plan import LogLikelihoodPlan
    steps, warmup = steps or args.steps, args.warmup if warmup is None else warmup
    if n_total % world:
        raise SystemExit(f"--config 4: {n_total} trials do not split over {world} ranks")
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    T, n_loc = 1000, n_total // world
    m = bc.hand2d_system(T, dev, dtype)
    if 8 % world == 0 and n_total % 8 == 0:
        per = 8 // world
        x = torch.cat([m.simulate(1400 + rank * per + k, n=n_total // 8)[..., :4] for k in range(per)], dim=0).contiguous()
    else:                                            # N does not divide 8: rank-seeded shards (not comparable across N)
        x = m.simulate(1400 + rank, n=n_loc)[..., :4].contiguous()
    x = workload.pack_trials(x)
    plan = LogLikelihoodPlan(m, x, events=True)

    def step():
        obj = _hip.sum_trials(plan.run())            # scalar fp64 partial objective of this rank's trials
        if dist is not None:
            obj = obj.reshape(1)
            dist.all_reduce(obj)
        return obj

    obj, elapsed, per_rank, step_ms = timed_steps(torch, dist, step, steps, warmup)
    ph = plan.phase_ms()
    ar_us = allreduce_us(torch, dist, 1)
    ll = plan.run()
    per_rank_obj = gather_partials(torch, dist, _hip.sum_trials(ll).sum())
    parity = None
    if rank == 0:
        try:
            parity = bc.oracle_check(m, x, ll, n_samples=3)
        except Exception as e:
            parity = repr(e)
    if rank != 0:
        return None
    w = 4 if args.dtype == "f32" else 8
    b4 = n_loc * (T + 1) * 4 * w + n_loc * w          # this rank's trajectories in, one scalar per trial out
    return {
        "metric": "trial-evals/sec (BASELINE config 4: 2-D hand model n=10, T=1000, 262144 trials over N GPUs)",
        "value": float(n_total) * steps / elapsed, "unit": "trial-evals/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "ms_per_step_min": float(np.min(step_ms)),
        "ms_per_step_median": float(np.median(step_ms)), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE config 4: hand model 2-D (x=b=10 u=2 y=4 d=4), T={T}, one system x {n_total} trials",
                   "trials": n_total, "trials_per_rank": n_loc, "T": T, "path": plan.description,
                   "parallelism": f"trial-split x{world}, all-reduce of the scalar fp64 objective"},
        "world_size": world if dist is None else dist.get_world_size(), "per_rank_s": per_rank,
        "allreduce_us": ar_us["mean"] if ar_us else None, "allreduce_us_percentiles": ar_us,
        "collective": collective_info(torch, dist, args, world), "share_gpu": bool(args.share_gpu),
        "objective_sum": float(obj.sum()), "per_rank_objective": per_rank_obj,
        "phase_ms": {"riccati": ph[0], "forward": ph[1], "trial": ph[2]},
        "max_rel_err_vs_fp64_oracle": parity,
        "roofline": {"bound": "hbm", "kernel": f"per-trial sweep over {n_loc} trajectories per rank (x streamed once)",
                     "kernel_ms": ph[2], "system_sweeps_ms": ph[0] + ph[1], "achieved": b4 / (ph[2] * 1e-3) / 1e9,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": b4 / (ph[2] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     "traffic": ((pmc_record("config4_sharded", "trial") or {}).get("hbm_bytes_per_launch") if world == 1 else None),
                     "algorithmic_bytes_per_launch": b4},
    }


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def leg_headline_other(torch, args, dev):
    other = "f64" if args.dtype == "f32" else "f32"
    leg = headline_leg(torch, None, args, dev, 0, 1, other, args.log2_batch, min(args.steps, 10), 2)
    out = {k: leg[k] for k in ("value", "unit", "ms_per_step", "dtype", "parity", "roofline")}
    out["path"] = leg["config"]["path"]
    return out


def leg_reference_layout(torch, args, dev):
    """The headline with the trajectories in the REFERENCE's layout x[B, 1, T+1, d] (what a drop-in caller hands over) instead
    of the packed [T+1][d][B] storage the default line is timed on, plus the one-time cost of the re-layout."""
    leg = headline_leg(torch, None, args, dev, 0, 1, args.dtype, args.log2_batch, min(args.steps, 10), 2, layout="reference")
    packed = headline_leg(torch, None, args, dev, 0, 1, args.dtype, args.log2_batch, 3, 1, layout="packed")
    return {"value": leg["value"], "unit": leg["unit"], "ms_per_step": leg["ms_per_step"], "dtype": args.dtype,
            "trajectory_layout": "reference [B][1][T+1][d] (each lane reads its own 8 kB trajectory: 64 cache lines per wave-load)",
            "kernel_ms": leg["roofline"]["kernel_ms"], "frac_hbm_algorithmic": leg["roofline"]["frac"],
            "pack_trials_ms_one_time": packed["config"]["pack_trials_ms_one_time"],
            "note": "pack_trials is paid once per data set (it is reused by every candidate / optimiser step); "
                    "ms_per_step here is the sweep on the unpacked data",
            "parity": leg.get("parity")}


def leg_dense_generic(dn):
    def run(torch, args, dev):
        leg = headline_leg(torch, None, args, dev, 0, 1, dn, 17, min(args.steps, 5), 1,
                           env={"LQG_NO_SPECIALIZE": "1", "LQG_NO_DECOUPLE": "1"})
        r = leg["roofline"]
        # dense joint n=6 (m=10) kernel: VALU-bound; the algorithmic flop rate is SURVEY 8(d)'s reference formulation
        fl_rate = r["algorithmic_flops_per_solve"] * leg["value"] / 1e12
        peak = PEAK_FP32_TFLOPS if dn == "f32" else PEAK_FP64_TFLOPS
        # achieved = wave64 VALU instructions actually ISSUED per second (PMC SQ_INSTS_VALU of this build's forward kernel) against
        # what the SIMDs can issue; the reference formulation's nominal flop rate (no symmetry, no hoisting) is kept beside it as
        # context only — it is not an achieved-FLOP figure
        rec = pmc_record("dense_generic_" + dn, "forward")
        rate = 2.0 if dn == "f32" else 4.0            # cycles per wave64 VALU instruction (fp64: half rate)
        ach = rec["valu_wave_insts_per_launch"] / (r["kernel_ms"] * 1e-3) if rec and rec.get("valu_wave_insts_per_launch") else None
        return {"value": leg["value"], "unit": leg["unit"], "ms_per_step": leg["ms_per_step"], "dtype": dn,
                "solves_per_gpu": 1 << 17, "path": leg["config"]["path"], "parity": leg.get("parity"),
                "env": "LQG_NO_SPECIALIZE=1 LQG_NO_DECOUPLE=1",
                "roofline": {"bound": "valu", "achieved": ach, "unit": "wave64 VALU instructions/s (issued, PMC)",
                             "peak": 1024 * 2.4e9 / rate, "frac": (ach / (1024 * 2.4e9 / rate)) if ach else None,
                             "kernel_ms": r["kernel_ms"], "riccati_kernel_ms": r["riccati_kernel_ms"],
                             "traffic": rec.get("hbm_bytes_per_launch") if rec else None,
                             "profile": rec.get("profile") if rec else None,
                             "nominal_reference_formulation_tflops": fl_rate, "nominal_frac_of_vector_peak": fl_rate / peak}}
    return run


def leg_specialised_joint(torch, args, dev):
    """What a user with a genuinely COUPLED n = 6 model gets: the headline workload with block decoupling switched off
    (LQG_NO_DECOUPLE=1) — the joint (x=4, b=6, u=2, y=4) problem on its pattern library (structural zeros compiled out, nothing
    split, nothing merged).  Between the decoupled headline and the dense generic floor."""
    leg = headline_leg(torch, None, args, dev, 0, 1, "f32", 18, min(args.steps, 10), 2, env={"LQG_NO_DECOUPLE": "1"},
                       leg="specialised_joint_n6")
    out = {k: leg[k] for k in ("value", "unit", "ms_per_step", "dtype", "parity", "roofline")}
    out.update(solves_per_gpu=1 << 18, path=leg["config"]["path"], env="LQG_NO_DECOUPLE=1")
    return out


def leg_one_system(cfg):
    def run(torch, args, dev):
        # one system x many trials (BASELINE configs 2 and 4, the inner loop of MLE / NUTS): the time-parallel path (scans over
        # the time axis + time-chunked per-trial sweep) against the sequential lane kernels, wall ms per evaluation
        import bench_configs as bc
        import lqg_amd
        from lqg_amd import workload
        if cfg == 2:
            m = lqg_amd.PointMassBoundedActor(T=500, action_variability=0.5, device=dev, dtype=torch.float32)
            x = workload.pack_trials(m.simulate(12, n=65536)[..., :2].contiguous())
        else:
            m = bc.hand2d_system(1000, dev, torch.float32)
            x = workload.pack_trials(m.simulate(14, n=32768)[..., :4].contiguous())
        leg = {"systems": 1, "trials": int(x.shape[-3]), "T": int(m.T), "dtype": "f32"}
        for name, env in (("time_parallel", {}), ("sequential", {"LQG_SCAN": "0", "LQG_TRIAL_CHUNKS": "0"})):
            ll, ph = _with_env(env, lambda: bc.timed_loglik(m, x, 10))
            leg[name] = {"wall_ms": ph["wall_ms"], "system_sweeps_ms": ph["riccati_ms"] + ph["forward_ms"],
                         "per_trial_sweep_ms": ph["trial_ms"], "path": ph["path"],
                         "max_rel_err_vs_fp64_oracle": bc.oracle_check(m, x, ll, n_samples=4)}
        leg["speedup"] = leg["sequential"]["wall_ms"] / leg["time_parallel"]["wall_ms"]
        leg["value"], leg["unit"] = leg["trials"] / (leg["time_parallel"]["wall_ms"] * 1e-3), "trial-evals/s"
        return leg
    return run


def leg_one_vector(torch, args, dev):
    # the inner loop of the reference's inference drivers (lqg/infer/mle.py:17-23, NUTS): value + gradient of ONE parameter
    # vector on 50 trials, T = 500, fp64 — batched central differences replayed as one hipGraph (lqg_amd/infer/graphed.py),
    # the same launched from Python, and the reverse-mode sweep; BASELINE.md derives ~54 evaluations/s for the reference
    import lqg_amd
    from lqg_amd.infer import gradient
    true = dict(sigma_target=25.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5)
    with torch.no_grad():
        xg = lqg_amd.BoundedActor(T=500, device=dev, dtype=torch.float64, **true).simulate(0, n=50)
    xg = torch.cat([xg, xg[:, -1:]], dim=1)                       # lqg_model's convention: T rows = T - 1 steps
    p0 = dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4)
    leg = {"workload": "BoundedActor T=500, 50 trials, 4 parameters, one parameter vector, fp64", "unit": "ms per value+gradient"}
    for name, method, env, reps in (("fd_graph", "fd", {}, 300), ("fd_eager", "fd", {"LQG_GRAPH": "0"}, 100), ("adjoint", "adjoint", {}, 50)):
        def go():
            for i in range(5):
                gradient.value_and_grad(xg, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 0.01 * i), method=method)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):          # (a different vector every call: nothing is cached across evaluations)
                gradient.value_and_grad(xg, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 1e-3 * i), method=method)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        leg[name] = _with_env(env, go)
    leg["evaluations_per_s"] = 1e3 / leg["fd_graph"]
    return leg


def leg_m2(torch, args, dev):
    # mode M2 of SURVEY 8(d): genuinely time-varying specs in, L, H, K, mu, Sigma out, [T][element][system] storage
    import bench_m2
    out = bench_m2.run(dev, "f32", 17, 500, 3)
    rec_r, rec_f = pmc_record("m2_f32", "riccati"), pmc_record("m2_f32", "forward")
    if rec_r and rec_f:
        out["roofline"]["traffic"] = rec_r["hbm_bytes_per_launch"] + rec_f["hbm_bytes_per_launch"]
        out["roofline"]["traffic_kernels"] = {"k_riccati_tv_sp": rec_r["hbm_bytes_per_launch"],
                                              "k_forward_tv_sp": rec_f["hbm_bytes_per_launch"]}
        out["roofline"]["profile"] = rec_f.get("profile")
    return out


def leg_timevarying(dn):
    def run(torch, args, dev):
        """`System.log_likelihood` on genuinely time-varying specs — the reference's literal data model ((T, ...)-stacked arrays,
        lqg/spec.py:5-19) with every non-zero entry of every matrix moving in time and over the systems, [T][element][system]
        storage — served by the pattern libraries since round 6 (k_riccati_tv_sp -> k_forward_tv_sp with nothing materialised).
        Two workloads: `costs_stay_psd` (Q_t, R_t moved by a congruence, as a model whose PARAMETERS move in time produces them:
        the two 1-D components decouple) is the leg's value; `per_entry_jitter` (mode M2's workload: Q_t loses positive
        semi-definiteness, the decoupling is rightly refused, the joint n = 10 problem is solved) beside it.  Roofline: the
        structurally non-zero spec entries + the trajectory, read once."""
        import numpy as np
        import bench_m2
        import bench_configs as bc
        from lqg_amd import workload
        from lqg_amd.plan import LogLikelihoodPlan
        dtype = torch.float32 if dn == "f32" else torch.float64
        w = 4 if dn == "f32" else 8
        # (fp64 at 2^17 systems: 124 GB of [T][element][system] spec stacks + as much again for the decoupled components' own
        # stacks + the fp64 temporaries of the positive-semi-definiteness certificate pass 288 GB; 2^16 = one wave per SIMD)
        B, T = (1 << 17) if dn == "f32" else (1 << 16), 500
        if os.environ.get("LQG_BENCH_TV_LOG2B"):
            B = 1 << int(os.environ["LQG_BENCH_TV_LOG2B"])
        out = {"dtype": dn, "systems": B, "T": T, "unit": "solves/s"}
        for name, psd in (("costs_stay_psd", True), ("per_entry_jitter", False)):
            system, base = bench_m2.m2_system(dev, dtype, B, T, psd=psd)
            x = workload.pack_trials(workload.simulate_one_trial_each(base, seed=5))
            del base
            conc = os.environ.get("LQG_BENCH_TV_CONCURRENT")
            plan = LogLikelihoodPlan(system, x, events=True, concurrent=(None if conc is None else bool(int(conc))))
            for _ in range(2):
                ll = plan.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms = []
            for _ in range(5):
                e0.record()
                ll = plan.run()
                e1.record()
                e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            ms = float(np.median(ms))
            ph = plan.phase_ms()
            # bytes the sweeps must read: the non-zero entries of every time-varying field per step + the trajectory rows
            from lqg_amd import _hip
            nz = 0
            for comp, cols, _ in (system.decoupled(4) or [(system, [0, 1, 2, 3], None)]):
                mk = _hip._time_varying_pattern(None, comp, len(cols))[1]       # (cached on the component by the plan)
                nz += sum(int(mk[k].sum()) for k in ("Aa", "Ba", "Fa", "Va", "Wa", "Ad", "Bd", "Fd", "Vd", "Wd"))
                nz += sum(int(np.triu(mk[k]).sum()) for k in ("Q", "Rr"))        # symmetric: the upper triangle is loaded
            alg = B * (T * nz + (T + 1) * 4 + 1) * w
            leg = {"ms": ms, "solves_per_s": B / (ms * 1e-3), "components": len(plan.work), "path": plan.description,
                   "specialised": [bool(wk["specialised"]) for wk in plan.work], "phase_ms": {"riccati": ph[0], "forward": ph[1]},
                   "nonzero_spec_entries_per_step": nz, "algorithmic_bytes_per_launch_set": alg,
                   "hbm_frac_algorithmic": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                   "max_rel_err_vs_fp64_oracle": bc.oracle_check(system, x, ll, n_samples=6)}
            out[name] = leg
            del plan, system, x, ll
            torch.cuda.empty_cache()
        out["value"] = out["costs_stay_psd"]["solves_per_s"]
        # (the PMC pass sees both workloads' launches of the same kernels: the family figure is their per-launch average)
        rf, rr = pmc_record(f"timevarying_{dn}", "forward"), pmc_record(f"timevarying_{dn}", "riccati")
        out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                           "achieved": out["costs_stay_psd"]["hbm_frac_algorithmic"] * PEAK_HBM_GBS,
                           "frac": out["costs_stay_psd"]["hbm_frac_algorithmic"],
                           "traffic": (rf["hbm_bytes_per_launch"] + rr["hbm_bytes_per_launch"]) if (rf and rr) else None,
                           "traffic_note": "per launch of ONE component (k_riccati_tv_sp + k_forward_tv_sp); the leg launches two",
                           "kernel": "k_riccati_tv_sp + k_forward_tv_sp over both decoupled components (spec entries streamed once per step)"}
        return out
    return run


def leg_config3(torch, args, dev):
    # BASELINE config 3 at its literal shape (4096 candidates x 1024 trials, T = 1067): trial-evals/s, k_trial_sp time
    class A3:
        dtype, steps, warmup, share_gpu = "f32", 10, 2, False
    c3 = config3(torch, None, A3, dev, 0, 1)
    return {k: c3[k] for k in ("metric", "value", "unit", "ms_per_step", "phase_ms", "roofline", "config", "best_candidate")}


def leg_config4(torch, args, dev):
    class A4:
        dtype, steps, warmup, share_gpu = "f32", 10, 2, False
    c4 = config4(torch, None, A4, dev, 0, 1)
    return {k: c4[k] for k in ("metric", "value", "unit", "ms_per_step", "phase_ms", "roofline", "config",
                               "max_rel_err_vs_fp64_oracle")}


def leg_config5_one_system(torch, args, dev):
    # BASELINE config 5 in its literal form: ONE system x 2^20 trials (the per-trial sweep streams x once: HBM-bound)
    import bench_configs as bc
    import lqg_amd
    from lqg_amd import workload
    m5 = lqg_amd.SubjectiveActor(dim=2, T=500, device=dev, dtype=torch.float32)
    x5 = workload.pack_trials(m5.simulate(15, n=1 << 20))
    ll5, ph5 = bc.timed_loglik(m5, x5, 10)
    b5 = (1 << 20) * 501 * 4 * 4 + (1 << 20) * 4
    rec = pmc_record("config5_one_system", "trial")
    return {"systems": 1, "trials": 1 << 20, "T": 500, "dtype": "f32", "wall_ms": ph5["wall_ms"], "path": ph5["path"],
            "value": (1 << 20) / (ph5["wall_ms"] * 1e-3), "unit": "trial-evals/s",
            "roofline": {"bound": "hbm", "kernel": "per-trial sweep (k_trial_sp) over 2^20 trajectories", "kernel_ms": ph5["trial_ms"],
                         "system_sweeps_ms": ph5["riccati_ms"] + ph5["forward_ms"],
                         "achieved": b5 / (ph5["trial_ms"] * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": b5 / (ph5["trial_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "traffic": rec["hbm_bytes_per_launch"] if rec else None, "profile": rec.get("profile") if rec else None,
                         "algorithmic_bytes_per_launch": b5},
            "max_rel_err_vs_fp64_oracle": bc.oracle_check(m5, x5, ll5, n_samples=4)}


def leg_delay12(torch, args, dev):
    # the reference's largest model: DelayedSubjectiveActor (delay 12: x = 26, b = 39, m = 65): time-parallel system sweeps on
    # windows of 39 / 63 (k_scan_level_rt) + the row-parallel per-trial sweep
    import bench_configs as bc
    from lqg_amd.infer import gradient
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
    _, phs = _with_env({"LQG_SCAN": "0"}, lambda: bc.timed_loglik(md, xd_all[:1].contiguous(), 5))
    legd["trials_1_sequential_sweeps"] = {"wall_ms": phs["wall_ms"], "riccati_ms": phs["riccati_ms"],
                                          "forward_ms": phs["forward_ms"], "trial_ms": phs["trial_ms"], "path": phs["path"]}
    t0 = time.perf_counter()
    legd["max_rel_err_vs_fp64_oracle"] = bc.oracle_check(md, xd_all[:2].contiguous(),
                                                         md.log_likelihood(xd_all[:2].contiguous()), n_samples=2)
    # (the same evaluation by the literal dense C port on ONE host core, two trials — context, not a target)
    legd["cpu_port_ms_two_trials_incl_check"] = (time.perf_counter() - t0) * 1e3
    # value + gradient of the same model (6 parameters): the reverse-mode sweep when the library has it for this shape, and
    # central differences = 13 systems, each its own workgroup, evaluated concurrently (one launch set)
    from lqg_amd.infer.models import get_model_params
    pd = {k: float(v) for k, v in get_model_params(DelayedSubjectiveActor).items()
          if k in ("c", "action_variability", "subj_noise", "subj_vel_noise", "sigma_target", "sigma_cursor")}
    xg2 = torch.cat([xd_all[:50], xd_all[:50, -1:]], dim=1).double()
    for method, key in (("fd", "value_and_grad_fd"), ("adjoint", "value_and_grad_adjoint")):
        try:
            for i in range(2):
                gradient.value_and_grad(xg2, DelayedSubjectiveActor, pd, method=method)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5):
                v_, g_ = gradient.value_and_grad(xg2, DelayedSubjectiveActor, dict(pd, sigma_target=pd["sigma_target"] + 1e-3 * i),
                                                 method=method)
            torch.cuda.synchronize()
            legd[key] = {"ms": (time.perf_counter() - t0) / 5 * 1e3, "parameters": len(pd), "trials": 50, "dtype": "f64",
                         "systems": 2 * len(pd) + 1 if method == "fd" else 1, "finite": bool(all(v == v for v in g_.values()))}
        except Exception as e:
            legd[key] = {"error": repr(e)[:300]}
    legd["value"] = legd["trials_1"]["wall_ms"]        # the summary figure of the contract line (unit above: ms per evaluation)
    return legd


def leg_delay12_batch(torch, args, dev):
    """m = 65 as a BATCH of candidates x 120 trials (the shape cpp_data_fit.py fits, /root/reference cpp_data_fit.py:15-55:
    20 trials x 6 conditions): 64 and 4096 candidates, sequential cooperative sweeps (one workgroup per system) against the
    time-parallel sweeps, ms per objective sweep and the bound that binds."""
    import bench_configs as bc
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    T, nt = 500, 120
    truth = DelayedSubjectiveActor(T=T, device=dev, dtype=torch.float32)
    x = truth.simulate(23, n=nt)[..., :2].contiguous()
    out = {"model": "DelayedSubjectiveActor x=26 b=39 m=65, T=500", "trials": nt, "dtype": "f32", "unit": "ms per objective sweep"}
    for nc in (64, 4096):
        sig = torch.linspace(3.0, 40.0, nc, device=dev, dtype=torch.float32)
        m = DelayedSubjectiveActor(T=T, sigma_target=sig, device=dev, dtype=torch.float32)
        leg = {}
        for name, env in (("sequential_cooperative", {"LQG_SCAN": "0"}), ("time_parallel", {"LQG_SCAN": "1"})):
            if name == "time_parallel" and nc > 64:
                leg[name] = {"skipped": "a level of 4096 x 500 windows of 63 x 63 is 2e6 workgroups of 0.1 ms: the scan costs per "
                                        "system what the sequential sweep costs per CU-round"}
                continue
            try:
                ll, ph = _with_env(env, lambda: bc.timed_loglik(m, x, 2))
                leg[name] = {"wall_ms": ph["wall_ms"], "riccati_ms": ph["riccati_ms"], "forward_ms": ph["forward_ms"],
                             "trial_ms": ph["trial_ms"], "path": ph["path"],
                             "candidate_evals_per_s": nc / (ph["wall_ms"] * 1e-3)}
            except Exception as e:
                leg[name] = {"error": repr(e)[:300]}
        seq = leg.get("sequential_cooperative", {})
        if "wall_ms" in seq:
            # one workgroup per system, 256 CUs: ceil(nc / 256) rounds of T dependent steps
            rounds = -(-nc // 256)
            leg["bound"] = (f"latency of the T = {T} dependent steps of one workgroup per system x {rounds} CU-round(s): "
                            f"{seq['forward_ms'] / rounds / T * 1e3:.1f} us per forward step per round")
        out[f"candidates_{nc}"] = leg
        del m
        torch.cuda.empty_cache()
    best = [v["wall_ms"] for v in out["candidates_4096"].values() if isinstance(v, dict) and "wall_ms" in v]
    if best:
        out["value"] = min(best)                       # the summary figure: ms per objective sweep of 4096 candidates x 120 trials
    return out


def _grad_leg(torch, dev, system, x, B, n_trials, T, dims, leg_name, steps=5, warmup=2, fd_check=None):
    """Value + reverse-mode gradient of the summed log-likelihood — what every inference driver of the reference runs
    (`jit(grad(fun))` lqg/optim.py:142, NUTS lqg/infer/utils.py:14-18, SVI-Adam lqg/infer/mle.py:14-25) — through a persistent
    lqg_amd.grad.GradPlan: the sweep's kernels only (specs resident in HBM, as the headline's forward sweeps are timed)."""
    import numpy as np
    from lqg_amd import grad as G
    gp = G.GradPlan(system, x, events=True)
    for _ in range(warmup):
        gp.run()
    torch.cuda.synchronize()
    tm = Timer(torch, steps)
    ph = []
    t0 = time.perf_counter()
    for it in range(steps):
        tm.ev[it][0].record()
        ll, bars = gp.run()
        tm.ev[it][1].record()
        ph.append(gp.phase_ms())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms = float(np.median(tm.ms()))
    phm = {k: float(np.median([p[k] for p in ph])) for k in ph[0]}
    kernel_ms = sum(phm.values())
    x_, b_, u_, y_, d_ = dims
    w = 4
    n_out = sum(int(np.prod(v.shape[-2:])) for v in bars[0].values())
    # algorithmic bytes: the forward's (SURVEY 8d mode M1: specs + trajectories in, one scalar out) + the bars out
    alg = B * (algorithmic_bytes_per_solve(x_, b_, u_, y_, d_, T, w) + (n_trials - 1) * ((T + 1) * d_ + 1) * w * (1 if x.dim() == 4 else 0)
               + n_out * w * len(bars))
    if x.dim() == 3:                       # trials shared by all candidates: read once
        alg += n_trials * (T + 1) * d_ * w
    fin = bool(torch.isfinite(ll).all()) and all(bool(torch.isfinite(v).all()) for bd in bars for v in bd.values())
    out = {"value": B / (ms * 1e-3), "unit": "solves+gradient/s", "ms_per_value_and_grad": ms, "wall_ms_incl_host": wall,
           "candidates": B, "trials_per_candidate": n_trials, "T": T, "dtype": "f32", "path": gp.description,
           "kernel_ms": phm, "kernel_ms_total": kernel_ms, "all_finite": fin,
           "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": alg / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch_set": alg,
                        "note": "instruction-issue-bound sweeps: the matrix adjoints are O(m^3) register arithmetic per system-step, "
                                "the per-trial mu-bar recursion O(m^2) per trial-step (limits.valu_issue_frac below, from the PMC pass)"}}
    recs = {k: pmc_record(leg_name, k) for k in ("sys_fwd", "sys_rev", "ric_rev", "trial_fwd", "trial_rev", "riccati")}
    recs = {k: v for k, v in recs.items() if v}
    if recs:
        out["roofline"]["traffic"] = sum(v.get("hbm_bytes_per_launch") or 0.0 for v in recs.values())
        out["roofline"]["traffic_kernels"] = {k: v.get("hbm_bytes_per_launch") for k, v in recs.items()}
        insts = sum(v.get("valu_wave_insts_per_launch") or 0.0 for v in recs.values())
        out["roofline"]["limits"] = {"valu_wave_insts_per_launch_set": insts,
                                     "valu_issue_frac": insts / (kernel_ms * 1e-3) / VALU_WAVE_INSTS_PER_S,
                                     "valu_insts_per_kernel": {k: v.get("valu_wave_insts_per_launch") for k, v in recs.items()}}
        out["roofline"]["profile"] = next(iter(recs.values())).get("profile")
    if fd_check is not None:
        try:
            out["parity"] = fd_check(ll)
        except Exception as e:
            out["parity"] = {"error": repr(e)[:300]}
    return out


def leg_value_and_grad_headline(torch, args, dev):
    """Headline shape: SubjectiveActor(dim=2), T = 500, 2^18 candidates x ONE trajectory each, fp32.  Parity: the parameter
    gradient of 3 sampled candidates through torch.autograd (the same sweep) against central differences of the fp64 C oracle."""
    import numpy as np
    import lqg_amd
    from lqg_amd import workload
    B, T = 1 << 18, 500
    system, params = workload.headline_system(B, T, seed=1234, device=dev, dtype=torch.float32)
    x_ref = workload.simulate_one_trial_each(system, seed=3)
    x = workload.pack_trials(x_ref)

    def check(ll):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as OC
        OC.build()
        idx = [0, B // 2, B - 1]
        names = list(params)
        th = {k: params[k][idx].double().clone().requires_grad_(True) for k in names}
        m = lqg_amd.SubjectiveActor(dim=2, T=T, process_noise=1.0, dt=1.0 / 60, device=dev, dtype=torch.float64, **th)
        xs = x_ref[torch.as_tensor(idx, device=dev)].double()
        l3 = m.log_likelihood(xs)
        l3.sum().backward()
        worst = 0.0
        val_err = float((ll[torch.as_tensor(idx, device=dev), 0].double() / l3[:, 0].detach() - 1).abs().max())
        h = 1e-5
        for k in ("sigma_target", "action_cost", "subj_noise"):
            for j in range(len(idx)):
                vals = []
                for sgn in (+1, -1):
                    kw = {q: (th[q][j].item() * ((1 + sgn * h) if q == k else 1.0)) for q in names}
                    mm = lqg_amd.SubjectiveActor(dim=2, T=T, process_noise=1.0, dt=1.0 / 60, device="cpu", dtype=torch.float64, **kw)
                    host = lambda spec: {f: getattr(spec, f).numpy()[None] if getattr(spec, f).dim() == workload._batched_ndim(f) - 1
                                         else getattr(spec, f).numpy() for f in lqg_amd.LQGSpec._fields}
                    vals.append(float(OC.log_likelihood(host(mm.actor), host(mm.dynamics), xs[j:j + 1].cpu().numpy())[0, 0]))
                fd = (vals[0] - vals[1]) / (2 * h * th[k][j].item())
                an = float(th[k].grad[j])
                worst = max(worst, abs(an - fd) / max(1.0, abs(fd)))
        return {"samples": len(idx), "value_max_rel_err_f32_vs_f64_sweep": val_err,
                "grad_max_rel_err_vs_central_differences_of_the_fp64_oracle": worst}

    return _grad_leg(torch, dev, system, x, B, 1, T, (4, 6, 2, 4, 4), "value_and_grad_headline", fd_check=check)


def leg_value_and_grad_config3(torch, args, dev):
    """BASELINE config 3's shape: BoundedActor, T = 1067, 4096 candidates x 1024 shared trials, fp32: the per-trial mu-bar
    sweep (4.4e9 trial-steps) carries the time; the matrix adjoints run once per candidate."""
    import lqg_amd
    from lqg_amd import workload
    Bc, Nt, T = 4096, 1024, 1067
    system, _ = workload.bounded_system(Bc, T, seed=5, device=dev, dtype=torch.float32)
    truth = lqg_amd.BoundedActor(T=T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, action_variability=0.5, device=dev,
                                 dtype=torch.float32)
    x = workload.pack_trials(truth.simulate(13, n=Nt).contiguous())

    def check(ll):
        # the same sweep on 3 candidates x 24 trials, T = 120 in fp64, chained to the gradient of the FOUR model parameters through the
        # constructor (every bar of every field enters: a bar the split sweep wrongly left at zero, or an error in any of aA / aB / aF /
        # aQ / dA / dB / dF / aQf, shows here — ADVICE r05), against (i) the round-1 lane kernels, one (system, trial) pair per lane, and
        # (ii) central differences of the forward path (tests/test_adjoint.py holds the entry-by-entry comparisons with the restatement)
        from lqg_amd import options
        names = ("sigma_target", "sigma_cursor", "action_cost", "action_variability")
        base = dict(sigma_target=[12.0, 20.0, 31.0], sigma_cursor=[2.0, 3.0, 4.5], action_cost=[0.2, 0.3, 0.6], action_variability=[0.4, 0.5, 0.7])
        xs = truth.to(torch.float64).simulate(13, n=24)[:, :121].contiguous()

        def value_grad(sp):
            par = {k: torch.tensor(base[k], dtype=torch.float64, device=dev, requires_grad=True) for k in names}
            with options.override(ADJOINT_SP=sp):
                val = lqg_amd.BoundedActor(T=120, device=dev, dtype=torch.float64, **par).log_likelihood(xs).sum(-1)
                val.sum().backward()
            return val.detach(), torch.stack([par[k].grad for k in names], -1)
        v1, g1 = value_grad(1)
        v0, g0 = value_grad(0)
        h, fd = 1e-5, torch.zeros_like(g1)
        with torch.no_grad():
            for j, k in enumerate(names):
                f = lambda s_: lqg_amd.BoundedActor(T=120, device=dev, dtype=torch.float64, **{
                    q: torch.tensor(base[q], dtype=torch.float64, device=dev) * ((1 + s_ * h) if q == k else 1.0) for q in names}).log_likelihood(xs).sum(-1)
                fd[:, j] = (f(1) - f(-1)) / (2 * h * torch.tensor(base[k], dtype=torch.float64, device=dev))
        rel = lambda a_, b_: float(((a_ - b_).abs() / b_.abs().clamp_min(1e-12)).max())
        return {"candidates": 3, "trials": 24, "T": 120, "parameters": list(names),
                "parameter_gradient_max_rel_diff_split_vs_round1_kernels_f64": rel(g1, g0),
                "parameter_gradient_max_rel_diff_split_vs_central_differences_f64": rel(g1, fd),
                "value_max_rel_diff": float((v1 / v0 - 1).abs().max())}

    out = _grad_leg(torch, dev, system, x, Bc, Nt, T, (2, 2, 1, 2, 2), "value_and_grad_config3", fd_check=check)
    out["trial_steps_per_s"] = float(Bc) * Nt * T / (out["ms_per_value_and_grad"] * 1e-3)
    return out


LEGS = {
    "value_and_grad_headline": leg_value_and_grad_headline, "value_and_grad_config3": leg_value_and_grad_config3,
    "headline_other": leg_headline_other, "reference_layout": leg_reference_layout,
    "specialised_joint_n6": leg_specialised_joint,
    "dense_generic_f32": leg_dense_generic("f32"), "dense_generic_f64": leg_dense_generic("f64"),
    "config2_one_system": leg_one_system(2), "config4_one_system": leg_one_system(4),
    "one_vector_value_and_grad": leg_one_vector, "m2_f32": leg_m2, "timevarying_f64": leg_timevarying("f64"), "config3": leg_config3, "config4_sharded": leg_config4,
    "config5_one_system": leg_config5_one_system, "delay12": leg_delay12, "delay12_batch": leg_delay12_batch,
}


# legs that run only when asked for by name (`--only <leg>`): not part of the default line
ONLY_LEGS = {"timevarying_f32": leg_timevarying("f32")}


def extra_legs(torch, args, dev, only=None):
    """Secondary legs, outside the headline's timed region (same workload generators, same timing protocol).  Each is
    self-sufficient: its own roofline object (bound, achieved, peak, frac, traffic from the stamped PMC record)."""
    extra = {}
    for name, fn in (LEGS if only is None else dict(LEGS, **ONLY_LEGS)).items():
        if only is not None and name != only:
            continue
        key = name
        if name == "headline_other":
            key = "headline_" + ("f64" if args.dtype == "f32" else "f32")
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            extra[key] = fn(torch, args, dev)
        except Exception as e:          # a secondary leg must never cost the headline line
            extra[key] = {"error": repr(e)[:400]}
        if isinstance(extra[key], dict):
            extra[key]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    torch.cuda.empty_cache()
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

    extra = None
    if args.only:
        if world != 1:
            raise SystemExit("--only runs one secondary leg on one GPU")
        key = {"headline_f32": None, "headline_f64": None}.get(args.only, args.only)
        if key is None:                                    # the headline itself in the named dtype (PMC passes)
            out = headline_leg(torch, None, args, dev, 0, 1, args.only[-3:], args.log2_batch, args.steps, args.warmup)
        else:
            if key not in LEGS and key not in ONLY_LEGS:
                raise SystemExit(f"--only {key}: unknown leg; known: headline_f32, headline_f64, {', '.join(LEGS)}")
            out = extra_legs(torch, args, dev, only=key)
    elif args.config == 3:
        out = config3(torch, dist, args, dev, rank, world)
    elif args.config == 4:
        out = config4(torch, dist, args, dev, rank, world)
    else:
        out = headline_leg(torch, dist, args, dev, rank, world, args.dtype, args.log2_batch, args.steps, args.warmup,
                           cpu=(world == 1 and not args.no_cpu_baseline))
        if world == 1 and not args.no_extra and out is not None:
            try:
                extra = extra_legs(torch, args, dev)
            except Exception as e:          # the secondary legs must never cost the headline line
                extra = {"error": repr(e)}
        elif world > 1 and not args.no_extra:
            # N > 1: BASELINE config 4 literally — 262144 / N trials per rank, one all-reduce per step (strong scaling) — measured
            # collectively by the same ranks after the headline (every rank runs the same code on the same shapes: a failure is
            # the same exception on every rank before any collective, and costs only this leg)
            c4 = None
            if 262144 % world == 0:
                try:
                    class A4:
                        dtype, steps, warmup, share_gpu = "f32", 10, 2, args.share_gpu
                    c4 = config4(torch, dist, A4, dev, rank, world)
                except Exception as e:
                    c4 = {"error": repr(e)[:300]}
            if out is not None:
                extra = {"config4_sharded": c4}
    if rank == 0 and out is not None:
        # ONE line the driver can parse and keep whole (< 8000 characters): the headline + [value, unit, frac] per secondary
        # leg; the legs in full go to gpurun_out/bench_extra.json (bench_line.py; round 5's 24.6 kB line came back unparsed)
        if extra is not None:
            bench_line.write_extra(out, extra, os.path.join(ROOT, bench_line.EXTRA_FILE))
        if args.only:
            print(json.dumps(bench_line._strict(out)), flush=True)      # a PMC pass of one leg: that leg in full
        else:
            print(json.dumps(bench_line.compact(out, extra)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
