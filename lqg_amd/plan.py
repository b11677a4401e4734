"""LogLikelihoodPlan — everything `System.log_likelihood(x)` decides per call (decoupling into independent
components, structure-specialised vs generic library, workspace, argument structs), decided ONCE so that repeated
evaluations (benchmarks, optimisation loops over the same data) are pure launches.  `System.log_likelihood` builds a
throw-away plan; bench.py / bench_configs.py keep one."""
import ctypes as C

import torch

from lqg_amd import _abi, _hip, _hipev, options, specialize


# Above this much operator-stream workspace a multi-trial evaluation is run as one fused single-trial sweep per
# trial instead (N x the per-system work, no [system][step][operator] stream): e.g. 2^18 systems x 2 trials at
# n=6, T=500 would need 71 GB of stream for 2 trials' worth of work.
# (16 GiB through round 3, whatever the device held.  288 GB of HBM leave room for more: the limit is the smaller of 64 GiB
# and 40 % of the device's total memory (round 5: a property of the device, not of the process history) — 4096 candidates of the reference's delay-12 model
# (m = 65: 4.4 k reals per step) x 120 trials need a 36 GB stream; looping over the trials instead re-ran the 4096 system
# sweeps 120 times: 34 s against 0.3 s.)
OPS_WORKSPACE_LIMIT = None          # None: ops_workspace_limit(device); an int pins it (tests)


_ops_limit_by_device = {}


def ops_workspace_limit(device):
    """min(64 GiB, 40 % of the device's TOTAL memory), evaluated once per device.  (Round 4 used the FREE memory at plan time:
    a transient figure — the same problem could take the MIXED route or not, or the one-sweep-per-trial route, depending on what
    the process had allocated before or on which rank it ran; shards of a trial split must agree bitwise, so the limit is now a
    property of the device.)"""
    if OPS_WORKSPACE_LIMIT is not None:
        return OPS_WORKSPACE_LIMIT
    key = str(device)
    if key not in _ops_limit_by_device:
        try:
            total = torch.cuda.get_device_properties(device).total_memory
            _ops_limit_by_device[key] = int(min(64 << 30, 0.4 * total))
        except Exception:
            return 16 << 30
    return _ops_limit_by_device[key]
# Up to this many (system, trial) pairs a multi-trial evaluation is run as ONE fused sweep per PAIR (the system part is
# recomputed per trial — the lanes are idle anyway — and the per-trial sweep over the operator stream, a third
# latency-bound 500-step kernel, disappears): one parameter vector (or the 2P+1 finite-difference candidates) x tens of
# trials, the inner loop of lqg/infer/mle.py:17-23 and of NUTS.  LQG_FUSE_TRIALS_MAX=0 disables.
FUSE_TRIALS_MAX = 2048          # default of options "FUSE_TRIALS_MAX"
# TIME-PARALLEL system sweeps (csrc/lqg_scan.hpp: Riccati, Kalman and moment recursions as associative scans, log2(T)
# dependent combines instead of T dependent steps), followed by the time-chunked per-trial sweep.  The sequential sweeps
# cost T dependent steps whatever the number of systems (one lane each); the scans cost a fixed ~0.16 ms plus ~4 us
# (m = 4) .. ~14 us (m = 8) per system.  Measured crossover ON THE GPU TIMELINE (scripts/small_batch.py, T = 500, fp32):
# m = 4: ~16 systems (0.21 ms sequential), m = 8: ~55 systems (1.0 ms sequential); a throw-away plan also pays ~0.1 ms of
# host-side checks for the scan route (eigenvalue floor, conditioning), so the default rule stays below that and takes the
# scans for at most 8 (m / 4)^2 systems (8 at m <= 4, capped at 64) with at least scan_min_steps(m) steps.
# LQG_SCAN=0 never, LQG_SCAN=1 wherever the path is defined (no affine terms, floor provably inactive, u, y, d <= 4).
SCAN_MAX_SYSTEMS = 0            # default of option "SCAN_MAX_SYSTEMS" (0: the rule above)
SCAN_MIN_STEPS = 0              # default of option "SCAN_MIN_STEPS" (0: the rule below)
# The scan elements hold (F' Sigma_oo^-1 F)-type terms, so an ill-conditioned observed block costs the scans more digits
# than the sequential recursion: scripts/scan_cond.py, point mass with all four states observed, scan against sequential
# fp64 — cond((V V')[:d, :d]) <= 5.6e6: 1e-11; 5.6e8 (golden pointmass_d4_T50): 2e-6.  Above this condition number of the
# observed process-noise block the default rule keeps the sequential sweeps.
SCAN_MAX_COND = 1e7             # default of option "SCAN_MAX_COND"


# MIXED precision (include/lqg_hip.h: LQG_F32_SYS64).  An fp32 problem scored on many trials per system runs its per-system
# sweeps (Riccati, Kalman, moment recursion: data-independent, amortised over the trials) in fp64 over an fp64 image of the
# specs and rounds the per-step trial operators to fp32 ONCE; the per-trial sweep stays fp32.  With the fp32 recursions the
# operators carry their accumulated rounding into every trial: config 3 (T = 1067) max 1.7e-6 relative on the log-likelihood
# against the fp64 path, 5e-7 with the fp64 system sweeps.  Default: WHENEVER the operator stream is used
# (3 or more trials per system) — the result of a trial then does not depend on how many other trials were scored with it
# (shards of a trial split agree bitwise); LQG_MIXED_MIN_TRIALS raises the threshold for workloads of many systems with a
# handful of trials each, where the system sweeps dominate and doubling their cost is not amortised; LQG_MIXED=0 disables.
MIXED_MIN_TRIALS = 3            # default of option "MIXED_MIN_TRIALS"
MIXED_LONG_HORIZON = 600        # steps beyond which an fp32 problem leaves the in-lane sweeps for the stream path, any n
FUSE_F32_MAX_STEPS = 256        # steps up to which small fp32 (system, trial) batches run as all-fp32 fused pairs
# WIDE: an fp32 problem whose observed process-noise block (V V')[:d, :d] has a condition number above this runs — for ANY
# number of trials — every sweep, the per-trial one included, in fp64 over an fp64 image of the specs and of the data; the
# log-likelihoods are rounded to fp32 once.  The whitening of an ill-conditioned innovation (Li ~ sqrt(cond)) amplifies
# the fp32 rounding of the recursions and of the mean state: the golden `pointmass_d4_T50` (cond 5.6e8: every state of the
# point mass observed, velocity / activation noise 1e-3) gives 1.5e-3 or NaN on the all-fp32 in-lane sweeps and 7e-6 MIXED,
# 1e-7 on this route.  Threshold = SCAN_MAX_COND: BASELINE config 2 (the same model observed through target and cursor, cond
# 1e6) holds 6e-8 in fp32 and keeps its fp32 per-trial sweep.  LQG_F32_WIDE=0 disables (A/B, tests of the fp32 kernels).
F32_MAX_COND = 1e7              # default of option "F32_MAX_COND"


def scan_min_steps(m):
    """Horizon from which the scans beat one lane walking the recursion: the sequential step costs ~0.4 us at m = 4 and
    ~2 us at m = 8 (~m^2), the scans ~0.11-0.15 ms whatever T (config 1, m = 4, T = 100: 0.064 ms sequential, 0.11 ms
    scans; m = 4, T = 500: 0.20 / 0.14; m = 8, T = 500: 1.0 / 0.21)."""
    if options.get("SCAN_MIN_STEPS") > 0:
        return options.get("SCAN_MIN_STEPS")
    return int(max(64, 6000 / (m * m)))


def scan_max_systems(m, fp64=False):
    """(fp64 problems would tolerate twice as many on the GPU timeline — their sequential sweeps are ~1.5x slower, the scans
    are fp64 either way — but for a throw-away plan the scan route's extra host checks eat the difference: one vector's
    central differences, 9 systems x 50 trials at m = 4 in fp64: 1.01 ms fused pairs / 1.12 ms scans end to end.)"""
    if options.get("SCAN_MAX_SYSTEMS") > 0:
        return options.get("SCAN_MAX_SYSTEMS")
    if m > 24:
        # windows of 25 .. 64 (k_scan_level_rt: one 1024-lane workgroup per window, 50 .. 95 us per combine, one system's 500
        # windows already fill the chip twice — the levels run in a work-efficient order there, round 4):
        # DelayedSubjectiveActor (m = 65, T = 500, 50 trials) 2.4 ms for one system + 1.1 ms per further one against 25.4 ms
        # (fp64) / 12.1 ms (fp32) + 0.1 ms per system for the cooperative sequential sweeps, which give every system
        # its own CU (profiles/r04_scan_order_time.txt: 13 systems 15.5 vs 26.8 ms in fp64, 15.2 vs 13.5 ms in fp32)
        return 24 if fp64 else 10
    return int(min(64, max(8, 8 * (m / 4.0) ** 2)))


def _observed_noise_cond(sub, d):
    """Largest condition number of (V V')[:d, :d] of the dynamics over systems and steps (cached on the system).  Few
    blocks: one small device-to-host copy and the eigenvalues on the host (a batched eigvalsh of 2x2..4x4 blocks costs
    ~0.3 ms on the GPU, more than the evaluation it guards).  Many blocks (a batch of candidates): reduced on the device —
    closed form for d <= 2, batched eigvalsh beyond — and ONE scalar comes back."""
    import numpy as np
    cache = sub.__dict__.setdefault("_lqg_noise_cond", {})
    key = (int(d), specialize.spec_versions(sub))
    if key in cache:
        return cache[key]
    V = specialize._first(sub.dynamics.V.detach())[..., :d, :].double()   # (one time slice when the spec is time-invariant;
    VV = V @ V.transpose(-1, -2)                                          # widened BEFORE the product: an fp32 Gram matrix near
    #                                                                       the threshold has a rounding-noise smallest eigenvalue)
    if VV.numel() <= d * d * 4096:
        ev = np.linalg.eigvalsh(VV.cpu().numpy())
        lo, hi = np.maximum(ev[..., 0], 0.0), ev[..., -1]
        cache[key] = float(np.max(hi / np.maximum(lo, 1e-300)))
    else:
        if d == 1:
            lo = hi = VV[..., 0, 0]
        elif d == 2:
            a, b, c = VV[..., 0, 0], VV[..., 0, 1], VV[..., 1, 1]
            h, r = 0.5 * (a + c), torch.sqrt(0.25 * (a - c) ** 2 + b * b)
            hi = h + r
            lo = (a * c - b * b) / hi                                 # (the small root without cancellation)
        else:
            ev = torch.linalg.eigvalsh(VV)
            lo, hi = ev[..., 0], ev[..., -1]
        cache[key] = float((hi / lo.clamp_min(1e-300)).max())
    return cache[key]


def f32_needs_wide(sub, d):
    """True when an fp32 problem is evaluated over an fp64 image of its specs and data (F32_MAX_COND above)."""
    return (sub.actor.A.dtype == torch.float32 and sub.actor.A.is_cuda and options.flag("F32_WIDE")
            and _observed_noise_cond(sub, d) > options.get("F32_MAX_COND"))


def scan_eligible(lib, ln, sub, eps, systems_scale=1):
    """True when the time-parallel sweeps may serve this launch (include/lqg_hip.h: lqg_log_likelihood_scan).
    systems_scale: a caller whose decisions are taken once and replayed (infer/graphed.py) does not pay the scan route's
    host checks per evaluation and can afford the GPU-timeline crossover instead of the end-to-end one (6x in fp64, 2x in fp32:
    profiles/r02_*_small_batch_f64.txt — BoundedActor, 32 / 64 systems: 0.28 / 0.41 ms scans, 0.32 / 0.38 ms lane kernels)."""
    mode = options.get("SCAN")
    if mode == "0" or not hasattr(lib, "lqg_log_likelihood_scan"):
        return False
    if mode != "1" and not (ln.B <= min(64, (1 if ln.m > 24 else systems_scale) * scan_max_systems(ln.m, ln.dtype == torch.float64)) and ln.T >= scan_min_steps(ln.m)):
        return False
    if not lib.lqg_scan_supported(C.byref(ln.p)):
        return False
    from lqg_amd import decouple
    if not decouple.floor_provably_inactive(sub, eps):
        return False
    return mode == "1" or _observed_noise_cond(sub, ln.dims["d"]) <= options.get("SCAN_MAX_COND")


class LogLikelihoodPlan:
    def __init__(self, system, x, Sigma0=None, eps=1e-8, events=False, concurrent=None, stack=False, merge=True):
        """stack=True (persistent plans over a fixed dataset): decoupled components that share dims and sparsity
        pattern are concatenated along the system axis and solved by ONE launch of C*B systems (twice the waves in
        flight per SIMD at the headline shape); costs a one-time re-packing copy of x, so it is off for the throw-away
        plan of System.log_likelihood."""
        self.system = system
        self.out_dtype = None                    # set when a component runs WIDE (fp64 image of an fp32 problem)
        d = x.shape[-1]
        lib = _abi.load()
        parts = system.decoupled(d, Sigma0, eps=eps) or [(system, list(range(d)), None)]
        # merge=True: decoupled components with bit-identical specs are ONE system observed on different data columns
        # (every dim=2 zoo model): the per-system sweeps run once and the components become trials of that system
        self.merged = []
        if merge and len(parts) > 1 and not options.flag("NO_MERGE"):
            from lqg_amd import decouple
            merged_parts, merged_x = [], []
            for g in decouple.identical_groups(system, d, parts, Sigma0):
                merged_parts.append(parts[g[0]])
                merged_x.append(_trial_stack(x, [parts[i][1] for i in g]) if len(g) > 1 else None)
                self.merged.append(len(g))
            if any(m > 1 for m in self.merged):
                parts = merged_parts
            else:
                self.merged, merged_x = [], []
        self.n_stacked = 1
        if stack and len(parts) > 1 and Sigma0 is None and not self.merged:
            stacked = _stack_components(parts, x)
            if stacked is not None:
                self.n_stacked = len(parts)
                parts, x = [(stacked[0], list(range(stacked[1].shape[-1])), None)], stacked[1]
        self.work = []
        # (one ill-conditioned component takes the whole evaluation WIDE: the components' results are added)
        wide_any = any(f32_needs_wide(sub, len(cols)) for sub, cols, _ in parts)
        for ip, (sub, cols, bs) in enumerate(parts):
            contiguous = cols == list(range(cols[0], cols[-1] + 1))
            xs = x[..., cols[0]:cols[-1] + 1] if contiguous else x[..., cols]
            if self.merged and merged_x[ip] is not None:
                xs = merged_x[ip]                # [(B,) G*n, T+1, d_c]: the group's components as trials
            S0 = Sigma0 if (Sigma0 is None or bs is None) else Sigma0[..., bs, :][..., :, bs]
            n = xs.shape[-3]
            n_sys0 = sub.n_systems or 1
            sub0 = sub                           # (keeps the zoo class: its sparsity pattern is cached per class)
            wide = wide_any
            if wide:                             # ill-conditioned fp32 problem: fp64 image of specs AND data, any n
                sub = sub.to(torch.float64)
                xs = xs.to(torch.float64)
                S0 = None if S0 is None else S0.to(torch.float64)
                self.out_dtype = torch.float32
            use_scan = scan_eligible(_abi.load(), _hip.Launch(sub.actor, sub.dynamics, d=len(cols), n_trials=n, Sigma0=S0,
                                                               eps=eps), sub, eps) if sub.actor.A.is_cuda else False
            # (an fp32 problem over a long horizon keeps the operator-stream path: its system sweeps then run in fp64 —
            # MIXED, below — where the fused pairs would run every recursion in fp32; §6a: the fp32 tail passes 1e-6 near T = 1000)
            long_f32 = (sub.actor.A.dtype == torch.float32 and sub.T > MIXED_LONG_HORIZON
                        and options.flag("MIXED"))
            # (fused pairs in fp32 only up to FUSE_F32_MAX_STEPS: worst pair of scripts/fuzz_mixed.py 8.7e-7 at T = 500 —
            # too close to the north star's 1e-6 for a path that exists for speed on tiny problems; beyond, the stream path
            # runs MIXED: worst 2.8e-7)
            short_f32 = not (sub.actor.A.dtype == torch.float32 and sub.T > FUSE_F32_MAX_STEPS
                             and options.flag("MIXED"))
            # (a lane-kernel route: the cooperative kernels walk one system per WORKGROUP — replicating the system per trial
            # multiplies their work, and their per-trial sweep is time-chunked for few systems, csrc/lqg_trial_chunk.hpp)
            lane_dims = sub.actor.A.shape[-1] + sub.dynamics.A.shape[-1] <= _hip.LANE_MAX_JOINT
            fuse_pairs = ((not use_scan) and 2 < n and n_sys0 * n <= options.get("FUSE_TRIALS_MAX") and _time_invariant(sub)
                          and not long_f32 and short_f32 and lane_dims)
            if fuse_pairs:                       # (system, trial) pairs as n_sys0 * n one-trial systems
                sub, xs, S0 = _pairs_as_systems(sub, xs, S0, n_sys0, n)
                n_pairs, n = n, 1
            ln = _hip.Launch(sub.actor, sub.dynamics, d=len(cols), n_trials=n, Sigma0=S0, eps=eps)
            # liblqg_hip.so, or the auxiliary library of an unlisted shape; the time-parallel sweeps live in the main library
            # only — once they are chosen (few systems, long horizon) a cached auxiliary lane-kernel library must not take the
            # launch back
            lib = ln.require_gpu() if not use_scan else (ln.require_gpu() and _abi.load())
            nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
            # MIXED: from MIXED_MIN_TRIALS trials per system on (the operator stream is used anyway), and for ANY number of
            # trials beyond MIXED_LONG_HORIZON steps (one or two trials per system then leave the in-lane fp32 sweeps for
            # the stream path) — while the stream fits the workspace limit (2^20 systems x T = 1000 would need 150 GB: such a
            # batch keeps the in-lane fp32 sweeps, whose tail passes 1e-6 near T = 1000; fp64 is the remedy there)
            mixed = ((not use_scan) and ln.dtype == torch.float32 and options.flag("MIXED")
                     and (n >= max(3, options.get("MIXED_MIN_TRIALS")) or (long_f32 and n >= 1))
                     and lib.lqg_strategy(C.byref(ln.p)) == _abi.STRATEGY_LANE)
            if mixed:                            # fp64 image of the specs (a few kB per system), float trajectories
                sub_m = sub.to(torch.float64)
                ln_m = _hip.Launch(sub_m.actor, sub_m.dynamics, d=len(cols), n_trials=n, Sigma0=S0, eps=eps,
                                   traj_dtype=torch.float32)
                nbytes_m = lib.lqg_workspace_bytes(C.byref(ln_m.p), _abi.OP_LOG_LIKELIHOOD)
                if nbytes_m > ops_workspace_limit(ln.device) and ln_m.p.tuning.hilo == 0:
                    # the operator stream fits but stream + residual stream (hi + lo operators, about as large again) does not:
                    # rounded operators on fp64-built gains are the better fall-back than all-fp32 sweeps
                    ln_m.p.tuning.hilo = -1
                    nbytes_m = lib.lqg_workspace_bytes(C.byref(ln_m.p), _abi.OP_LOG_LIKELIHOOD)
                if nbytes_m <= ops_workspace_limit(ln.device):
                    sub, ln, nbytes = sub_m, ln_m, nbytes_m
                else:
                    mixed = False
            xb, is_b = _hip._prep_x(ln, xs)
            sp = None if use_scan else _hip.specialised_entry(ln, sub0, len(cols))
            if mixed and sp is None and not use_scan and ln.p.tuning.hilo == 0:
                ln.p.tuning.hilo = -1            # the generic kernels never read a residual stream: do not size one
                nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
            if use_scan:
                use_scan = lib is _abi.load()            # (an auxiliary lane-kernel library has no scan entry)
            scan_entry = None
            if use_scan:
                nbytes = lib.lqg_scan_workspace_bytes(C.byref(ln.p))
                # the per-trial sweep after the scans: the pattern library's (structural zeros of the operator compiled
                # out) when the system has one, else the main library's dense sweep
                spl = _hip.specialised_library(ln, sub0, len(cols), check_strategy=False) if n > 2 else None
                fn = C.cast(spl.lqg_trial_sweep_sp, C.c_void_p) if spl is not None else C.c_void_p(None)
                scan_entry = (lambda *a, _f=lib.lqg_log_likelihood_scan_with, _t=fn: _f(*a, _t))
                scan_sp = spl is not None
            if sp is not None and n == 2 and not mixed and not _hip._varies_or_affine(ln):   # the specialised library sweeps two trials in-lane: no operator stream (time-invariant kernels only)
                ln.p.n_trials = 1
                nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
                ln.p.n_trials = 2
            loop_trials = (not use_scan) and n > 1 and nbytes > ops_workspace_limit(ln.device)
            if loop_trials:                      # one fused sweep per trial: the problem describes ONE trial
                ln.p.n_trials = 1
                nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
            ev = None
            if events:
                ev = [_hipev.Event() for _ in range(4)]
                for i in range(4):
                    ln.p.phase_events[i] = ev[i].h
            ll_buf = ln.empty(n)
            ll_sb = n if ln.batched else 0
            if fuse_pairs:                       # [n_sys0 * n_pairs, 1] -> the caller's [(n_sys0,) n_pairs] view of it
                ll_buf = ll_buf.view(n_sys0, n_pairs) if sub0.n_systems is not None else ll_buf.view(n_pairs)
                n, ll_sb = n_pairs, 1
            self.work.append(dict(ln=ln, x=xb, traj=ln.traj(xb, is_b), ll=ll_buf, ll_sb=ll_sb, nbytes=nbytes, ev=ev,
                                  ws=torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ln.device),
                                  entry=(scan_entry if use_scan else (sp or lib.lqg_log_likelihood)),
                                  scan_sp=bool(use_scan and scan_sp),
                                  generic=lib.lqg_log_likelihood, scan=use_scan,
                                  specialised=sp is not None, n=n, fused_pairs=fuse_pairs, mixed=mixed, wide=wide,
                                  coop=bool(not use_scan and sp is None
                                            and lib.lqg_strategy(C.byref(ln.p)) == _abi.STRATEGY_COOP),
                                  pattern_key=(specialize.system_pattern(sub0, len(cols))[2] if sp is not None else None),
                                  loop_trials=loop_trials, is_b=is_b, group=(self.merged[ip] if self.merged else 1),
                                  dims=(sub.xdim, sub.bdim, sub.udim, sub.ydim, len(cols))))
        self.lib = lib
        self.device = self.work[0]["ln"].device
        self.ll = self.work[0]["ll"]
        if self.merged:                        # [.., G*n] per work item -> [.., n]: sum over the G merged components
            n0 = x.shape[-3]
            self.ll = torch.empty(tuple(self.ll.shape[:-1]) + (n0,), dtype=self.ll.dtype, device=self.ll.device)
        if self.n_stacked > 1:                 # [C*B, n] of the stacked launch -> per-solve sum over the C components
            self._ll_stacked = self.ll
            self.ll = torch.empty((self.ll.shape[0] // self.n_stacked,) + tuple(self.ll.shape[1:]),
                                  dtype=self.ll.dtype, device=self.ll.device)
        # concurrent=True runs the independent components on side streams.  With few systems (one parameter vector, a
        # handful of candidates) each per-system sweep is a single latency-bound wave and the components simply overlap:
        # the default (None) turns it on up to 2^17 systems — at most two waves per SIMD per launch (config 4: 8.2 -> ~5 ms wall) — and
        # for the time-varying sweeps at any size: they wait on their per-step spec loads, and two launches in flight hide each
        # other's (round 6, two components: fp64 2^16 systems 6.40 -> 4.25 ms, fp32 2^17 systems 6.2 -> 4.25 ms,
        # profiles/r06_timevarying.txt).  The time-invariant VALU-bound sweeps gain 4 % at B = 2^18 (the bench stacks their
        # components into one launch instead).
        if concurrent is None:
            n_sys = self.work[0]["ln"].B
            concurrent = len(self.work) > 1 and (n_sys <= (1 << 17) or any(_hip._varies_or_affine(wk["ln"]) for wk in self.work))
        self.side = [torch.cuda.Stream(device=self.device) for _ in self.work[1:]] if concurrent else []
        self._fork = torch.cuda.Event() if self.side else None
        self._join = [torch.cuda.Event() for _ in self.side]

    @property
    def description(self):
        w = self.work
        in_lane = w[0]["n"] == 1 or (w[0]["n"] == 2 and w[0]["specialised"])
        tail = ")" if in_lane else " + k_trial)"
        if all(k.get("scan") for k in w):
            kind = ("time-parallel scans (Riccati, Kalman, moment recursion: k_scan_level x log2 T) + per-trial sweep ("
                    + ("pattern library" if all(k.get("scan_sp") for k in w) else "dense") + ", time-chunked when few trials)")
            if len(w) > 1:
                kind += f", {len(w)} decoupled components of dims (x,b,u,y,d)={w[0]['dims']}"
            if self.merged and max(self.merged) > 1:
                kind += f"; {max(self.merged)} identical components as trials of one system"
            return kind
        if all(k.get("coop") for k in w):
            kind = ("cooperative, one workgroup per system, run-time dims (k_coop_riccati + k_coop_forward + per-trial sweep "
                    "k_trial / k_coop_trial_rows)")
            if len(w) > 1:
                kind += f", {len(w)} decoupled components of dims (x,b,u,y,d)={w[0]['dims']}"
            return kind
        kind = ("structure-specialised (k_riccati_sp + k_forward_sp" if all(k["specialised"] for k in w) else
                "generic dense (k_riccati + k_forward") + tail
        if all(k.get("mixed") for k in w):
            kind += " [system sweeps in fp64, operators rounded to fp32 once, per-trial sweep fp32]"
        if any(k.get("wide") for k in w):
            kind += " [ill-conditioned fp32 problem: every sweep over an fp64 image of specs and data, results rounded to fp32 once]"
        if any(k.get("loop_trials") for k in w):
            kind += " [operator stream over the workspace limit: one fused sweep per trial]"
        if len(w) > 1:
            kind += f", {len(w)} decoupled components of dims (x,b,u,y,d)={w[0]['dims']}"
        if self.n_stacked > 1:
            kind += f", {self.n_stacked} decoupled components of dims (x,b,u,y,d)={w[0]['dims']} stacked into one launch"
        if self.merged and max(self.merged) > 1:
            kind += (f"; {max(self.merged)} identical decoupled components of dims (x,b,u,y,d)={w[0]['dims']} solved as ONE "
                     "system with the components as in-lane trials")
        return kind

    def run(self):
        """Launch the whole evaluation on the current stream; returns ll[(B,) n] (a buffer owned by the plan)."""
        if self.work[0]["n"] == 0:
            return self._result()              # no trials: nothing to launch
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            if self.side:
                self._fork.record(main)
            for i, wk in enumerate(self.work):
                ln = wk["ln"]
                stream = self.side[i - 1] if (i > 0 and self.side) else main
                if stream is not main:
                    stream.wait_event(self._fork)      # inputs produced on the caller's stream are ready
                ll_sb = wk["ll_sb"]
                trials = range(wk["n"]) if wk["loop_trials"] else (None,)
                for tr in trials:
                    traj, llp = wk["traj"], wk["ll"].data_ptr()
                    if tr is not None:           # view of trial `tr`: same strides, shifted base pointers
                        esz = wk["x"].element_size()
                        traj = _abi.Traj(traj.ptr + tr * traj.sn * esz, traj.sb, traj.sn, traj.st, traj.sd)
                        llp += tr * esz
                    args = (C.byref(ln.p), traj, C.c_void_p(llp), ll_sb, 1,
                            C.c_void_p(wk["ws"].data_ptr()), wk["nbytes"], C.c_void_p(stream.cuda_stream))
                    if wk["entry"](*args) != 0:
                        if wk["specialised"]:      # the specialised library refused: use the generic one
                            wk["entry"], wk["specialised"] = wk["generic"], False
                        need = self.lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
                        if need > wk["nbytes"]:    # (the in-lane two-trial sweep needed no operator stream)
                            wk["ws"] = torch.empty(int(need), dtype=torch.uint8, device=ln.device)
                            wk["nbytes"] = need
                            args = args[:5] + (C.c_void_p(wk["ws"].data_ptr()), wk["nbytes"]) + args[7:]
                        _abi.check(wk["generic"](*args), "lqg_log_likelihood")
                if stream is not main:
                    self._join[i - 1].record(stream)
            for ev in self._join:
                main.wait_event(ev)
            if self.merged:
                for i, wk in enumerate(self.work):
                    part = wk["ll"].view(*wk["ll"].shape[:-1], wk["group"], -1)
                    if i == 0:
                        torch.sum(part, dim=-2, out=self.ll)
                    else:
                        self.ll.add_(part.sum(-2))
            else:
                for wk in self.work[1:]:
                    self.ll.add_(wk["ll"])     # log p(x) = sum over independent components
            if self.n_stacked > 1:
                torch.sum(self._ll_stacked.view(self.n_stacked, *self.ll.shape), dim=0, out=self.ll)
        return self._result()

    def _result(self):
        return self.ll if self.out_dtype is None else self.ll.to(self.out_dtype)       # WIDE: rounded to fp32 once

    def use_events(self, sets):
        """Point the library's phase-event hook of every component at its own event set for the NEXT run:
        `sets[i]` = 4 _hipev.Event for component i (lets a caller keep one set per timed step without syncing)."""
        for wk, ev in zip(self.work, sets):
            wk["ev"] = ev
            for i in range(4):
                wk["ln"].p.phase_events[i] = ev[i].h

    def phase_ms(self):
        """(riccati, forward, trial) milliseconds of the last run, summed over components (events=True only)."""
        for wk in self.work:                       # components may run on different streams
            wk["ev"][3].synchronize()
        return tuple(sum(wk["ev"][i].elapsed_ms(wk["ev"][i + 1]) for wk in self.work) for i in range(3))


def _time_invariant(sub):
    for spec in (sub.actor, sub.dynamics):
        for f in ("A", "B", "F", "V", "W", "Q", "R"):
            t = getattr(spec, f)
            if t.shape[-3] > 1 and t.stride(-3) != 0:
                return False
    return True


def _pairs_as_systems(sub, xs, S0, B, n):
    """System of B * n one-trial systems: pair (s, k) = system s of `sub` with trial k of `xs` (system-major order).
    Spec fields with a system axis are repeated n times (a few kB), shared fields stay shared (system stride 0)."""
    from lqg_amd.spec import LQGSpec
    from lqg_amd.system import System
    from lqg_amd.utils import mark_zero
    batched = sub.n_systems is not None

    def rep_spec(spec):
        out = {}
        for f in LQGSpec._fields:
            t = getattr(spec, f)
            zero = getattr(t, "_lqg_zero", False)
            notime = f in ("Qf", "qf")
            nd = (1 if f in ("q", "qf", "r") else 2) + (0 if notime else 1)
            has_b = t.dim() == nd + 1
            if not notime:
                tax = -(2 if f in ("q", "r") else 3)
                T = t.shape[tax]
                base = t.select(tax, 0)
                base = base.repeat_interleave(n, dim=0) if has_b else base.unsqueeze(0).expand(B * n, *base.shape)
                t2 = base.unsqueeze(tax).expand(*base.shape[:base.dim() + tax + 1], T, *base.shape[base.dim() + tax + 1:])
            else:
                t2 = t.repeat_interleave(n, dim=0) if has_b else t.unsqueeze(0).expand(B * n, *t.shape)
            out[f] = mark_zero(t2) if zero else t2
        return LQGSpec(**out)

    a = rep_spec(sub.actor)
    d = a if sub.actor is sub.dynamics else rep_spec(sub.dynamics)
    if xs.dim() == 3:
        xe = xs.unsqueeze(0).expand(B, *xs.shape)
    else:
        xe = xs.expand(B, *xs.shape[1:])
    xe = xe.reshape(B * n, 1, *xs.shape[-2:])
    if S0 is not None and S0.dim() == 3:
        S0 = S0.repeat_interleave(n, dim=0)
    del batched
    return System(actor=a, dynamics=d), xe, S0


def _trial_stack(x, cols_list, rows=None):
    """Data columns of G identical components as G*n trials of one system: x[(B,) n, T+1, d] -> [(B,) G*n, T+1, d_c],
    re-laid so that the system index (or, without one, the trial index) is the fastest-varying one in HBM.
    rows (default: option X4_LAYOUT): fp32 batches whose lane reads exactly four floats per row (G*n*d_c == 4) are laid
    [T+1][B][G*n][d_c] instead — one 16-byte vector per lane and step (k_forward_sp<X4>)."""
    comps = [x[..., c] for c in cols_list]
    st = torch.stack(comps, dim=-4)                                   # [(B,) G, n, T+1, d_c]
    st = st.reshape(*st.shape[:-4], st.shape[-4] * st.shape[-3], *st.shape[-2:])
    rows = options.flag("X4_LAYOUT") if rows is None else rows
    if rows and st.dim() == 4 and st.dtype == torch.float32 and st.shape[1] * st.shape[3] == 4:
        return st.permute(2, 0, 1, 3).contiguous().permute(1, 2, 0, 3)    # storage [T+1][B][G*n][d_c]
    if st.dim() == 4:
        return st.permute(1, 2, 3, 0).contiguous().permute(3, 0, 1, 2)    # storage [G*n][T+1][d_c][B]
    return st.permute(1, 2, 0).contiguous().permute(2, 0, 1)              # storage [T+1][d_c][G*n]


def _stack_components(parts, x):
    """Concatenate decoupled components of identical shape and sparsity pattern along the system axis:
    returns (stacked System of C*B systems, x re-packed as [C*B, 1, T+1, d_c]) or None when they do not match."""
    from lqg_amd import specialize, workload
    from lqg_amd.spec import LQGSpec
    from lqg_amd.system import System

    subs = [p[0] for p in parts]
    B = subs[0].n_systems
    if B is None or x.dim() != 4 or x.shape[1] != 1 or x.shape[0] != B:
        return None
    shapes = {(s.xdim, s.bdim, s.udim, s.ydim, len(c)) for s, c, _ in parts}
    keys = {specialize.system_pattern(s, len(c))[2] for s, c, _ in parts}
    if len(shapes) != 1 or len(keys) != 1:
        return None
    T = subs[0].T

    def cat(name, which):
        ts = []
        for s in subs:
            t = getattr(getattr(s, which), name)
            if getattr(t, "_lqg_zero", False):
                return t                                    # known-zero affine terms stay shared NULLs
            notime = name in ("Qf", "qf")
            vec = name in ("q", "qf", "r")
            base_nd = (1 if vec else 2) + (0 if notime else 1)
            if not notime:
                tax = -(2 if vec else 3)
                if t.stride(tax) != 0 and t.shape[tax] > 1:
                    return None                             # time-varying: not stacked
                t = t.select(tax, 0)
                base_nd -= 1
            if t.dim() == base_nd:
                t = t.expand(B, *t.shape)
            ts.append(t)
        out = torch.cat(ts, dim=0)
        if name not in ("Qf", "qf"):
            nd = 1 if name in ("q", "r") else 2
            out = out.unsqueeze(-(nd + 1)).expand(*out.shape[:-nd], T, *out.shape[-nd:])
        return out

    fields = {}
    for which in ("actor", "dynamics"):
        vals = {}
        for f in LQGSpec._fields:
            v = cat(f, which)
            if v is None:
                return None
            vals[f] = v
        fields[which] = LQGSpec(**vals)
    xs = torch.cat([x[..., c[0]:c[-1] + 1] for _, c, _ in parts], dim=0)      # [C*B, 1, T+1, d_c]
    return System(actor=fields["actor"], dynamics=fields["dynamics"]), workload.pack_trials(xs)
