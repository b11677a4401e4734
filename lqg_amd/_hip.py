"""Launch glue: torch tensors -> include/lqg_hip.h argument structs -> liblqg_hip.so.

PyTorch is plumbing here (device memory, the current HIP stream); every number is produced by the HIP
kernels behind the C ABI.  There is NO CPU path: tensors must live on a `cuda` (ROCm) device, otherwise
LqgHipError is raised.
"""
import ctypes as C

import torch

from lqg_amd import _abi, options
from lqg_amd._abi import LqgHipError
from lqg_amd.spec import LQGSpec

_VEC = {"q", "qf", "r"}
_NOTIME = {"Qf", "qf"}
_DT = {torch.float32: _abi.F32, torch.float64: _abi.F64}


def _es(t):
    return list(t.stride())


def _is_zero(t):
    return t is None or getattr(t, "_lqg_zero", False)


def spec_is_batched(spec: LQGSpec):
    return spec.A.dim() == 4


def _field_view(t, name, batched):
    """View of one spec field; a field without the system axis is shared (sb = 0)."""
    base = (1 if name in _VEC else 2) + (0 if name in _NOTIME else 1)
    has_b = t.dim() == base + 1
    if t.dim() not in (base, base + 1) or (has_b and not batched):
        raise LqgHipError(f"spec field {name}: unexpected shape {tuple(t.shape)}")
    return _abi.mat_view(t.data_ptr(), t.shape, _es(t), has_b, name not in _NOTIME, name in _VEC)


class LqgLayoutWarning(UserWarning):
    pass


_layout_warned = False


def _layout_hint(t, name, B):
    """Once per process: a big batch of TIME-VARYING spec arrays stored system-major makes every per-step wave-load touch
    64 cache lines (measured: mode M2 20.8 ms instead of 12.7).  Correct either way -- the C ABI takes any strides."""
    global _layout_warned
    if _layout_warned or B < 4096 or name in _NOTIME or t.dim() != (3 if name in _VEC else 4):
        return
    if t.shape[1] > 1 and t.stride(1) != 0 and t.stride(0) != 1:
        import warnings
        _layout_warned = True
        warnings.warn(f"lqg_amd: time-varying spec field {name}{tuple(t.shape)} has system stride {t.stride(0)}; the sweeps read "
                      "it at every step with one system per lane -- lqg_amd.workload.pack_systems(t) stores it "
                      "[T][r][c][B] (same logical tensor) and is ~1.6x faster for the materialising sweeps", LqgLayoutWarning,
                      stacklevel=3)


class Launch:
    """One problem description plus the tensors it points into (kept alive for the call)."""

    def __init__(self, actor: LQGSpec, dynamics: LQGSpec = None, d=None, n_trials=1, Sigma0=None, eps=1e-8,
                 traj_dtype=None):
        """traj_dtype=torch.float32 with float64 specs describes the MIXED problem LQG_F32_SYS64 (include/lqg_hip.h):
        trajectories, results and the operator stream are float, the spec arrays double."""
        dynamics = actor if dynamics is None else dynamics
        A = actor.A
        if A.dtype not in _DT:
            raise LqgHipError(f"unsupported dtype {A.dtype}: float32 or float64")
        self.spec_dtype = A.dtype
        self.dtype, self.device = (traj_dtype or A.dtype), A.device     # dtype: trajectories / results
        self.mixed = self.dtype != self.spec_dtype
        if self.mixed and (self.dtype, self.spec_dtype) != (torch.float32, torch.float64):
            raise LqgHipError(f"mixed precision is float32 trajectories over float64 specs, got {self.dtype} / {self.spec_dtype}")
        self.batched = spec_is_batched(actor) or spec_is_batched(dynamics)
        self.B = 1
        for sp in (actor, dynamics):
            if spec_is_batched(sp):
                self.B = max(self.B, sp.A.shape[0])
        self.T = A.shape[-3]
        b, u, y, x = A.shape[-1], actor.B.shape[-1], actor.F.shape[-2], dynamics.A.shape[-1]
        self.dims = dict(x=x, b=b, u=u, y=y, d=(x if d is None else int(d)), nva=actor.V.shape[-1],
                         nwa=actor.W.shape[-1], nvd=dynamics.V.shape[-1], nwd=dynamics.W.shape[-1])
        self.m = x + b
        p = _abi.Problem()
        p.dtype, p.T, p.n_sys, p.n_trials, p.eps = (_abi.F32_SYS64 if self.mixed else _DT[A.dtype]), self.T, self.B, n_trials, float(eps)
        p.dims = _abi.Dims(**self.dims)
        options.fill_tuning(p.tuning)          # which of the library's equivalent kernels serve it (include/lqg_hip.h: lqg_tuning)
        self._keep = []
        for spec, dst, fields in ((actor, p.actor, _abi.SPEC_FIELDS), (dynamics, p.dynamics, ("A", "B", "F", "V", "W"))):
            for f in _abi.SPEC_FIELDS:
                t = getattr(spec, f) if f in fields else None
                if t is None or (f in ("q", "qf", "P", "r") and _is_zero(t)):
                    setattr(dst, f, _abi.NULL_VIEW)
                    continue
                if t.dtype != self.spec_dtype or t.device != self.device:
                    raise LqgHipError(f"spec field {f}: dtype/device {t.dtype}/{t.device} differs from A's")
                self._keep.append(t)
                setattr(dst, f, _field_view(t, f, self.batched))
                _layout_hint(t, f, self.B)
        if Sigma0 is not None:
            S0 = Sigma0.to(dtype=self.spec_dtype, device=self.device)
            self._keep.append(S0)
            p.Sigma0 = _abi.mat_view(S0.data_ptr(), S0.shape, _es(S0), S0.dim() == 3, False, False)
        else:
            p.Sigma0 = _abi.NULL_VIEW
        self.p = p

    # ---- helpers
    def require_gpu(self, family=_abi.FAM_FORWARD):
        if self.device.type != "cuda":
            raise LqgHipError(
                f"lqg_amd computes on MI355X only: tensors are on '{self.device}'. Move the spec to a cuda "
                "(ROCm) device; there is no CPU fallback.")
        return _abi.library_for(self.dims, family, n_sys=self.B)

    def lead(self):
        return (self.B,) if self.batched else ()

    def empty(self, *shape):
        return torch.empty(self.lead() + tuple(shape), dtype=self.dtype, device=self.device)

    def empty_system_fastest(self, *shape):
        """Logical [(B,) *shape] with the SYSTEM index fastest in memory (storage [*shape][B]): one system per lane then
        reads / writes 64 consecutive elements per (step, entry) — for intermediates that only kernels of this library touch."""
        if not self.batched:
            return self.empty(*shape)
        t = torch.empty(tuple(shape) + (self.B,), dtype=self.dtype, device=self.device)
        return t.permute(len(shape), *range(len(shape)))

    def view(self, t, vector=False):
        return _abi.mat_view(t.data_ptr(), t.shape, _es(t), self.batched, True, vector)

    def traj(self, t, batched=None):
        batched = self.batched if batched is None else batched
        return _abi.traj_view(t.data_ptr(), t.shape, _es(t), batched)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, lib, op):
        nbytes = lib.lqg_workspace_bytes(C.byref(self.p), op)
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device), nbytes


def riccati_backward(spec: LQGSpec, eps=1e-8, system_fastest=False):
    """system_fastest: lay the gains out [T][u][b][B] in memory (same logical shape): callers that only hand them on to
    another kernel of this library (System.simulate) get coalesced writes here and coalesced reads there."""
    ln = Launch(spec, eps=eps)
    lib = ln.require_gpu(_abi.FAM_RICCATI)          # k_riccati is instantiated per (b, u) only
    dm = ln.dims
    mk = ln.empty_system_fastest if system_fastest else ln.empty
    L, l, H = mk(ln.T, dm["u"], dm["b"]), mk(ln.T, dm["u"]), mk(ln.T, dm["u"], dm["u"])
    with torch.cuda.device(ln.device):
        _abi.check(lib.lqg_riccati_backward(C.byref(ln.p), ln.view(L), ln.view(l, vector=True), ln.view(H),
                                            ln.stream()), "lqg_riccati_backward")
    return L, l, H


def kalman_forward(spec: LQGSpec, Sigma0=None, system_fastest=False):
    ln = Launch(spec, Sigma0=Sigma0)
    lib = ln.require_gpu(_abi.FAM_KALMAN)           # k_kalman is instantiated per (b, y) only
    K = (ln.empty_system_fastest if system_fastest else ln.empty)(ln.T, ln.dims["b"], ln.dims["y"])
    with torch.cuda.device(ln.device):
        _abi.check(lib.lqg_kalman_forward(C.byref(ln.p), ln.view(K), ln.stream()), "lqg_kalman_forward")
    return K


def _prep_x(ln, x):
    """x[n,T+1,d] (shared by all systems) or [B,n,T+1,d]."""
    if x.dim() not in (3, 4):
        raise LqgHipError(f"x must be [n,T+1,d] or [B,n,T+1,d], got {tuple(x.shape)}")
    if x.shape[-2] != ln.T + 1:
        raise LqgHipError(f"x has {x.shape[-2]} rows; a system with T={ln.T} steps needs T+1={ln.T + 1}")
    if x.dtype != ln.dtype or x.device != ln.device:
        x = x.to(dtype=ln.dtype, device=ln.device)
    xb = x.dim() == 4
    if xb and x.shape[0] not in (1, ln.B):
        raise LqgHipError(f"x has {x.shape[0]} systems, spec has {ln.B}")
    return x, xb


def conditional_moments(actor, dynamics, x, Sigma0=None, eps=1e-8, want_mu=True, want_sigma=True, system=None):
    """x[n,T+1,d] | [B,n,T+1,d] -> mu[(B,)n,T,m], Sigma[(B,)T,m,m].  system: the System that owns the two specs — the
    scan-eligibility checks (eigenvalue floor, conditioning: one host synchronisation each) are cached on it."""
    d, n = x.shape[-1], x.shape[-3]
    if actor.A.dtype == torch.float32 and actor.A.is_cuda:
        # an fp32 problem whose observed noise block is ill-conditioned (plan.F32_MAX_COND: the point mass seen in full,
        # cond 5.6e8) is evaluated over an fp64 image of specs and data and rounded once — the policy of the log-likelihood
        # (plan.f32_needs_wide), whatever kernel family a caller forces: all-fp32 sweeps return 1e-3-wrong moments or NaN there
        from lqg_amd import plan as _plan
        from lqg_amd.system import System
        owner = system if system is not None else System(actor=actor, dynamics=dynamics)
        if _plan.f32_needs_wide(owner, d):
            o64 = owner.to(torch.float64)
            mu, Sig = conditional_moments(o64.actor, o64.dynamics, x.double(), None if Sigma0 is None else Sigma0.double(), eps,
                                          want_mu, want_sigma, system=o64)
            return (None if mu is None else mu.float()), (None if Sig is None else Sig.float())
    ln = Launch(actor, dynamics, d=d, n_trials=n, Sigma0=Sigma0, eps=eps)
    lib = ln.require_gpu()
    x, xb = _prep_x(ln, x)
    mu = ln.empty(n, ln.T, ln.m) if want_mu else None
    Sig = ln.empty(ln.T, ln.m, ln.m) if want_sigma else None
    use_scan = False
    if hasattr(_abi.load(), "lqg_conditional_moments_scan"):
        from lqg_amd import plan as _plan          # (same rule as the log-likelihood: few systems, long horizon)
        from lqg_amd.system import System
        use_scan = _plan.scan_eligible(_abi.load(), ln, system if system is not None else System(actor=actor, dynamics=dynamics), eps)
        if use_scan:
            lib = _abi.load()                      # (the scans live in the main library, whatever serves the shape's lane kernels)
    with torch.cuda.device(ln.device):
        if use_scan:                     # time-parallel system sweeps (csrc/lqg_scan.hpp)
            nbytes = lib.lqg_scan_workspace_bytes(C.byref(ln.p))
            ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ln.device)
            entry, what = lib.lqg_conditional_moments_scan, "lqg_conditional_moments_scan"
        else:
            ws, nbytes = ln.workspace(lib, _abi.OP_CONDITIONAL_MOMENTS)
            entry, what = lib.lqg_conditional_moments, "lqg_conditional_moments"
        _abi.check(entry(
            C.byref(ln.p), ln.traj(x, xb), ln.traj(mu) if want_mu else _abi.NULL_TRAJ,
            ln.view(Sig) if want_sigma else _abi.NULL_VIEW, C.c_void_p(ws.data_ptr()), nbytes, ln.stream()), what)
    return mu, Sig


def solve_materialised(actor, dynamics, x, Sigma0=None, eps=1e-8, out=None, system=None):
    """One pass producing everything the reference materialises: dict(L, l, H, K, mu, Sigma, ll).

    `out` may pre-supply any of those tensors (any strides, e.g. [T][element][system] storage).  system: the System that owns
    the two specs — with it (one trial per system, no affine cost terms) the pass runs on the structure-specialised library
    of the specs' sparsity pattern, time-varying specs included (materialised_entry); the pattern is cached on the system."""
    d, n = x.shape[-1], x.shape[-3]
    ln = Launch(actor, dynamics, d=d, n_trials=n, Sigma0=Sigma0, eps=eps)
    lib = ln.require_gpu()
    x, xb = _prep_x(ln, x)
    dm, T, m = ln.dims, ln.T, ln.m
    o = dict(out or {})
    o.setdefault("L", ln.empty(T, dm["u"], dm["b"]))
    o.setdefault("l", ln.empty(T, dm["u"]))
    o.setdefault("H", ln.empty(T, dm["u"], dm["u"]))
    o.setdefault("K", ln.empty(T, dm["b"], dm["y"]))
    o.setdefault("mu", ln.empty(n, T, m))
    o.setdefault("Sigma", ln.empty(T, m, m))
    o.setdefault("ll", ln.empty(n))
    with torch.cuda.device(ln.device):
        ws, nbytes = ln.workspace(lib, _abi.OP_CONDITIONAL_MOMENTS)
        sp = materialised_entry(ln, system, d) if (system is not None and n == 1 and lib is _abi.load()) else None
        if sp is not None:
            o["l"].zero_()
            rc = sp(C.byref(ln.p), ln.traj(x, xb), ln.view(o["L"]), _abi.NULL_VIEW, ln.view(o["H"]), ln.view(o["K"]),
                    ln.traj(o["mu"]), ln.view(o["Sigma"]), C.c_void_p(o["ll"].data_ptr()),
                    o["ll"].stride(0) if ln.batched else 0, C.c_void_p(ws.data_ptr()), nbytes, ln.stream())
            if rc == 0:
                return o
        _abi.check(lib.lqg_solve_materialised(
            C.byref(ln.p), ln.traj(x, xb), ln.view(o["L"]), ln.view(o["l"], vector=True), ln.view(o["H"]),
            ln.view(o["K"]), ln.traj(o["mu"]), ln.view(o["Sigma"]), C.c_void_p(o["ll"].data_ptr()),
            o["ll"].stride(0) if ln.batched else 0, o["ll"].stride(-1), C.c_void_p(ws.data_ptr()), nbytes, ln.stream()),
            "lqg_solve_materialised")
    return o


LANE_MAX_JOINT = 20             # largest x + b a lane (register-resident) kernel is generated for (tracking/delay.py)


def specialised_entry(ln, system, d):
    """The structure-specialised `lqg_log_likelihood_sp` for this launch, or None (lqg_amd/specialize.py)."""
    lib = specialised_library(ln, system, d)
    return None if lib is None else lib.lqg_log_likelihood_sp


def specialised_library(ln, system, d, check_strategy=True):
    """The structure-specialised library for this launch, or None (lqg_amd/specialize.py).

    Time-invariant specs without affine cost terms: the pattern of the spec VALUES (zoo classes: of the class).  Specs that vary
    in time (the reference's data model is (T, ...)-stacked: lqg/spec.py:5-19, lqg/utils.py:10-35) and / or carry affine cost
    terms, as long as they keep one sparsity pattern over systems and steps (round 6): the conservative masks of
    specialize.pattern_of_time_varying; `lqg_log_likelihood_sp` then runs k_riccati_tv_sp -> k_forward_tv_sp (csrc/lqg_sp_entry.hpp:
    run_sp_tv).  LQG_NO_SPECIALIZE=1 forces the generic dense library (A/B measurements, tests)."""
    import os
    if system is None or options.flag("NO_SPECIALIZE") or ln.p.n_trials < 1:
        return None
    if ln.m > LANE_MAX_JOINT:
        return None             # (pattern libraries hold lane kernels: one system's matrices in registers)
    if check_strategy and _abi.load().lqg_strategy(C.byref(ln.p)) == _abi.STRATEGY_COOP:
        return None             # few systems of a large joint dimension: the cooperative kernels of the main library
    from lqg_amd import specialize
    if _varies_or_affine(ln):
        dims, masks, key = _time_varying_pattern(ln, system, d)
        # a pattern nobody has compiled yet costs ~40 s of hipcc: only for work that repays it (TV_JIT_MIN_WORK system-steps, the
        # scale of _abi.JIT_MIN_SYSTEMS); a few systems over a short horizon run on the dense generic kernels in milliseconds.
        # (Time-invariant patterns keep round 5's rule — compile on first use: the zoo's are prebuilt, a user's model is one pattern.)
        if ln.B * ln.T < options.get("TV_JIT_MIN_WORK") and not specialize.pattern_on_disk(key):
            return None
    else:
        dims, masks, key = specialize.system_pattern(system, d)
    return specialize.load_pattern(key, dims, masks)


def _varies_or_affine(ln):
    """True when the launch's specs vary in time (a non-zero time stride at T > 1) or carry affine cost terms: what the
    time-invariant kernels of the pattern libraries do not serve."""
    p = ln.p
    varying = ln.T > 1 and any(getattr(spec, f).st != 0 for spec, fields in ((p.actor, ("Q", "R", "A", "B", "V", "F", "W")),
                                                                             (p.dynamics, ("A", "B", "V", "F", "W"))) for f in fields)
    return varying or any(getattr(p.actor, f).ptr for f in ("q", "qf", "P", "r"))


def _time_varying_pattern(ln, system, d):
    """(dims, masks, key) of specialize.pattern_of_time_varying, cached on the system per (d, versions of its spec tensors): the
    masks are reductions over every system and step — a few launches, not something to repeat per evaluation."""
    from lqg_amd import specialize
    cache = system.__dict__.setdefault("_lqg_materialised_pattern", {})
    key = (int(d), specialize.spec_versions(system), "tv")
    if key not in cache:
        dims, masks = specialize.pattern_of_time_varying(system, d)
        cache[key] = (dims, masks, specialize.pattern_key(dims, masks))
    return cache[key]


def materialised_entry(ln, system, d):
    """`lqg_solve_materialised_sp` of the structure-specialised library for this launch — time-varying specs included, as
    long as they keep one sparsity pattern (specialize.pattern_of_time_varying) — or None: then `lqg_solve_materialised` of
    the main library serves the call (dense kernels).  One trial per system, no affine cost terms."""
    import os
    if system is None or options.flag("NO_SPECIALIZE") or ln.p.n_trials != 1 or ln.m > LANE_MAX_JOINT:
        return None
    if any(getattr(ln.p.actor, f).ptr for f in ("q", "qf", "P", "r")):
        return None
    from lqg_amd import specialize
    cache = system.__dict__.setdefault("_lqg_materialised_pattern", {})
    key = (int(d), specialize.spec_versions(system))
    if key not in cache:
        varying = any(getattr(spec, f).st != 0 for spec, fields in ((ln.p.actor, ("Q", "R", "A", "B", "V", "F", "W")),
                                                                    (ln.p.dynamics, ("A", "B", "V", "F", "W"))) for f in fields)
        cache[key] = _time_varying_pattern(ln, system, d) if (varying and ln.T > 1) else specialize.system_pattern(system, d)
    dims, masks, pkey = cache[key]
    lib = specialize.load_pattern(pkey, dims, masks)
    return getattr(lib, "lqg_solve_materialised_sp", None) if lib is not None else None


def sum_trials(ll):
    """ll[(B,)n] -> fp64 sums [(B,)] with a fixed reduction tree (lqg_sum_trials)."""
    lib = _abi.load()
    if ll.device.type != "cuda":
        raise LqgHipError("sum_trials needs a cuda tensor")
    batched = ll.dim() == 2
    B = ll.shape[0] if batched else 1
    n = ll.shape[-1]
    out = torch.empty((B,), dtype=torch.float64, device=ll.device)
    with torch.cuda.device(ll.device):
        nb = lib.lqg_sum_trials_workspace_bytes(B, n)
        ws = torch.empty(max(int(nb), 8), dtype=torch.uint8, device=ll.device)
        _abi.check(lib.lqg_sum_trials(_DT[ll.dtype], C.c_void_p(ll.data_ptr()), B, n,
                                      ll.stride(0) if batched else 0, ll.stride(-1), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(ws.data_ptr()), nb,
                                      C.c_void_p(torch.cuda.current_stream(ll.device).cuda_stream)), "lqg_sum_trials")
    return out if batched else out[0]


def gaussian_logprob(value, mu, Sigma, k):
    """value, mu [(B,)n,T,>=k]; Sigma [(B,)T,m,m] -> [(B,)n] (lqg_gaussian_logprob)."""
    lib = _abi.load()
    if value.device.type != "cuda":
        raise LqgHipError("gaussian_logprob needs cuda tensors")
    batched = mu.dim() == 4
    B = mu.shape[0] if batched else 1
    n, T = mu.shape[-3], mu.shape[-2]
    value = value.to(dtype=mu.dtype, device=mu.device)
    vb = value.dim() == 4
    out = torch.empty(((B,) if batched else ()) + (n,), dtype=mu.dtype, device=mu.device)
    sview = _abi.mat_view(Sigma.data_ptr(), Sigma.shape, _es(Sigma), Sigma.dim() == 4, True, False)
    with torch.cuda.device(mu.device):
        _abi.check(lib.lqg_gaussian_logprob(
            _DT[mu.dtype], k, T, B, n, _abi.traj_view(value.data_ptr(), value.shape, _es(value), vb),
            _abi.traj_view(mu.data_ptr(), mu.shape, _es(mu), batched), sview, C.c_void_p(out.data_ptr()),
            n if batched else 0, 1, C.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream)),
            "lqg_gaussian_logprob")
    return out


def simulate(actor, dynamics, L, l, K, eps_noise=None, eta_noise=None, x0=None, xhat0=None, return_all=True, seed=None,
             n=None):
    """eps_noise[(B,)n,T,x], eta_noise[(B,)n,T,y] -> x[(B,)n,T+1,x] (and xhat, y, u); or, with seed (an int) and n
    instead of the draws, the same recursion with the normals drawn in-kernel (lqg_simulate_rng: counter-based Philox)."""
    rng = eps_noise is None and eta_noise is None
    if rng and (seed is None or n is None):
        raise LqgHipError("simulate: pass the draws (eps_noise, eta_noise) or seed and n")
    n = n if rng else eps_noise.shape[-3]
    ln = Launch(actor, dynamics, n_trials=n)
    lib = ln.require_gpu(_abi.FAM_SIMULATE)         # k_simulate is instantiated per (x, b, u, y) only
    dm = ln.dims
    xs = ln.empty(n, ln.T + 1, dm["x"])
    xh = ln.empty(n, ln.T + 1, dm["b"]) if return_all else None
    ys = ln.empty(n, ln.T, dm["y"]) if return_all else None
    us = ln.empty(n, ln.T, dm["u"]) if return_all else None

    def init(v):
        if v is None:
            return _abi.NULL_VIEW, None
        v = torch.as_tensor(v, dtype=ln.dtype, device=ln.device).contiguous()
        if v.dim() == 1:
            return _abi.View(v.data_ptr(), 0, 0, 1, 0), v
        return _abi.View(v.data_ptr(), v.stride(0), 0, v.stride(1), 0), v

    v0, k0 = init(x0)
    vh, kh = init(xhat0)
    nt = _abi.NULL_TRAJ
    lview = _abi.NULL_VIEW if (l is None or _is_zero(l)) else _abi.mat_view(l.data_ptr(), l.shape, _es(l),
                                                                           l.dim() == 3, True, True)
    gview = lambda t: _abi.mat_view(t.data_ptr(), t.shape, _es(t), t.dim() == 4, True, False)
    outs = (ln.traj(xs), ln.traj(xh) if return_all else nt, ln.traj(ys) if return_all else nt,
            ln.traj(us) if return_all else nt)
    with torch.cuda.device(ln.device):
        if rng:
            _abi.check(lib.lqg_simulate_rng(C.byref(ln.p), gview(L), lview, gview(K), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                            v0, vh, *outs, ln.stream()), "lqg_simulate_rng")
        else:
            eb = eps_noise.dim() == 4
            _abi.check(lib.lqg_simulate(
                C.byref(ln.p), gview(L), lview, gview(K), ln.traj(eps_noise, eb), ln.traj(eta_noise, eb), v0, vh,
                *outs, ln.stream()), "lqg_simulate")
    return xs, xh, ys, us
