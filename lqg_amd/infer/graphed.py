"""One objective evaluation as ONE hipGraph.

The inner loop of the reference's inference drivers (lqg/infer/mle.py:17-23, the NUTS transitions of
lqg/infer/utils.py:18,37-39) evaluates  theta -> sum_n log p(x_n | theta)  (and its gradient) thousands of times on the SAME
data with the SAME model class.  After round 2's time-parallel kernels one such evaluation is ~0.25 ms of GPU work under
~0.75 ms of Python: ~30 small torch kernels that build the model matrices from the parameters, the host-side decisions of
LogLikelihoodPlan, and ~35 kernel launches.  None of that depends on the parameter VALUES, so it is recorded once:

    parameters (a static device tensor)  ->  model constructor  ->  C-ABI log-likelihood entry  ->  lqg_sum_trials
                                                                                              ->  central differences

is captured into a hipGraph (torch.cuda.CUDAGraph; the ctypes launches of liblqg_hip.so go to the capturing stream like any
other kernel) and an evaluation becomes: copy the parameters in, replay, read the result.

What makes the capture legal: the zoo constructors are sync-free and copy nothing from the host once their constants are
cached (lqg_amd/tracking/_build.py); every decision that needs host values — which library, scan path or lane kernels, the
pattern library — is taken in an EAGER warm-up evaluation of the same shapes and frozen (the time-parallel scans are taken
for up to six times (fp64; twice in fp32) the systems of the eager rule: their host-side checks are paid once here).
Frozen decisions must hold for EVERY parameter vector a replay will see, so (i) everything read off spec VALUES — the
sparsity pattern of a non-zoo class, its decoupling — is derived from >= 8 random positive probe vectors OR-ed together,
never from one (degenerate) point such as theta = 1 where A = a I, a (theta - 1) term or coinciding subjective dynamics
would narrow it; (ii) the value-dependent PRECONDITIONS of the frozen path (eigenvalue floor of lqr.py:27-28 provably
inactive — decoupling and the scans rest on it —, conditioning of the observed noise block for the scans) are
re-evaluated ON THE DEVICE inside the graph (Gershgorin bounds, elementwise ops) and a violated one turns the result
into NaN: consumers that read the result re-evaluate on the eager path, nothing is ever silently wrong.  PointMassBoundedActor's
discretisation runs as one kernel (csrc/lqg_setup.hip) instead of host-synchronising torch.linalg calls.  What is not
captured: a user model whose constructor synchronises or copies from the host, a model that decouples into DIFFERENT
components (identical ones — the dim = 2 tracking models — are merged as trials of one component, through the measured
affine map), an initialised process group (the all-reduce stays outside).  `make()` returns None in those cases and the callers keep the eager path.  LQG_GRAPH=0 disables.
"""
import ctypes as C
import warnings

import torch

from lqg_amd import _abi, _hip, options
from lqg_amd.infer.models import get_model_params


# ---- parameters -> spec matrices as ONE affine map -------------------------------------------------------------------------
# The constructors of the tracking models place their parameters (and fixed constants) into the spec matrices: every entry
# is an affine function of the parameter vector (V = diag(process_noise, action_variability), W = diag(sigma_target,
# sigma_cursor), R = action_cost dt, ...).  Inside the captured graph the ~30 small kernels of such a constructor are
# replaced by one addmm: flat[C, F] = base[F] + theta[C, P] D[P, F], the fields being views of `flat`.  Nothing is assumed:
# base and D are measured by probing the real constructor, and the map is used only if two further random probes reproduce
# the constructor to 1e-12 (PointMassBoundedActor's Cholesky factor is not affine in action_variability: it keeps its
# constructor, which is one kernel anyway).
_TIME_FIELDS = {"Q": 3, "P": 3, "R": 3, "A": 3, "B": 3, "V": 3, "F": 3, "W": 3, "q": 2, "r": 2}    # ndim without candidates
_FLAT_FIELDS = {"Qf": 2, "qf": 1}


def _flatten_spec(spec, C):
    """-> (layout [(field, shape without time / candidate axes, structural-zero tag)], flat [C, F]); None when a field
    varies in time."""
    layout, pieces = [], []
    for f in spec._fields:
        t = getattr(spec, f)
        nd = _TIME_FIELDS.get(f, _FLAT_FIELDS.get(f))
        has_c = t.dim() == nd + 1
        if f in _TIME_FIELDS:
            td = 1 if has_c else 0
            if t.shape[td] > 1 and t.stride(td) != 0:
                return None
            t = t.select(td, 0)
        if not has_c:
            t = t.expand(C, *t.shape)
        layout.append((f, tuple(t.shape[1:]), bool(getattr(getattr(spec, f), "_lqg_zero", False))))
        pieces.append(t.reshape(C, -1))
    return layout, torch.cat(pieces, dim=1)


def _unflatten_spec(cls, layout, flat, T):
    C, out, o = flat.shape[0], {}, 0
    for f, shape, is_zero in layout:
        n = 1
        for s_ in shape:
            n *= s_
        t = flat[:, o:o + n].view(C, *shape)
        o += n
        t = t.unsqueeze(1).expand(C, T, *shape) if f in _TIME_FIELDS else t
        if is_zero:                      # (the structural-zero tag of time_stack_spec: the ABI gets a NULL view)
            t._lqg_zero = True
        out[f] = t
    return cls(**out)


class _Specs:
    """What the log-likelihood entry needs of a model: the two specs."""

    def __init__(self, actor, dynamics):
        self.actor, self.dynamics = actor, dynamics


_deferred = []          # graphs whose owner was finalised during another capture (GraphedLogLik.__del__)


def _drain_deferred():
    """Release parked graphs: outside any capture, after the device has finished with them."""
    if _deferred and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        _deferred.clear()


class GraphedLogLik:
    """obj[c] = sum_n log p(x_n | theta[c]) for a FIXED number of parameter vectors, fixed data, fixed model class."""

    def __init__(self, x, model_type, names, n_cand, fixed=None, process_noise=1.0, dt=1.0 / 60, eps=1e-8):
        if not x.is_cuda:
            raise _abi.LqgHipError("GraphedLogLik needs cuda data")
        self.x = x.contiguous()
        self.model_type, self.names, self.fixed = model_type, list(names), dict(fixed or {})
        self.pn, self.dt, self.eps = process_noise, dt, eps
        self.n, self.rows, self.d = self.x.shape
        self.C, self.P = int(n_cand), len(self.names)
        self.theta = torch.ones(self.C, self.P, dtype=self.x.dtype, device=self.x.device)      # static input
        self.out = None                                                                        # static output, fp64 [C]
        self.graph = None
        self._keep = None
        self._pins = None                # cached constructor constants the graph reads by address (tracking/_build.py)
        self._affine = None
        self._merged_cols = None
        self._guarded = False

    # ---- the captured region -------------------------------------------------------------------------------------
    def _probe_affine(self):
        """Measure base / D of the affine map parameters -> flattened specs on the real constructor (eager); None when the
        constructor is not affine in the parameters or a field varies in time."""
        P, dev, dt_ = self.P, self.x.device, self.x.dtype
        g = torch.Generator().manual_seed(1234)
        th0 = torch.exp(torch.rand(P, generator=g, dtype=torch.float64) * 2.0 - 1.0)
        rows = [th0] + [th0 + 0.5 * th0[i] * torch.eye(P, dtype=torch.float64)[i] for i in range(P)]
        rows += [torch.exp(torch.rand(P, generator=g, dtype=torch.float64) * 3.0 - 1.5) for _ in range(2)]
        th = torch.stack(rows).to(device=dev, dtype=dt_)                              # [1 + P + 2, P]
        m = self._construct(th)
        if self._merged_cols is not None:                    # the composite map parameters -> COMPONENT specs
            m, _ = self._component(m)
            if m is None:
                return None
        fa, fd = _flatten_spec(m.actor, th.shape[0]), _flatten_spec(m.dynamics, th.shape[0])
        if fa is None or fd is None:
            return None
        flat = torch.cat([fa[1], fd[1]], dim=1).to(torch.float64)
        thd = th.to(torch.float64)
        D = (flat[1:P + 1] - flat[:1]) / (thd[1:P + 1] - thd[:1]).diagonal()[:, None]   # [P, F]
        base = flat[0] - thd[0] @ D
        pred = base + thd[P + 1:] @ D
        err = float((pred - flat[P + 1:]).abs().max() / (1.0 + flat[P + 1:].abs().max()))
        if not err < 1e-12:
            return None
        return dict(layout_a=fa[0], layout_d=fd[0], n_a=fa[1].shape[1], base=base.to(dt_).contiguous(), D=D.to(dt_).contiguous(),
                    base64=base.contiguous(), D64=D.contiguous(), actor_cls=type(m.actor), dyn_cls=type(m.dynamics))

    def _model(self, theta):
        af = self._affine
        if af is not None:                                   # one addmm; the fields are views of its result
            flat = torch.addmm(af["base"].unsqueeze(0), theta, af["D"])
            T = self.rows - 1
            return _Specs(_unflatten_spec(af["actor_cls"], af["layout_a"], flat[:, :af["n_a"]], T),
                          _unflatten_spec(af["dyn_cls"], af["layout_d"], flat[:, af["n_a"]:], T))
        return self._construct(theta)

    def _construct(self, theta):
        kw = dict(get_model_params(self.model_type))
        kw.update(self.fixed)
        kw.update({k: theta[:, i] for i, k in enumerate(self.names)})
        return self.model_type(process_noise=self.pn, dt=self.dt, T=self.rows - 1, device=self.x.device,
                               dtype=self.x.dtype, **kw)

    def _component(self, model):
        """The system the kernels solve: the model itself, or — when it decouples into components with identical specs (every
        dim = 2 tracking model: two copies of one 1-D model) — ONE component, the others' data columns becoming trials of
        it (what LogLikelihoodPlan's merge does).  None: several DIFFERENT components (the plan's multi-launch path)."""
        parts = model.decoupled(self.d_full, None, eps=self.eps)
        if not parts:
            return model, None
        from lqg_amd import decouple
        if len(decouple.identical_groups(model, self.d_full, parts, None)) != 1:
            return None, None
        return parts[0][0], [p_[1] for p_ in parts]

    def _probe_theta(self, n):
        """n random positive parameter vectors, log-uniform in [e^-1.5, e^1.5] (seeded): what structure is read off."""
        g = torch.Generator().manual_seed(4321)
        th = torch.exp(torch.rand(n, self.P, generator=g, dtype=torch.float64) * 3.0 - 1.5)
        return th.to(device=self.x.device, dtype=self.x.dtype)

    def _guards(self, model):
        """Device-side restatement of the host checks the frozen decisions rest on (decouple.floor_provably_inactive:
        lambda_min(R) >= eps, Q, Qf >= 0; plan._observed_noise_cond <= SCAN_MAX_COND for the scans) with Gershgorin
        bounds — no synchronisation, capturable.  -> 0-dim bool tensor (True: all hold).  One kernel of the library
        (lqg_precondition_flags); the torch restatement below serves libraries without that entry."""
        from lqg_amd import plan
        a = model.actor
        lib = _abi.load()
        if hasattr(lib, "lqg_precondition_flags"):
            return self._guard_flag(model)[0] != 0

        def bounds(M):                               # [C, k, k] -> Gershgorin (lower, upper) per candidate
            M = 0.5 * (M + M.transpose(-1, -2))
            diag = torch.diagonal(M, dim1=-2, dim2=-1)
            off = M.abs().sum(-1) - diag.abs()
            return (diag - off).amin(-1), (diag + off).amax(-1)

        first = lambda t: t.select(-3, 0)
        ok = (bounds(first(a.R))[0] >= self.eps) & (bounds(first(a.Q))[0] >= -1e-12) & (bounds(a.Qf)[0] >= -1e-12)
        if self.use_scan:
            V = first(model.dynamics.V)[..., :self.d, :]
            lo, hi = bounds(V @ V.transpose(-1, -2))
            ok = ok & (lo > 0) & (hi <= options.get("SCAN_MAX_COND") * lo)
        return ok.all()

    def _guard_flag(self, model):
        """int32 [1] on the device: 1 when the preconditions of the frozen path hold (lqg_precondition_flags)."""
        from lqg_amd import plan
        lib = _abi.load()
        ln = _hip.Launch(model.actor, model.dynamics, d=self.d, n_trials=1, eps=self.eps)
        flag = torch.empty(1, dtype=torch.int32, device=self.x.device)
        _abi.check(lib.lqg_precondition_flags(C.byref(ln.p), float(options.get("SCAN_MAX_COND")), 1 if self.use_scan else 0,
                                              C.c_void_p(flag.data_ptr()), ln.stream()), "lqg_precondition_flags")
        self._guard_keep = (ln, flag)
        return flag

    def _decide(self):
        """Eager, once: everything that needs host values.  Structure (decoupling, sparsity pattern) is read off a model
        built from >= 8 random positive probe vectors; rules that count systems see the launch-shaped model (C vectors)."""
        from lqg_amd import plan
        self.d_full, self.n_full, self.x_full = self.d, self.n, self.x
        probes = self._probe_theta(max(self.C, 8))
        model_s = self._construct(probes)                    # structure
        model = model_s if probes.shape[0] == self.C else self._construct(probes[:self.C])
        sub_s, cols = self._component(model_s)
        if sub_s is None:
            return False                     # several distinct components: LogLikelihoodPlan's launches are the better path
        if model is not model_s:
            sub, cols_c = self._component(model)
            if sub is None or cols_c != cols:
                return False                 # (the subset of probes decouples differently: no frozen structure to trust)
        else:
            sub = sub_s
        self._merged_cols = cols
        if cols is not None:                 # the components' columns as trials of one component system
            self.x = plan._trial_stack(self.x_full, cols)
            self.n, self.d = self.x.shape[0], self.x.shape[-1]
            model, model_s = sub, sub_s
        ln = _hip.Launch(model.actor, model.dynamics, d=self.d, n_trials=self.n, eps=self.eps)
        lib = ln.require_gpu()
        main = _abi.load()
        self.use_scan = lib is main and plan.scan_eligible(main, ln, model, self.eps,
                                                           systems_scale=6 if self.x.dtype == torch.float64 else 2)
        spl = _hip.specialised_library(ln, model_s, self.d, check_strategy=not self.use_scan)
        self.sp_lib = spl
        # the frozen path rests on value-dependent preconditions when it decouples or scans: guard them inside the graph
        self._guarded = bool(self.use_scan or cols is not None)
        if self._guarded and not bool(self._guards(model)):
            # the host checks passed (exact eigenvalues) where the device bounds are inconclusive: nothing to replay safely
            if cols is not None:
                return False
            self.use_scan, self._guarded = False, False
        # The affine shortcut is offered to constructors KNOWN to be affine (the tracking models; a user class opts in with
        # `_lqg_affine_constructor = True`) and still verified numerically — two random probes cannot prove that an
        # arbitrary constructor is affine everywhere.
        import lqg_amd
        known = self.model_type in (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor,
                                    lqg_amd.SubjectiveActor) or getattr(self.model_type, "_lqg_affine_constructor", False)
        if known and (options.flag("GRAPH_AFFINE") or cols is not None):
            self._affine = self._probe_affine()
        if cols is not None and self._affine is None:
            return False                     # (decoupling inside the graph is only available through the measured affine map)
        return True

    def _loglik(self, model, poison=True):
        ln = _hip.Launch(model.actor, model.dynamics, d=self.d, n_trials=self.n, eps=self.eps)
        lib = ln.require_gpu()
        ll = ln.empty(self.n)
        traj = ln.traj(self.x, False)
        ll_sb = self.n if ln.batched else 0
        if self.use_scan:
            nbytes = lib.lqg_scan_workspace_bytes(C.byref(ln.p))
            fn = C.cast(self.sp_lib.lqg_trial_sweep_sp, C.c_void_p) if (self.sp_lib is not None and self.n > 2) else C.c_void_p(None)
            entry = lambda *a: lib.lqg_log_likelihood_scan_with(*a, fn)
        else:
            sp = self.sp_lib.lqg_log_likelihood_sp if self.sp_lib is not None else None
            if sp is not None and self.n == 2:            # (the specialised library sweeps two trials in-lane)
                ln.p.n_trials = 1
                nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
                ln.p.n_trials = 2
            else:
                nbytes = lib.lqg_workspace_bytes(C.byref(ln.p), _abi.OP_LOG_LIKELIHOOD)
            entry = sp or lib.lqg_log_likelihood
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=ln.device)
        args = (C.byref(ln.p), traj, C.c_void_p(ll.data_ptr()), ll_sb, 1, C.c_void_p(ws.data_ptr()), nbytes, ln.stream())
        rc = entry(*args)
        if rc != 0 and not self.use_scan and entry is not lib.lqg_log_likelihood:
            rc = lib.lqg_log_likelihood(*args)            # the specialised library refused (checked in the eager warm-up too)
        _abi.check(rc, "lqg_log_likelihood (graphed)")
        self._keep = (model, ln, ws, ll)
        obj = _hip.sum_trials(ll)                         # fp64 [C]
        if self._guarded and poison:                      # a violated precondition of the frozen path poisons the result
            obj = torch.where(self._guards(model), obj, torch.full_like(obj, float("nan")))
        return obj

    def _forward(self):
        return self._loglik(self._model(self.theta))

    # ---- capture / replay ----------------------------------------------------------------------------------------
    def capture(self):
        dev = self.x.device
        with torch.cuda.device(dev):
            with torch.no_grad():
                if not self._decide():
                    return False
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):             # warm-up on a side stream (allocator, constant caches, library loads)
                    for _ in range(2):
                        self._forward()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                # No finalizer may run inside the capture: a garbage GraphedLogLik (or anything else whose __del__ synchronises the
                # device) collected by the cyclic GC in the middle of it is an illegal call on a capturing stream, and the HIP
                # runtime aborts the process (seen in round 6: `Fatal Python error: Aborted ... Garbage-collecting`, in whichever
                # test happened to capture when the collector fired).  Collect now, keep the collector off until the capture ends.
                import gc
                _drain_deferred()
                gc.collect()
                was_enabled = gc.isenabled()
                gc.disable()
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.out = self._forward()
                finally:
                    if was_enabled:
                        gc.enable()
                self.graph = g
                from lqg_amd.tracking import _build
                self._pins = _build.cached_tensors()      # constants the captured constructor reads by address
        return True

    def release(self):
        """Drop the graph and everything it owns, after the device has finished with it."""
        if self.graph is not None:
            torch.cuda.synchronize(self.x.device)
        self.graph, self.out, self._keep, self._pins = None, None, None, None

    def __del__(self):
        try:
            if self.graph is not None and torch.cuda.is_current_stream_capturing():
                # collected while ANOTHER graph is being captured (reference counting can do that too): synchronising now would
                # break that capture — park what the graph owns until the next capture / release drains it
                _deferred.append((self.graph, self.out, self._keep, self._pins))
                self.graph, self.out, self._keep, self._pins = None, None, None, None
                return
            self.release()
        except Exception:
            pass

    def __call__(self, theta):
        """theta [C, P] (any device / float dtype) -> fp64 [C] on the device (valid until the next call)."""
        self.theta.copy_(theta, non_blocking=True)
        self.graph.replay()
        return self.out


class GraphedFiniteDifference(GraphedLogLik):
    """z [C, P] (log-parameters) -> [C, 1 + P]: objective and its central-difference gradient w.r.t. z, in one graph
    (the 2P+1 perturbed parameter vectors of every z are candidates of ONE launch)."""

    def __init__(self, x, model_type, names, n_points, h=1e-4, extra=None, **kw):
        P = len(names)
        super().__init__(x, model_type, names, n_points * (2 * P + 1), **kw)
        self.K, self.h = int(n_points), float(h)
        self.extra = extra               # optional z [K, P] -> (value [K], gradient [K, P]) added to the result (prior terms)
        self.z = torch.zeros(self.K, P, dtype=torch.float64, device=x.device)                  # static input
        self._eye = self.h * torch.eye(P, dtype=torch.float64, device=x.device)

    def _forward(self):
        K, P = self.K, self.P
        z = self.z
        lib = _abi.load()
        if self._affine is not None and self.extra is None and hasattr(lib, "lqg_fd_candidates"):
            # two kernels of the library around the sweep instead of a dozen of torch's (each ~4.7 us inside the graph):
            # z -> the flattened specs of all K (2 P + 1) candidates; their objectives (+ the guard flag) -> value and gradient
            af = self._affine
            F = af["base"].shape[0]
            flat = torch.empty(K * (2 * P + 1), F, dtype=self.x.dtype, device=self.x.device)
            st = C.c_void_p(torch.cuda.current_stream(self.x.device).cuda_stream)
            _abi.check(lib.lqg_fd_candidates(C.c_void_p(z.data_ptr()), C.c_void_p(af["base64"].data_ptr()),
                                             C.c_void_p(af["D64"].data_ptr()), C.c_void_p(flat.data_ptr()),
                                             _abi.F64 if self.x.dtype == torch.float64 else _abi.F32, K, P, F, self.h, st),
                       "lqg_fd_candidates")
            T = self.rows - 1
            model = _Specs(_unflatten_spec(af["actor_cls"], af["layout_a"], flat[:, :af["n_a"]], T),
                           _unflatten_spec(af["dyn_cls"], af["layout_d"], flat[:, af["n_a"]:], T))
            obj = self._loglik(model, poison=False)
            flag = self._guard_flag(model) if self._guarded else None
            out = torch.empty(K, 1 + P, dtype=torch.float64, device=self.x.device)
            _abi.check(lib.lqg_fd_combine(C.c_void_p(obj.data_ptr()), C.c_void_p(flag.data_ptr()) if flag is not None else None,
                                          C.c_void_p(out.data_ptr()), K, P, self.h, st), "lqg_fd_combine")
            self._fd_keep = (flat, obj, flag)
            return out
        Z = torch.cat([z[:, None, :], z[:, None, :] + self._eye, z[:, None, :] - self._eye], dim=1).reshape(K * (2 * P + 1), P)
        f = self._loglik(self._model(torch.exp(Z).to(self.x.dtype))).reshape(K, 2 * P + 1)
        out = torch.cat([f[:, :1], (f[:, 1:P + 1] - f[:, P + 1:]) / (2 * self.h)], dim=1)
        if self.extra is not None:
            ev, eg = self.extra(z)
            out = out + torch.cat([ev[:, None], eg], dim=1)
        return out

    def __call__(self, z):
        self.z.copy_(z, non_blocking=True)
        self.graph.replay()
        return self.out


_warned = set()


def _sharded():
    """True under an initialised process group of more than one rank: the objective is then all-reduced over the ranks'
    trial shards (lqg_amd.dist), which stays outside any graph."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def make(cls, x, model_type, names, n, group=None, **kw):
    """A captured evaluator, or None when this evaluation cannot be captured (the caller keeps its eager path)."""
    if not options.flag("GRAPH") or group is not None or not x.is_cuda or _sharded():
        return None
    try:
        ev = cls(x, model_type, names, n, **kw)
        return ev if ev.capture() else None
    except Exception as e:                       # a constructor that synchronises, an allocator / capture restriction, ...
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        key = (getattr(model_type, "__name__", str(model_type)), type(e).__name__)
        if key not in _warned:
            _warned.add(key)
            warnings.warn(f"lqg_amd: {key[0]} objective not captured into a hipGraph ({type(e).__name__}: {str(e)[:160]}); "
                          "using the eager path")
        return None
