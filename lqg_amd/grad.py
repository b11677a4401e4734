"""Reverse-mode gradients of the trajectory log-likelihood — the torch.autograd face of the HIP adjoint sweep
(lqg_amd/csrc/lqg_adjoint.hpp, C ABI lqg_log_likelihood_grad).

What the reference gets from jax.grad / jax.value_and_grad of `System.log_likelihood` through all three scans
(lqg/optim.py:142-147, lqg/infer/utils.py:18,37-39, lqg/infer/mle.py:17-23, notebooks/Tutorial.ipynb cell 40):

    sigma = torch.tensor(6., device="cuda", requires_grad=True)
    ll = lqg_amd.BoundedActor(T=500, sigma_target=sigma).log_likelihood(x).sum()
    ll.backward()                      # sigma.grad == d ll / d sigma

`System.log_likelihood` routes here whenever a spec tensor (or Sigma0) requires grad.  forward() runs the first two
sweeps (Riccati, forward: they keep the per-step state and return the value), backward() the two adjoint sweeps with
the upstream weights, and hands the bars of the spec matrices to autograd, which chains them through the (torch) model constructors to the parameters.  Systems
that decouple (every dim=2 model) are differentiated component by component — the gathers of lqg_amd/decouple.py are
differentiable.  Time-invariant specs get one bar per matrix, time-varying specs one per step; no CPU fallback.
"""
import ctypes as C

import torch

from lqg_amd import _abi, _hip

ACTOR_FIELDS = ("A", "B", "F", "V", "W", "Q", "R")           # time-stacked actor fields that get a gradient
DYN_FIELDS = ("A", "B", "F", "V", "W")


def needs_grad(system, Sigma0=None):
    if not torch.is_grad_enabled():
        return False
    ts = [getattr(system.actor, f) for f in ACTOR_FIELDS + ("Qf",)] + [getattr(system.dynamics, f) for f in DYN_FIELDS]
    if Sigma0 is not None:
        ts.append(Sigma0)
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def _varies(t):
    return t.shape[-3] > 1 and t.stride(-3) != 0


def _time_slice(t, name):
    """[.., T, r, c] time-invariant field -> its [.., r, c] matrix, staying on the autograd graph."""
    base = getattr(t, "_lqg_base", None)
    if base is not None and base.shape == t.shape[:-3] + t.shape[-2:]:
        return base                     # the matrix time_stack expanded: avoids a [.., T, r, c] zero-fill in backward
    return t.select(-3, 0)


def _layout(dm):
    x, b, u, y = dm["x"], dm["b"], dm["u"], dm["y"]
    order = [("dA", x, x), ("dB", x, u), ("dF", y, x), ("dVV", x, x), ("dWW", y, y), ("aA", b, b), ("aB", b, u),
             ("aF", y, b), ("aVV", b, b), ("aWW", y, y), ("aQ", b, b), ("aR", u, u), ("aQf", b, b), ("aS0", b, b),
             ("aA2", b, b), ("aB2", b, u)]
    off, out = 0, {}
    for name, r, c in order:
        out[name] = (off, r, c)
        off += r * c
    return out, off


def _promote64(spec):
    """fp64 image of a spec that keeps stride-0 time axes stride-0 (a time-invariant field stays ONE matrix)."""
    def up(t):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
            return t
        if t.dim() >= 3 and t.shape[-3] > 1 and t.stride(-3) == 0:
            return t.narrow(-3, 0, 1).double().expand_as(t)
        return t.double()
    return spec._replace(**{f: up(getattr(spec, f)) for f in spec._fields})


class Sweep:
    """One adjoint evaluation in two phases sharing the kept forward state: `forward()` (Riccati + forward sweep, returns
    the log-likelihood) and `reverse(g)` (the two adjoint sweeps with upstream weights g, returns the bars).  The
    workspace lives as long as this object (an autograd node keeps it between forward and backward)."""

    def __init__(self, actor, dynamics, x, Sigma0=None, eps=1e-8, system=None):
        """system: the System that owns the two specs — with it (time-invariant specs, no affine cost terms) the sweep runs on
        the structure-specialised adjoint library of the specs' sparsity pattern (round 5: csrc/lqg_adjoint_sp.hpp — matrix
        adjoints once per SYSTEM, trials reduced to per-step sums, bars returned summed over the trials)."""
        d = x.shape[-1]
        self.sp = None
        self.out_dtype = None

        def promoted():
            return (_promote64(actor), _promote64(dynamics), x.double(), Sigma0.double() if Sigma0 is not None else None)

        # ---- the structure-specialised adjoint library first (no on-demand compile of lane kernels it would not use).  fp32
        # stays fp32 there unless the observed noise block is ill-conditioned (plan.f32_needs_wide: the rule of the forward path;
        # the point mass seen in full, cond ~ 5e8): then the sweeps run over an fp64 image, value and bars rounded once
        if system is not None and actor.A.is_cuda:
            ln = _hip.Launch(actor, dynamics, d=d, n_trials=x.shape[-3], Sigma0=Sigma0, eps=eps)
            self.sp = _specialised_adjoint(ln, system, d)
            if self.sp is not None and not _trial_offsets_fit(x, x.shape[-3]):
                self.sp = None
            if self.sp is not None:
                from lqg_amd import plan as _plan
                if actor.A.dtype == torch.float32 and _plan.f32_needs_wide(system, d):
                    self.out_dtype = torch.float32
                    actor, dynamics, x, Sigma0 = promoted()
                    ln = _hip.Launch(actor, dynamics, d=d, n_trials=x.shape[-3], Sigma0=Sigma0, eps=eps)
                self.ln = ln
                self.x, self.xb = _hip._prep_x(ln, x)
                self.N = self.x.shape[-3]
                self.per_sys, self.lanes = 1, ln.B
                self.ld = (self.lanes + 63) // 64 * 64
                self.lay, self.total = _layout(ln.dims)
                self.slabs = 1
                self.nbytes = int(self.sp.lqg_grad_workspace_bytes_sp(C.byref(ln.p)))
                if self.nbytes > 0:
                    self.ws, self.transient = None, False
                    self._workspace()
                    self.fresh = False
                    return
                self.sp = None
        # ---- round-1 lane kernels / cooperative sweep.  fp32 with EVERY state observed (the point-mass model seen in full:
        # cond(Sigma_oo) ~ 5e8): that reverse sweep conditions through the explicit S_oo^-1, which fp32 cannot carry -- same policy
        # as the mixed forward problem LQG_F32_SYS64 (include/lqg_hip.h): fp64 image, value and bars rounded to fp32 once
        if actor.A.dtype == torch.float32 and d == dynamics.A.shape[-1]:
            self.out_dtype = torch.float32
            actor, dynamics, x, Sigma0 = promoted()
        ln = _hip.Launch(actor, dynamics, d=d, n_trials=x.shape[-3], Sigma0=Sigma0, eps=eps)
        self.lib = ln.require_gpu(_abi.FAM_ADJOINT)     # lane kernels (an unlisted small shape is compiled on first use) or,
        # for every other shape (x + b > 12: the delay models), the cooperative sweep of the main library — fp64 only: an fp32
        # caller gets the fp64 image of the sweeps, value and bars rounded once (the policy of LQG_F32_SYS64)
        if not self.lib.lqg_grad_supported(ln.p.dtype, C.byref(ln.p.dims)) and actor.A.dtype == torch.float32 \
                and self.lib.lqg_grad_supported(_abi.F64, C.byref(ln.p.dims)):
            self.out_dtype = torch.float32
            actor, dynamics, x, Sigma0 = promoted()
            ln = _hip.Launch(actor, dynamics, d=d, n_trials=x.shape[-3], Sigma0=Sigma0, eps=eps)
        if not self.lib.lqg_grad_supported(ln.p.dtype, C.byref(ln.p.dims)):
            raise _abi.LqgHipError(f"no adjoint kernels for model shape {tuple(ln.dims[k] for k in 'xbuyd')} "
                                   "(lane kernels: LQG_ADJOINT_DIMS of lqg_amd/csrc/lqg_dims.def; cooperative sweep: x, b <= 64, "
                                   "u, y, d <= 4); there is no CPU path")
        self.ln = ln
        self.x, self.xb = _hip._prep_x(ln, x)
        self.N = self.x.shape[-3]
        # lanes of the gradient array per system: the trials (lane kernels: the caller sums over them) or 1 (cooperative
        # sweep: bars already summed over the trials) — include/lqg_hip.h: lqg_grad_lanes_per_system
        self.per_sys = int(self.lib.lqg_grad_lanes_per_system(C.byref(ln.p)))
        self.lanes = ln.B * self.per_sys
        self.ld = (self.lanes + 63) // 64 * 64
        self.lay, self.total = _layout(ln.dims)
        assert self.total == self.lib.lqg_grad_elements(C.byref(ln.p.dims))
        self.slabs = int(self.lib.lqg_grad_slabs(C.byref(ln.p)))          # 1 (time-invariant) or T (bars per step)
        self.nbytes = int(self.lib.lqg_grad_workspace_bytes(C.byref(ln.p), self.ld))
        self.ws, self.transient = None, False
        self._workspace()
        self.fresh = False              # True while the workspace holds an unconsumed forward state

    def _workspace(self):
        if self.ws is None:
            self.ws = torch.empty(max(self.nbytes, 256), dtype=torch.uint8, device=self.ln.device)
        return self.ws

    def release(self):
        """Give the workspace back (and with it the forward state: `reverse` then re-runs the forward sweep).  A `transient` sweep
        does so after every forward and every reverse: of the pieces of a candidate-chunked evaluation (`_one`) only the one being
        swept holds a workspace, under autograd too — where all pieces run forward before any runs backward."""
        self.ws, self.fresh = None, False

    def _call(self, phases, g, ll, out):
        ln, N = self.ln, self.N
        ptr = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
        entry = self.sp.lqg_log_likelihood_grad_sp if self.sp is not None else self.lib.lqg_log_likelihood_grad
        with torch.cuda.device(ln.device):
            _abi.check(entry(
                C.byref(ln.p), ln.traj(self.x, self.xb), ptr(g), N if ln.batched else 0, 1, ptr(ll),
                N if ln.batched else 0, 1, ptr(out), self.ld, ptr(self._workspace()), self.nbytes, phases, ln.stream()),
                "lqg_log_likelihood_grad")

    def forward(self):
        ll = self.ln.empty(self.N)
        self._call(1, None, ll, None)
        self.fresh = True
        return ll if self.out_dtype is None else ll.to(self.out_dtype)

    def forward_value_only(self):
        """`forward()` of a transient sweep: the value is kept, the workspace is not."""
        ll = self.forward()
        if self.transient:
            self.release()
        return ll

    def reverse(self, g=None):
        """-> {name: [B, N, r, c]} per-(system, trial) bars (lqg_hip.h: order of the gradient elements); with
        time-varying specs [B, N, T, r, c] (aQf and aS0, which have no time axis, stay [B, N, r, c]).  Shapes served by the
        cooperative sweep return [B, 1, ...]: the bars already summed over the trials."""
        ln = self.ln
        if g is not None:
            g = g.to(dtype=ln.dtype, device=ln.device).expand(ln.lead() + (self.N,)).contiguous()
        out = torch.empty(self.slabs, self.total, self.ld, dtype=ln.dtype, device=ln.device)
        if not self.fresh:              # a second backward (retain_graph): the reverse sweep overwrote L_t with Lbar_t
            self.forward()
        self._call(2, g, None, out)
        self.fresh = False
        if self.transient:
            self.release()
        bars = {}
        for k, (o, r, c) in self.lay.items():
            v = out[:, o:o + r * c, :self.lanes].reshape(self.slabs, r, c, ln.B, self.per_sys).permute(3, 4, 0, 1, 2)
            v = v[:, :, 0] if (self.slabs == 1 or k in ("aQf", "aS0")) else v
            bars[k] = v if self.out_dtype is None else v.to(self.out_dtype)
        return bars


def _specialised_adjoint(ln, system, d):
    """The adjoint library of the system's sparsity pattern (lqg_amd/specialize.py: padj_<key>.so), or None: specs that vary
    in time, affine cost terms, joint dimensions beyond the lane kernels, LQG_ADJOINT_SP=0, LQG_COOP_ADJOINT=1, no compiler."""
    from lqg_amd import options, specialize
    if system is None or not options.flag("ADJOINT_SP") or options.get("COOP_ADJOINT") or ln.m > ADJOINT_SP_MAX_JOINT:
        return None
    p = ln.p
    if ln.T > 1:
        for spec, fields in ((p.actor, ("Q", "R", "A", "B", "V", "F", "W")), (p.dynamics, ("A", "B", "V", "F", "W"))):
            if any(getattr(spec, f).st != 0 for f in fields):
                return None
    if any(getattr(p.actor, f).ptr for f in ("q", "qf", "P", "r")):
        return None
    dims, masks, key, live = specialize.adjoint_pattern(system, d)
    return specialize.load_adjoint_pattern(key, dims, masks, live=live)


def _trial_offsets_fit(x, n_trials):
    """The per-trial reverse sweep addresses a trial's rows by a 32-bit byte offset from a wave-uniform row pointer
    (csrc/lqg_adjoint_trial_sp.hpp): the trials of one system must lie within 2 GiB of each other (packed trajectories always do)."""
    return n_trials <= 2 or abs(x.stride(-3)) * (n_trials - 1) * x.element_size() < (1 << 31)


ADJOINT_SP_MAX_JOINT = 12       # largest x + b the specialised adjoint libraries are generated for (registers: the chunk's states)


class GradPlan:
    """Value + gradient of one (system, data) pair, decided ONCE (decoupling into components, identical components merged as
    trials, adjoint libraries, workspaces, argument structs) so that repeated evaluations are pure launches — the analogue of
    plan.LogLikelihoodPlan for the reverse mode; bench.py keeps one per timed leg.  `run(g)` -> (ll[(B,) n], [bars per
    component]): the bars are those of the components' spec matrices ({name: [B, 1, r, c]}, summed over the trials).

    For the ZOO classes (and their decoupled components) the adjoint masks are a property of the constructor: hoisted products
    that cancel for every parameter value (`F_d B_d - F_a B_a = 0` for a shared spec) are pruned, and bars of fields no parameter
    moves are zeros.  Such bars are exact only CONTRACTED through the constructor to its parameters (what `lqg_amd.grad` /
    `lqg_amd.infer` do); they are not the partial derivatives with respect to each spec matrix taken on its own.  A System built
    from leaf matrices gets structurally full masks for every field that requires grad (specialize.pattern_of(grad_full=True))."""

    def __init__(self, system, x, events=False):
        from lqg_amd import _hipev, decouple
        from lqg_amd.plan import _trial_stack
        import lqg_amd
        d = x.shape[-1]
        with torch.no_grad():
            parts = system.decoupled(d, None, for_grad=True)
            whole = parts is None
            parts = parts or [(system, list(range(d)), None)]
            zoo = (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor, lqg_amd.SubjectiveActor)
            by_class = len(parts) > 1 and type(system) in zoo and getattr(system, "_zoo_structure", None) is not None
            groups = decouple.identical_groups(system, d, parts, None) if by_class else [[i] for i in range(len(parts))]
            self.items = []
            for gr in groups:
                sub, cols, _ = parts[gr[0]]
                # (a system that does not decouple keeps the caller's trajectory layout — e.g. workload.pack_trials' trial-fastest
                # storage; an index copy would re-lay it trial-major: 3.5x slower per-trial sweeps, measured)
                xs = x if whole else (_trial_stack(x, [parts[i][1] for i in gr]) if len(gr) > 1 else x[..., cols].contiguous())
                sw = Sweep(sub.actor, sub.dynamics, xs, system=sub)
                ev = None
                if events:
                    ev = ([_hipev.Event() for _ in range(4)], [_hipev.Event() for _ in range(4)])
                self.items.append(dict(sweep=sw, group=len(gr), ev=ev))
        self.n = x.shape[-3]

    @property
    def description(self):
        sw = self.items[0]["sweep"]
        dims = tuple(sw.ln.dims[k] for k in "xbuyd")
        kind = ("split reverse-mode sweep on the pattern library (k_riccati_sp + k_asp_sys_fwd"
                + (")" if sw.N <= 2 else " + k_trial_sp keeping checkpoints)") + " / (" + ("" if sw.N <= 2 else "k_asp_trial_rev + ")
                + "k_asp_sys_rev + k_asp_ric_rev)") if sw.sp is not None else "round-1 lane kernels (one (system, trial) pair per lane)"
        if len(self.items) > 1 or self.items[0]["group"] > 1:
            kind += (f"; {sum(it['group'] for it in self.items)} decoupled components of dims (x,b,u,y,d)={dims}"
                     + (f", {self.items[0]['group']} identical ones as trials of one system" if self.items[0]["group"] > 1 else ""))
        return kind

    def _set_events(self, it, which):
        if it["ev"] is not None:
            for i in range(4):
                it["sweep"].ln.p.phase_events[i] = it["ev"][which][i].h

    def run(self, g=None):
        ll_tot, bars = None, []
        with torch.no_grad():
            for it in self.items:
                sw, G = it["sweep"], it["group"]
                self._set_events(it, 0)
                ll = sw.forward()
                gg = None
                if g is not None:
                    gg = g.repeat_interleave(1, -1) if G == 1 else torch.cat([g] * G, dim=-1)
                self._set_events(it, 1)
                bars.append(sw.reverse(gg))
                ll = ll.view(*ll.shape[:-1], G, -1).sum(-2) if G > 1 else ll
                ll_tot = ll if ll_tot is None else ll_tot + ll
        return ll_tot, bars

    def phase_ms(self):
        """Kernel milliseconds of the last run, summed over components (events=True): forward (riccati, system, trial) and
        reverse (trial, system, riccati)."""
        out = {"fwd_riccati": 0.0, "fwd_system": 0.0, "fwd_trial": 0.0, "rev_trial": 0.0, "rev_system": 0.0, "rev_riccati": 0.0}
        for it in self.items:
            f, r = it["ev"]
            r[3].synchronize()
            for k, (ev, i) in zip(out, [(f, 0), (f, 1), (f, 2), (r, 0), (r, 1), (r, 2)]):
                out[k] += ev[i].elapsed_ms(ev[i + 1])
        return out


def raw_grad(actor, dynamics, x, g=None, Sigma0=None, eps=1e-8, want_value=True, system=None):
    """Both phases at once.  x[n,T+1,d] or [B,n,T+1,d]; g like the log-likelihood ([n] / [B,n]) or None.
    Returns (ll, {name: [B, N, r, c] per-(system, trial) bars — [B, 1, r, c], summed over the trials, from the specialised
    adjoint libraries and the cooperative sweep}, launch).  With `system` a zoo class (or one of its decoupled components) the bars
    are valid contracted through the constructor only (GradPlan's docstring); pass system=None for the partials of each matrix."""
    sw = Sweep(actor, dynamics, x, Sigma0=Sigma0, eps=eps, system=system)
    ll = sw.forward()
    return ll, sw.reverse(g), sw.ln


class _LogLikelihood(torch.autograd.Function):
    """ll = log_likelihood(system, x); inputs are the time-invariant matrices (slices of the spec fields)."""

    @staticmethod
    def forward(ctx, system, x, Sigma0, *mats):
        ctx.system, ctx.n_mats = system, len(mats)
        ctx.time_varying = mats[0].dim() == system.actor.A.dim()        # full [.., T, r, c] fields were passed
        ctx.save_for_backward(*(mats + ((Sigma0,) if Sigma0 is not None else ())))
        ctx.has_s0 = Sigma0 is not None
        with torch.no_grad():
            ctx.sweep = Sweep(system.actor, system.dynamics, x, Sigma0=Sigma0, system=system)
            ctx.sweep.transient = bool(getattr(system, "_lqg_transient_workspace", False))
            return ctx.sweep.forward_value_only()

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        mats = saved[:ctx.n_mats]
        S0 = saved[ctx.n_mats] if ctx.has_s0 else None
        sys_ = ctx.system
        with torch.no_grad():
            bars = ctx.sweep.reverse(g)
            tot = {k: v.sum(1) for k, v in bars.items()}                    # over trials -> [B, (T,) r, c]
            sym2 = lambda M: M + M.transpose(-1, -2)
            tv = ctx.time_varying
            first = (lambda t: t) if tv else (lambda t: t.select(-3, 0))
            res = {"aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"], "aF": tot["aF"],
                   "aV": sym2(tot["aVV"]) @ first(sys_.actor.V), "aW": sym2(tot["aWW"]) @ first(sys_.actor.W),
                   "aQ": tot["aQ"], "aR": 0.5 * sym2(tot["aR"]), "aQf": tot["aQf"],
                   "dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"],
                   "dV": sym2(tot["dVV"]) @ first(sys_.dynamics.V), "dW": sym2(tot["dWW"]) @ first(sys_.dynamics.W)}
            names = ["a" + f for f in ACTOR_FIELDS] + ["aQf"] + ["d" + f for f in DYN_FIELDS]
            outs = []
            for name, m in zip(names, mats):
                gm = res[name]
                if m.dim() == (3 if (tv and name != "aQf") else 2):          # field shared by all systems
                    gm = gm.sum(0)
                outs.append(gm.to(m.dtype) if ctx.needs_input_grad[3 + len(outs)] else None)
            gS0 = None
            if S0 is not None and ctx.needs_input_grad[2]:
                gS0 = tot["aS0"] if S0.dim() == 3 else tot["aS0"].sum(0)
        return (None, None, gS0) + tuple(outs)


def _candidate_chunks(system, x, Sigma0):
    """Candidate ranges [(lo, hi)] such that the reverse sweep's workspace of each fits plan.ops_workspace_limit — for the shapes
    the cooperative sweep serves (x + b > ADJOINT_SP_MAX_JOINT: it keeps S, L, P, Sigma and every trial's mean per step, ~42 MB per
    system for DelayedSubjectiveActor at T = 500 x 50 trials: 4096 candidates would ask for 172 GB in one piece) — or None."""
    B = system.n_systems
    if not B or B <= 1 or system.xdim + system.bdim <= ADJOINT_SP_MAX_JOINT or not system.actor.A.is_cuda:
        return None
    from lqg_amd import plan as _plan, workload
    with torch.no_grad():
        one = workload.slice_system(system, 0, 1)
        one64 = one.to(torch.float64)
        ln = _hip.Launch(one64.actor, one64.dynamics, d=x.shape[-1], n_trials=x.shape[-3],
                         Sigma0=None if Sigma0 is None else (Sigma0[:1] if Sigma0.dim() == 3 else Sigma0).double())
        lib = ln.require_gpu(_abi.FAM_ADJOINT)
        per = int(lib.lqg_grad_workspace_bytes(C.byref(ln.p), 64))
    limit = _plan.ops_workspace_limit(system.actor.A.device)
    if per <= 0 or per * B <= limit:
        return None
    size = max(1, int(limit // per))
    return [(lo, min(B, lo + size)) for lo in range(0, B, size)]


def _one(system, x, Sigma0):
    chunks = _candidate_chunks(system, x, Sigma0)
    if chunks is not None and len(chunks) > 1:
        from lqg_amd import workload
        outs = []
        B = system.n_systems
        for lo, hi in chunks:                      # (slices are views: the bars flow back to the caller's leaves through autograd)
            S0 = Sigma0 if (Sigma0 is None or Sigma0.dim() == 2) else Sigma0[lo:hi]
            piece = workload.slice_system(system, lo, hi)
            # each piece gives its workspace back after its forward and re-runs the forward sweep inside its backward: under
            # autograd every piece runs forward before any runs backward, and the workspaces must not coexist
            piece._lqg_transient_workspace = True
            xs = x[lo:hi] if (x.dim() == 4 and x.shape[0] == B) else x          # (a [1, n, T+1, d] x is shared, not indexed)
            outs.append(_one_piece(piece, xs, S0))
        return torch.cat(outs, dim=0)
    return _one_piece(system, x, Sigma0)


def _one_piece(system, x, Sigma0):
    a, dy = system.actor, system.dynamics
    if not (getattr(a.P, "_lqg_zero", False) or not a.P.requires_grad):
        raise NotImplementedError("gradient w.r.t. the cross-cost P is not provided")
    if any(_varies(getattr(a, f)) for f in ACTOR_FIELDS) or any(_varies(getattr(dy, f)) for f in DYN_FIELDS):
        # time-varying specs: the sweep returns one bar per step; the fields themselves are the autograd inputs
        mats = [getattr(a, f) for f in ACTOR_FIELDS] + [a.Qf] + [getattr(dy, f) for f in DYN_FIELDS]
    else:
        mats = [_time_slice(getattr(a, f), "actor." + f) for f in ACTOR_FIELDS] + [a.Qf] \
            + [_time_slice(getattr(dy, f), "dynamics." + f) for f in DYN_FIELDS]
    return _LogLikelihood.apply(system, x, Sigma0, *mats)


def log_likelihood(system, x, Sigma0=None):
    """Differentiable log-likelihood: x[n,T+1,d] (or [B,n,T+1,d]) -> [n] (or [B,n])."""
    d = x.shape[-1]
    parts = system.decoupled(d, Sigma0, for_grad=True)
    if parts is None:
        return _one(system, x, Sigma0)
    # Identical axes of a ZOO model are one system observed on several data columns: one sweep with the axes as trials.
    # Only for the exact zoo classes — there every block is the same FUNCTION of the parameters by construction, so
    # routing all of the gradient through the first block's entries is exact; components of an arbitrary System that merely
    # hold equal values may depend on different leaves and are differentiated one by one.
    import lqg_amd
    from lqg_amd import decouple
    from lqg_amd.plan import _trial_stack
    zoo = (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor, lqg_amd.SubjectiveActor)
    by_class = type(system) in zoo and getattr(system, "_zoo_structure", None) is not None and Sigma0 is None
    groups = decouple.identical_groups(system, d, parts, Sigma0) if by_class else [[i] for i in range(len(parts))]
    total = None
    for g in groups:
        sub, cols, bs = parts[g[0]]
        if len(g) > 1:
            xm = _trial_stack(x, [parts[i][1] for i in g])                  # [(B,) G*n, T+1, d_c]
            ll = _one(sub, xm, None)
            ll = ll.view(*ll.shape[:-1], len(g), -1).sum(-2)
        else:
            S0 = None if Sigma0 is None else Sigma0[..., bs, :][..., :, bs]
            ll = _one(sub, x[..., cols].contiguous(), S0)
        total = ll if total is None else total + ll
    return total
