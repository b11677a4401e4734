"""Block decoupling of a System: if the joint (state, belief, control, observation) interaction graph of a model
splits into independent components, the trajectory log-likelihood factorises exactly,
    log p(x) = sum_c log p_c(x[..., cols_c]),
and every recursion of the path (Riccati, Kalman, moment recursion) decouples with it.  Each component is a smaller
LQG problem — cubic work shrinks by (m / m_c)^3 per component.  Every `dim=2` model of the reference's zoo is two
independent 1-D models built with block_diag and then permuted (lqg/tracking/basic.py:21-35,
lqg/tracking/subjective.py:18-44), so this halves... in fact quarters their arithmetic.

The components are found from DATA (which entries of A, B, F, V V^T, W W^T, Q, R, Sigma0 are non-zero for any
candidate), with the class-level probe of lqg_amd/specialize.py for the model zoo.  The eigenvalue floor of
lqr.backward (lqg/control/lqr.py:27-28) acts on the JOINT H: when it is active it couples the components, so a system
is only decoupled when the floor is provably inactive (floor_provably_inactive: lambda_min(R) >= eps, Q, Qf >= 0);
otherwise the joint problem is solved, exactly as the reference does.
"""
import numpy as np
import torch

from lqg_amd.spec import LQGSpec
from lqg_amd.system import System


def _find(parent, i):
    while parent[i] != i:
        parent[i] = parent[parent[i]]
        i = parent[i]
    return i


def _union(parent, i, j):
    ri, rj = _find(parent, i), _find(parent, j)
    if ri != rj:
        parent[rj] = ri


def components_from_masks(dims, m, extra_bb=None):
    """Connected components of the interaction graph.  Nodes: x_i, b_j, u_k, y_l.  Returns a list of dicts
    (xs, bs, us, ys as sorted index lists), or None when the graph is connected / a component is degenerate."""
    nx, nb, nu, ny = dims["x"], dims["b"], dims["u"], dims["y"]
    X, B, U, Y = 0, nx, nx + nb, nx + nb + nu
    parent = list(range(nx + nb + nu + ny))

    def link(mask, ro, co):
        for i, j in zip(*np.nonzero(mask)):
            _union(parent, ro + int(i), co + int(j))

    link(m["Ad"], X, X); link(m["Bd"], X, U); link(m["Fd"], Y, X); link(m["N1"], X, X); link(m["WWd"], Y, Y)
    link(m["Aa"], B, B); link(m["Ba"], B, U); link(m["Fa"], Y, B); link(m["VVa"], B, B); link(m["WWa"], Y, Y)
    link(m["Q"], B, B); link(m["Rr"], U, U)
    if extra_bb is not None:
        link(extra_bb, B, B)
    groups = {}
    for n in range(len(parent)):
        groups.setdefault(_find(parent, n), []).append(n)
    comps = []
    for nodes in groups.values():
        c = dict(xs=[n - X for n in nodes if n < B], bs=[n - B for n in nodes if B <= n < U],
                 us=[n - U for n in nodes if U <= n < Y], ys=[n - Y for n in nodes if n >= Y])
        comps.append(c)
    if len(comps) < 2:
        return None
    d = dims["d"]
    keep = []
    for c in comps:
        obs = [i for i in c["xs"] if i < d]
        if not obs:
            continue                       # a component without observed dims does not enter the likelihood
        if not (c["bs"] and c["us"] and c["ys"]):
            return None                    # degenerate component: keep the joint problem
        c["xs"] = obs + [i for i in c["xs"] if i >= d]      # observed dims lead (the kernels read x[..., :d_c])
        c["cols"] = obs
        keep.append(c)
    if len(keep) < 1:
        return None
    keep.sort(key=lambda c: c["cols"][0])
    return keep


def _take(t, rows, cols=None, time_axis=True):
    """Select rows (and columns) of the last dims of a spec field.  A stride-0 time axis (time-invariant field,
    lqg_amd.utils.time_stack) is preserved: the selection is made on one time slice and re-expanded, so that the
    sub-spec is still recognised as time-invariant by the launch glue."""
    nd = 1 if cols is None else 2
    tax = -(nd + 1)
    if time_axis and t.dim() > nd and t.stride(tax) == 0 and t.shape[tax] > 1:
        T = t.shape[tax]
        base = _take(t.select(tax, 0), rows, cols, time_axis=False)
        out = base.unsqueeze(tax).expand(*base.shape[:tax + 1 + base.dim()][:base.dim() - nd], T, *base.shape[-nd:])
        out._lqg_base = base
        return out
    if cols is None:
        out = torch.index_select(t, -1, _index(rows, t.device))
    else:
        out = torch.index_select(torch.index_select(t, -2, _index(rows, t.device)), -1, _index(cols, t.device))
    if t.dim() == nd + 2 and t.shape[0] > 1 and t.stride(0) == 1 and out.stride(0) != 1:
        # a batched time-varying field stored [T][element][system] (workload.pack_systems: a wave's 64 lanes read 64 consecutive
        # elements per (step, entry)) keeps that storage — index_select returns [system][T][element], 64 cache lines per wave-load
        perm = tuple(range(1, out.dim())) + (0,)
        inv = (out.dim() - 1,) + tuple(range(out.dim() - 1))
        out = out.permute(perm).contiguous().permute(inv)
    return out


_index_cache = {}


def _index(idx, device):
    """Device index tensor of a Python index list, cached (indexing with the list itself builds the index on the host and
    copies it to the device on every call: 64 gathers per decoupling plan, 1.5 ms)."""
    key = (tuple(int(i) for i in idx), str(device))
    t = _index_cache.get(key)
    if t is None:
        if len(_index_cache) > 4096:
            _index_cache.clear()
        t = _index_cache[key] = torch.tensor(list(key[0]), dtype=torch.long, device=device)
    return t


def _noise_cols(mask_rows):
    """Columns of a noise factor V that touch the selected rows."""
    return [int(j) for j in np.nonzero(mask_rows.any(axis=0))[0]]


def split_system(system, comps, vmask_a, wmask_a, vmask_d, wmask_d):
    """Sub-Systems of the components (plain System objects on views / small gathers of the original specs)."""
    from lqg_amd.utils import mark_zero
    a, dy = system.actor, system.dynamics
    out = []
    for c in comps:
        xs, bs, us, ys = c["xs"], c["bs"], c["us"], c["ys"]
        va = _noise_cols(vmask_a[bs]) or [0]
        wa = _noise_cols(wmask_a[ys]) or [0]
        vd = _noise_cols(vmask_d[xs]) or [0]
        wd = _noise_cols(wmask_d[ys]) or [0]

        def zero_like(t, *shape):
            z = torch.zeros((), dtype=t.dtype, device=t.device).expand(*shape)
            return mark_zero(z)

        T = a.A.shape[-3]
        keepz = lambda t, sel: (mark_zero(sel) if getattr(t, "_lqg_zero", False) else sel)
        act = LQGSpec(Q=_take(a.Q, bs, bs), q=keepz(a.q, _take(a.q, bs)), Qf=_take(a.Qf, bs, bs, time_axis=False),
                      qf=keepz(a.qf, _take(a.qf, bs, time_axis=False)), P=keepz(a.P, _take(a.P, us, bs)), R=_take(a.R, us, us),
                      r=keepz(a.r, _take(a.r, us)), A=_take(a.A, bs, bs), B=_take(a.B, bs, us), V=_take(a.V, bs, va),
                      F=_take(a.F, ys, bs), W=_take(a.W, ys, wa))
        dyn = LQGSpec(Q=zero_like(dy.A, T, len(xs), len(xs)), q=zero_like(dy.A, T, len(xs)),
                      Qf=zero_like(dy.A, len(xs), len(xs)), qf=zero_like(dy.A, len(xs)),
                      P=zero_like(dy.A, T, len(us), len(xs)), R=zero_like(dy.A, T, len(us), len(us)),
                      r=zero_like(dy.A, T, len(us)), A=_take(dy.A, xs, xs), B=_take(dy.B, xs, us),
                      V=_take(dy.V, xs, vd), F=_take(dy.F, ys, xs), W=_take(dy.W, ys, wd))
        out.append((System(actor=act, dynamics=dyn), c["cols"], bs))
    return out


_ZOO_GROUPS = {}        # (class, zoo structure, d) -> groups of identical components (a property of the constructor)


def _same_spec(sa, sb):
    """True when two sub-systems have bit-identical specs (values AND time structure)."""
    for which in ("actor", "dynamics"):
        for f in LQGSpec._fields:
            ta, tb = getattr(getattr(sa, which), f), getattr(getattr(sb, which), f)
            za, zb = getattr(ta, "_lqg_zero", False), getattr(tb, "_lqg_zero", False)
            if za and zb:
                continue
            if za != zb or ta.shape != tb.shape:
                return False
            notime = f in ("Qf", "qf")
            if not notime:
                tax = -(2 if f in ("q", "r") else 3)
                ia, ib = ta.stride(tax) == 0 or ta.shape[tax] == 1, tb.stride(tax) == 0 or tb.shape[tax] == 1
                if ia != ib:
                    return False
                if ia:
                    ta, tb = ta.select(tax, 0), tb.select(tax, 0)
            if not torch.equal(ta, tb):
                return False
    return True


def identical_groups(system, d, parts, Sigma0=None):
    """Partition the components into groups with bit-identical specs: such components are the SAME system observed on
    different data columns (every dim=2 model of the zoo is block_diag of one 1-D model with shared parameters), so the
    per-system sweeps are done once and the components become trials (lqg_amd/plan.py).  For the exact zoo classes the
    grouping is a property of the constructor and is cached per class; otherwise it is decided from the data."""
    if parts is None or len(parts) < 2 or Sigma0 is not None:
        return [[i] for i in range(len(parts or []))]
    import lqg_amd
    zoo = (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor, lqg_amd.SubjectiveActor)
    zs = getattr(system, "_zoo_structure", None) if type(system) in zoo else None
    key = (type(system), tuple(sorted(zs.items())), d) if zs is not None else None
    if key is not None and key in _ZOO_GROUPS:
        return _ZOO_GROUPS[key]
    from lqg_amd import specialize
    cache = system.__dict__.setdefault("_lqg_groups", {})
    d_key = (d, specialize.spec_versions(system))
    if d_key in cache:
        return cache[d_key]
    groups = []
    for i, (sub, cols, _) in enumerate(parts):
        for g in groups:
            ref, rcols, _ = parts[g[0]]
            if len(rcols) == len(cols) and _same_spec(ref, sub):
                g.append(i)
                break
        else:
            groups.append([i])
    cache[d_key] = groups
    if key is not None:
        _ZOO_GROUPS[key] = groups
    return groups


def floor_provably_inactive(system, eps=1e-8):
    """True when the eigenvalue floor of lqr.backward (lqg/control/lqr.py:27-28: Ht = H + max(0, eps - lambda_min(H)) I)
    can never be active, for any system and step.  H = R + B'SB >= R whenever S >= 0, and S stays >= 0 when Q and Qf
    are, so  lambda_min(R) >= eps, Q >= 0, Qf >= 0  suffice.  Only then is block decoupling EXACT: an active floor
    shifts every diagonal entry of the joint H by the same amount, i.e. it couples the components through the smallest
    eigenvalue among them.  Decided with Gershgorin bounds (elementwise, no factorisation) and, where those are
    inconclusive, the exact eigenvalues; one host synchronisation, cached on the instance per spec version."""
    from lqg_amd import specialize
    cache = system.__dict__.setdefault("_lqg_floor", {})
    key = (float(eps), specialize.spec_versions(system))
    if key in cache:
        return cache[key]
    a = system.actor
    first = specialize._first
    # "zero" to the rounding of the dtype the specs are STORED in: a matrix that is positive semi-definite in exact arithmetic with
    # a zero eigenvalue (every tracking model's Q) carries eigenvalues of -eps |Q| once its entries are rounded — fp32 specs that
    # move in time (round 6) would otherwise never pass.  The reference's own eigh of H in that dtype resolves no finer.
    tol = -1e-12 if a.Q.dtype == torch.float64 else -2e-6

    def slabs(M, max_elems=1 << 24):
        """Views of a (possibly huge) stack of small matrices, cut along its longest leading axis so that the fp64 temporaries of one
        slab stay small (2^17 systems x 500 steps of 6 x 6 fp32 costs are 9.4 GB; their fp64 images and intermediates were 60+ GB)."""
        if M.numel() <= max_elems or M.dim() < 3:
            yield M
            return
        ax = max(range(M.dim() - 2), key=lambda i: M.shape[i])
        step = max(1, int(max_elems // max(1, M.numel() // M.shape[ax])))
        for lo in range(0, M.shape[ax], step):
            yield M.narrow(ax, lo, min(step, M.shape[ax] - lo))

    def lower_bound(Mfull):
        """Lower bounds of the smallest eigenvalue over the whole stack: (relative to max(1, |M|max): Gershgorin on M, on its
        unit-diagonal scaling; absolute: the same two).  Gershgorin on M itself — every eigenvalue >= min_i (M_ii - sum_{j != i}
        |M_ij|) — and, where that is inconclusive, on the unit-diagonal scaling E M E (E = diag M^-1/2, a congruence: same
        signature): a cost D Q D whose Q is diagonally dominant passes the second form whatever the D (time-varying costs, round 6).
        Elementwise, no factorisation, slab by slab."""
        dev = Mfull.device
        inf = torch.tensor(float("inf"), dtype=torch.float64, device=dev)
        g1, g2, dmin, scale = inf.clone(), inf.clone(), inf.clone(), torch.ones((), dtype=torch.float64, device=dev)
        bad = torch.zeros((), dtype=torch.bool, device=dev)
        for Ms in slabs(Mfull):
            M = 0.5 * (Ms + Ms.transpose(-1, -2)).detach().double()
            diag = torch.diagonal(M, dim1=-2, dim2=-1)
            rows = M.abs().sum(-1)
            scale = torch.maximum(scale, M.abs().max())
            g1 = torch.minimum(g1, (diag - (rows - diag.abs())).min())
            pos = diag > 0
            inv = torch.where(pos, diag.clamp_min(1e-300).rsqrt(), torch.zeros_like(diag))
            offs = (M.abs() * inv.unsqueeze(-1) * inv.unsqueeze(-2)).sum(-1) - pos.double()
            # a row with a zero diagonal entry must vanish altogether; a negative diagonal entry fails both forms
            bad = bad | ((~pos) & (rows > 0)).any()
            g2 = torch.minimum(g2, torch.where(pos, 1.0 - offs, torch.zeros_like(offs)).min())
            dmin = torch.minimum(dmin, torch.where(pos, diag, torch.full_like(diag, float("inf"))).min())
        lb1 = g1 / scale
        lb2 = torch.where(bad, torch.full_like(g2, -1.0), g2)
        # (lambda_min(M) >= lambda_min(E M E) min_i M_ii when the scaled matrix is positive semi-definite)
        abs2 = torch.where(lb2 >= 0, lb2 * dmin, torch.full_like(lb2, -1.0))
        return scale, lb1, lb2, g1, abs2

    def eig_min(Mfull, chunk=1 << 18):
        """Smallest eigenvalue over a (possibly huge) stack of small matrices, in slabs (one syevd over 65 M matrices asks rocSOLVER
        for more workspace than it can get)."""
        best = float("inf")
        for Ms in slabs(Mfull):
            flat = (0.5 * (Ms + Ms.transpose(-1, -2))).detach().double().reshape(-1, *Ms.shape[-2:])
            for i in range(0, flat.shape[0], chunk):
                best = min(best, float(torch.linalg.eigvalsh(flat[i:i + chunk]).min()))
        return best

    R_, Q_, Qf_ = first(a.R), first(a.Q), a.Qf
    r_scale, _, _, r_abs1, r_abs2 = lower_bound(R_)
    q_scale, q_lb1, q_lb2, _, _ = lower_bound(Q_)
    qf_scale, qf_lb1, qf_lb2, _, _ = lower_bound(Qf_)
    checks = torch.stack([torch.maximum(r_abs1, r_abs2) - eps, torch.maximum(q_lb1, q_lb2) - tol, torch.maximum(qf_lb1, qf_lb2) - tol])
    flags = (checks >= 0).tolist()                       # the one synchronisation
    ok = all(flags)
    if not ok:
        r_ok = flags[0] or eig_min(R_) >= eps
        q_ok = r_ok and (flags[1] or eig_min(Q_) >= tol * max(1.0, float(q_scale)))
        ok = q_ok and (flags[2] or eig_min(Qf_) >= tol * max(1.0, float(qf_scale)))
    cache[key] = ok
    return ok


def plan(system, d, Sigma0=None, for_grad=False):
    """Decoupling plan of a System for data with d observed dims: list of (sub_system, data columns, belief dims)
    or None.  The component structure is cached on the instance — keyed by d, the sparsity pattern of Sigma0 (which
    enters the interaction graph) and the in-place version counters of the spec tensors; the model zoo uses the
    class-level probe pattern.

    for_grad=True (lqg_amd.grad, differentiable evaluation): (i) fields that require grad count as structurally FULL
    unless the structure is a property of a zoo constructor — a leaf matrix that merely holds zeros right now must not
    be split, its off-block derivatives are not zero; (ii) the sub-spec gathers are rebuilt on every call so that they
    live on the caller's current autograd graph (a cached gather made under no_grad, or one whose graph a previous
    backward() freed, would silently drop or break the gradient)."""
    from lqg_amd import specialize
    cache = system.__dict__.setdefault("_lqg_decouple", {})
    extra = None
    if Sigma0 is not None:
        nz = specialize._any_nz(Sigma0)
        if for_grad and Sigma0.requires_grad:
            nz = np.ones_like(nz)
        extra = nz | nz.T
    key = (d, None if extra is None else np.packbits(extra).tobytes(), bool(for_grad), specialize.spec_versions(system))
    if key not in cache:
        dims, masks, _ = specialize.system_pattern(system, d, grad_full=for_grad)
        cache[key] = dict(comps=components_from_masks(dims, masks, extra_bb=extra), parts=None)
    entry = cache[key]
    if entry["comps"] is None:
        return None
    if entry["parts"] is not None:
        return entry["parts"]
    first = specialize._first
    nz = specialize._any_nz
    # noise-factor column masks come from the instance (cheap) — only used to pick columns, never to drop values
    result = split_system(system, entry["comps"], nz(first(system.actor.V)), nz(first(system.actor.W)),
                          nz(first(system.dynamics.V)), nz(first(system.dynamics.W)))
    zs = specialize.zoo_structure(system)
    if zs is not None and not for_grad and Sigma0 is None:
        # the components' sparsity patterns are a property of the constructor too: specialize.system_pattern answers from
        # a class-level probe instead of reading the component's values back from the device (22 synchronisations)
        for i, (sub, _, _) in enumerate(result):
            sub._lqg_zoo_component = (type(system), tuple(sorted(zs.items())), d, i)
    elif zs is not None and for_grad and Sigma0 is None:
        for i, (sub, _, _) in enumerate(result):      # (specialize.adjoint_pattern: the constructor's structure serves the gradient too)
            sub._lqg_zoo_component_grad = (type(system), tuple(sorted(zs.items())), d, i)
    if not for_grad:
        entry["parts"] = result
    return result
