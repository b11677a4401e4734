"""Reverse-mode gradient of the log-likelihood (jax.grad in lqg/optim.py:142-147).

CPU (-m "not gpu"): the NumPy restatement of the adjoint sweep (oracle/lqg_adjoint_np.py) is pinned against
torch.autograd of the literal torch restatement (oracle/lqg_torch_ref.py, itself checked against the golden vectors
here) and against central finite differences of the golden-pinned C oracle.
GPU (-m gpu): the HIP sweeps through the C ABI against that restatement, and end to end through torch.autograd and the
model constructors against finite differences of the fp64 HIP forward path."""
import numpy as np
import pytest
import torch

import lqg_adjoint_np as ADJ
import lqg_torch_ref as TR
from conftest import load_golden

SYM = ("Q", "Qf")


def _cut(spec, T):
    return {k: (v[:T] if (v.ndim == 3 or (v.ndim == 2 and k in ("q", "r"))) else v) for k, v in spec.items()}


@pytest.mark.parametrize("name,T", [("subjective1d_T50", 50), ("pointmass_d2_T50", 20), ("timevarying_T30", 30),
                                    ("tutorial_lqg_T100", 100), ("delay1_bounded_T30", 30)])
def test_numpy_adjoint_matches_torch_autograd(name, T):
    g, actor, dyn = load_golden(name)
    actor, dyn, x = _cut(actor, T), _cut(dyn, T), g["x"][:2, :T + 1]
    S0 = g.get("Sigma0")
    w = np.array([0.7, 1.3])
    a = {k: torch.tensor(v, requires_grad=True) for k, v in actor.items()}
    d = {k: torch.tensor(v, requires_grad=True) for k, v in dyn.items()}
    S0t = None if S0 is None else torch.tensor(S0, requires_grad=True)
    ll = TR.log_likelihood(a, d, torch.tensor(x), S0t)
    if T == actor["A"].shape[0] and T == g["x"].shape[1] - 1:
        assert np.abs(ll.detach().numpy() - g["ll"][:2]).max() < 1e-10 * np.abs(g["ll"]).max()   # the torch restatement is pinned
    (ll * torch.tensor(w)).sum().backward()
    ll2, ga, gd, S0b = ADJ.loglik_grad(actor, dyn, x, w, S0)
    assert np.abs(ll2 - ll.detach().numpy()).max() < 1e-10 * np.abs(ll2).max()
    for who, mine, ref in (("actor", ga, a), ("dyn", gd, d)):
        for k, v in mine.items():
            r = ref[k].grad.numpy()
            if k in SYM:
                r = 0.5 * (r + np.swapaxes(r, -1, -2))
            assert np.abs(v - r).max() < 1e-7 * max(np.abs(r).max(), 1e-3), (who, k)
    if S0 is not None:
        r = S0t.grad.numpy()
        assert np.abs(S0b - 0.5 * (r + r.T)).max() < 1e-9


def test_numpy_adjoint_matches_finite_differences_of_the_c_oracle(oracle_lib):
    g, actor, dyn = load_golden("subjective1d_T50")
    x = g["x"]
    w = np.linspace(0.5, 1.5, x.shape[0])
    _, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x, w)
    f = lambda a, d: float((oracle_lib.log_likelihood(a, d, x) * w).sum())
    h = 1e-6
    for who, spec, grads in (("dyn", dyn, gd), ("actor", actor, ga)):
        for k, (i, j) in (("A", (0, 1)), ("B", (1, 0)), ("F", (1, 1)), ("V", (1, 1)), ("W", (0, 0))):
            Mp, Mm = spec[k].copy(), spec[k].copy()
            Mp[:, i, j] += h
            Mm[:, i, j] -= h
            sp, sm = dict(spec, **{k: Mp}), dict(spec, **{k: Mm})
            fd = (f(actor, sp) - f(actor, sm)) / (2 * h) if who == "dyn" else (f(sp, dyn) - f(sm, dyn)) / (2 * h)
            an = grads[k][:, i, j].sum()
            assert abs(fd - an) < 2e-6 * max(1.0, abs(an)), (who, k, fd, an)


@pytest.mark.parametrize("name,T,n", [("subjective1d_T50", 50, 3), ("pointmass_d2_T50", 25, 2), ("tutorial_lqg_T100", 60, 4),
                                      ("bounded_T100", 100, 3), ("pointmass_d4_T50", 20, 2), ("relobs_T40", 40, 2),
                                      ("subjective2d_T60", 30, 3)])
def test_split_restatement_equals_the_per_lane_restatement(name, T, n):
    """oracle/lqg_adjoint_split_np.py — the sweep cut as the round-5 kernels cut it (matrix adjoints once per system, the trials
    entering through per-step sums, conditioning in Cholesky form) — against the per-(system, trial) restatement that is pinned
    to autograd of the literal restatement above: value, every bar, the Sigma0 bar."""
    import lqg_adjoint_split_np as SPL
    g, actor, dyn = load_golden(name)
    actor, dyn, x = _cut(actor, T), _cut(dyn, T), g["x"][:n, :T + 1]
    w = np.linspace(0.6, 1.4, x.shape[0])
    for S0 in (None, g.get("Sigma0")) if g.get("Sigma0") is not None else (None,):
        ll, ga, gd, s0 = ADJ.loglik_grad(actor, dyn, x, w, S0)
        ll2, ga2, gd2, s02, sums = SPL.loglik_grad(actor, dyn, x, w, S0)
        assert np.abs(ll - ll2).max() < 1e-11 * np.abs(ll).max()
        for a_, b_ in ((ga, ga2), (gd, gd2)):
            for k in a_:
                assert np.abs(a_[k] - b_[k]).max() < 1e-8 * max(np.abs(a_[k]).max(), 1e-3), k
        assert np.abs(s0 - s02).max() < 1e-8 * max(np.abs(s0).max(), 1e-3)
        assert abs(sums["g"] - w.sum()) < 1e-12


# ------------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu
TI_CASES = ["bounded_T100", "optimal_T30", "relobs_T40", "subjective1d_T50", "pointmass_d2_T50", "pointmass_d4_T50",
            "tutorial_lqg_T100"]


def _ti(spec):
    """Golden specs are materialised [T, ...] stacks of one matrix: re-express them as stride-0 views."""
    return spec._replace(**{f: getattr(spec, f)[:1].expand_as(getattr(spec, f))
                            for f in ("A", "B", "F", "V", "W", "Q", "R", "q", "P", "r")})


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 2e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("name", TI_CASES)
def test_hip_adjoint_matches_the_restatement(name, dtype, tol):
    from gpu_common import system_from_golden
    from lqg_amd import grad as G
    # (pointmass_d4 in fp32 -- every state observed, cond(Sigma_oo) ~ 5e8 -- runs its sweeps over an fp64 image and rounds
    # value and bars once: grad.Sweep; the bars still arrive as fp32 tensors)
    g, actor, dyn = load_golden(name)
    x = g["x"]
    w = np.linspace(0.5, 1.5, x.shape[0])
    if name == "pointmass_d4_T50" and dtype == torch.float32:
        # at this conditioning the ROUNDING OF THE INPUTS to fp32 alone moves the gradient by 7e-4 (measured): the reference
        # is the restatement on the problem the fp32 caller actually poses
        r32 = lambda d_: {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float64
                              else v) for k, v in d_.items()}
        ll_ref, ga, gd, _ = ADJ.loglik_grad(r32(actor), r32(dyn), x.astype(np.float32).astype(np.float64), w)
    else:
        ll_ref, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x, w)
    s = system_from_golden(actor, dyn, dtype)
    ll, bars, _ = G.raw_grad(_ti(s.actor), _ti(s.dynamics), torch.as_tensor(x, dtype=dtype, device="cuda"),
                             g=torch.as_tensor(w, dtype=dtype, device="cuda"))
    assert np.abs(ll.double().cpu().numpy() - g["ll"]).max() < max(tol * 1e-1, 1e-10) * np.abs(g["ll"]).max()
    tot = {k: v.sum(1)[0].double().cpu().numpy() for k, v in bars.items()}
    sym2 = lambda M: M + M.T
    got = {"dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"], "dV": sym2(tot["dVV"]) @ dyn["V"][0],
           "dW": sym2(tot["dWW"]) @ dyn["W"][0], "aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"],
           "aF": tot["aF"], "aV": sym2(tot["aVV"]) @ actor["V"][0], "aW": sym2(tot["aWW"]) @ actor["W"][0],
           "aQ": tot["aQ"], "aR": tot["aR"], "aQf": tot["aQf"]}
    ref = {"d" + k: v.sum(0) for k, v in gd.items()}
    ref.update({"a" + k: (v.sum(0) if v.ndim == 3 else v) for k, v in ga.items()})
    scale = max(np.abs(v).max() for v in ref.values())
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() < tol * max(np.abs(r).max(), 1e-3 * scale), k


def _dense_grad_system(name, dtype, n, device="cuda", T=None):
    """A golden case as a System whose differentiable fields all REQUIRE GRAD (time-invariant views of one leaf each): its
    adjoint pattern is the dense one, so every entry of every bar is formed and can be compared with the restatement."""
    import lqg_amd
    from gpu_common import system_from_golden
    g, actor, dyn = load_golden(name)
    x = g["x"]
    if T is not None:
        actor, dyn, x = _cut(actor, T), _cut(dyn, T), x[:, :T + 1]
    rng = np.random.default_rng(5)
    while x.shape[0] < n:                                  # more trials: smooth perturbations of the first one
        x = np.concatenate([x, x[:1] + 0.05 * rng.standard_normal(x[:1].shape).cumsum(1)], 0)
    x = x[:n]
    s = system_from_golden(actor, dyn, dtype, device)

    def leafed(spec):
        out = {}
        for f in spec._fields:
            t = getattr(spec, f)
            if f in ("A", "B", "F", "V", "W", "Q", "R", "q", "P", "r"):
                base = t[:1].clone()
                if f in ("A", "B", "F", "V", "W", "Q", "R"):
                    base.requires_grad_(True)
                t2 = base.expand_as(t)
                if getattr(t, "_lqg_zero", False):
                    from lqg_amd.utils import mark_zero
                    t2 = mark_zero(t2)
                t = t2
            out[f] = t
        return spec._replace(**out)

    a = leafed(s.actor)
    d = a if s.actor is s.dynamics else leafed(s.dynamics)
    return lqg_amd.System(actor=a, dynamics=d), actor, dyn, x, g


def split_sweep_test_patterns():
    """(dims, masks, key, live) of the dense adjoint patterns the GPU tests below use (compiled by __graft_entry__.build())."""
    from lqg_amd import specialize
    out = []
    for name in ("bounded_T100", "subjective1d_T50", "relobs_T40", "pointmass_d2_T50"):
        s, _, _, x, _ = _dense_grad_system(name, torch.float64, 1, device="cpu")
        out.append(specialize.adjoint_pattern(s, x.shape[-1]))
    return out


def _compare_bars(bars, actor, dyn, ga, gd, tol):
    tot = {k: v.sum(1)[0].double().cpu().numpy() for k, v in bars.items()}
    sym2 = lambda M: M + M.T
    got = {"dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"], "dV": sym2(tot["dVV"]) @ dyn["V"][0],
           "dW": sym2(tot["dWW"]) @ dyn["W"][0], "aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"],
           "aF": tot["aF"], "aV": sym2(tot["aVV"]) @ actor["V"][0], "aW": sym2(tot["aWW"]) @ actor["W"][0],
           "aQ": tot["aQ"], "aR": tot["aR"], "aQf": tot["aQf"]}
    ref = {"d" + k: v.sum(0) for k, v in gd.items()}
    ref.update({"a" + k: (v.sum(0) if v.ndim == 3 else v) for k, v in ga.items()})
    scale = max(np.abs(v).max() for v in ref.values())
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() < tol * max(np.abs(r).max(), 1e-3 * scale), k


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("n", [1, 2, 5])
@pytest.mark.parametrize("name", ["bounded_T100", "subjective1d_T50", "relobs_T40", "pointmass_d2_T50"])
def test_split_sweep_matches_the_restatement(name, n, dtype, tol):
    """The round-5 sweep (csrc/lqg_adjoint_sp.hpp: matrix adjoints once per system; 1 / 2 trials swept in the system's lane, more
    through the operator stream + the per-trial mu-bar sweep and its per-step trial sums) against oracle/lqg_adjoint_np.py:
    value and every entry of every bar, upstream weights included.  fp32 stays fp32 here (the round-1 kernels ran these
    fully observed models over an fp64 image)."""
    from lqg_amd import grad as G
    s, actor, dyn, x, g = _dense_grad_system(name, dtype, n)
    w = np.linspace(0.5, 1.5, n)
    ll_ref, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x, w)
    with torch.no_grad():
        sw = G.Sweep(s.actor, s.dynamics, torch.as_tensor(x, dtype=dtype, device="cuda"), system=s)
        assert sw.sp is not None and sw.per_sys == 1 and sw.out_dtype is None
        ll = sw.forward()
        bars = sw.reverse(torch.as_tensor(w, dtype=dtype, device="cuda"))
    assert np.abs(ll.double().cpu().numpy() - ll_ref).max() < max(tol * 1e-2, 1e-10) * np.abs(ll_ref).max()
    _compare_bars(bars, actor, dyn, ga, gd, tol)


@gpu
def test_split_sweep_many_trials_several_workgroups_per_system():
    """More trials than one workgroup holds (1024): the per-step trial sums arrive as several partial records per system; a
    candidate axis on top (two systems sharing the trials)."""
    import lqg_adjoint_split_np as SPL
    from lqg_amd import grad as G
    import lqg_amd
    name, n, T = "bounded_T100", 2500, 24
    sT, actor, dyn, x, g = _dense_grad_system(name, torch.float64, n, T=T)
    w = np.cos(np.arange(n))
    ll_ref, ga, gd, _, _ = SPL.loglik_grad(actor, dyn, x, w)
    with torch.no_grad():
        sw = G.Sweep(sT.actor, sT.dynamics, torch.as_tensor(x, dtype=torch.float64, device="cuda"), system=sT)
        assert sw.sp is not None
        ll = sw.forward()
        bars = sw.reverse(torch.as_tensor(w, dtype=torch.float64, device="cuda"))
    assert np.abs(ll.cpu().numpy() - ll_ref).max() < 1e-10 * np.abs(ll_ref).max()
    _compare_bars(bars, actor, dyn, ga, gd, 1e-9)


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("ctor,kw,n", [("BoundedActor", dict(T=150), 7), ("SubjectiveActor", dict(T=90, dim=2), 1),
                                       ("SubjectiveActor", dict(T=90, dim=2), 4), ("PointMassBoundedActor", dict(T=70), 3)])
def test_split_sweep_equals_the_round1_lane_kernels_through_the_constructors(ctor, kw, n, dtype, tol):
    """Parameter gradients through the zoo constructors (class patterns: structural zeros compiled out, identical components
    merged as trials) from the split sweep against the round-1 kernels (LQG_ADJOINT_SP=0), a candidate axis of 3."""
    import lqg_amd
    from lqg_amd import options
    cls = getattr(lqg_amd, ctor)
    d = 4 if kw.get("dim") == 2 else 2
    base = dict(sigma_target=[4.0, 6.0, 9.0], action_cost=[0.05, 0.1, 0.3])
    with torch.no_grad():
        x = cls(device="cuda", dtype=torch.float64, **kw).simulate(3, n=n)[..., :d].to(dtype).contiguous()
    out = {}
    for sp in (1, 0):
        with options.override(ADJOINT_SP=sp):
            th = {k: torch.tensor(v, dtype=dtype, device="cuda", requires_grad=True) for k, v in base.items()}
            ll = cls(device="cuda", dtype=dtype, **kw, **th).log_likelihood(x)
            (ll * torch.linspace(0.5, 1.5, n, dtype=dtype, device="cuda")).sum().backward()
            out[sp] = (ll.detach().double().cpu().numpy(), {k: v.grad.double().cpu().numpy() for k, v in th.items()})
    assert np.abs(out[1][0] - out[0][0]).max() < max(tol * 1e-2, 1e-10) * np.abs(out[0][0]).max()
    for k in base:
        assert np.abs(out[1][1][k] - out[0][1][k]).max() < tol * max(np.abs(out[0][1][k]).max(), 1e-3), k


def _fd(make, x, names, h=1e-6):
    """Central differences of sum(ll) w.r.t. scalar model parameters through the fp64 HIP forward path."""
    out = {}
    base = {k: v for k, v in names.items()}
    for k in names:
        vals = []
        for s in (+1, -1):
            kw = dict(base)
            kw[k] = base[k] * (1 + s * h)
            with torch.no_grad():
                vals.append(float(make(**kw).log_likelihood(x).sum()))
        out[k] = (vals[0] - vals[1]) / (2 * h * base[k])
    return out


@gpu
@pytest.mark.parametrize("ctor,kw,params,d", [
    ("BoundedActor", dict(T=120), dict(sigma_target=6.0, sigma_cursor=1.5, action_cost=0.05, action_variability=0.4), 2),
    ("SubjectiveActor", dict(T=100, dim=1), dict(subj_noise=1.2, subj_vel_noise=0.6, sigma_target=5.0, sigma_cursor=2.0,
                                                 action_cost=0.3, action_variability=0.5), 2),
    ("SubjectiveActor", dict(T=60, dim=2), dict(subj_noise=1.2, subj_vel_noise=0.6, sigma_target=5.0, action_cost=0.3), 4),
    ("BoundedActor", dict(T=60, dim=2), dict(sigma_target=6.0, action_cost=0.1), 4),
    ("RelativeObservationBoundedActor", dict(T=80), dict(sigma=4.0, action_cost=0.2), 2),
    ("PointMassBoundedActor", dict(T=60), dict(sigma_target=5.0, sigma_cursor=1.0, action_cost=0.02, action_variability=0.5), 2),
])
def test_autograd_through_the_model_constructors_matches_finite_differences(ctor, kw, params, d):
    """`jax.grad(lambda theta: Model(**theta).log_likelihood(x).sum())` of the reference, via torch.autograd."""
    import lqg_amd
    cls = getattr(lqg_amd, ctor)
    make = lambda **p: cls(device="cuda", dtype=torch.float64, **kw, **p)
    with torch.no_grad():
        x = make(**params).simulate(3, n=5)[..., :d].contiguous()
    theta = {k: torch.tensor(v, dtype=torch.float64, device="cuda", requires_grad=True) for k, v in params.items()}
    model = make(**theta)
    ll = model.log_likelihood(x)
    with torch.no_grad():
        assert torch.allclose(ll.detach(), make(**params).log_likelihood(x), rtol=1e-12)
    ll.sum().backward()
    fd = _fd(make, x, params)
    for k in params:
        an = float(theta[k].grad)
        assert abs(an - fd[k]) < 1e-5 * max(1.0, abs(fd[k])), (k, an, fd[k])


@gpu
def test_candidate_axis_gradients_and_upstream_weights():
    import lqg_amd
    sig = torch.tensor([3.0, 6.0, 12.0], dtype=torch.float64, device="cuda", requires_grad=True)
    cost = torch.tensor(0.1, dtype=torch.float64, device="cuda", requires_grad=True)
    with torch.no_grad():
        x = lqg_amd.BoundedActor(T=80, sigma_target=6.0, device="cuda", dtype=torch.float64).simulate(1, n=4)
    w = torch.tensor([[1.0, 0.5, 2.0, 0.0], [0.3, 0.3, 0.3, 0.3], [1.0, -1.0, 1.0, -1.0]], dtype=torch.float64, device="cuda")
    ll = lqg_amd.BoundedActor(T=80, sigma_target=sig, action_cost=cost, device="cuda", dtype=torch.float64).log_likelihood(x)
    assert ll.shape == (3, 4)
    (ll * w).sum().backward(retain_graph=True)
    g_sig, g_cost = sig.grad.clone(), float(cost.grad)
    sig.grad = None
    (ll * w).sum().backward()                                                # a second backward re-runs the forward sweeps
    assert torch.allclose(sig.grad, g_sig, rtol=1e-12)
    cost.grad = cost.grad - g_cost
    tot_cost = 0.0
    for c in range(3):
        s1 = torch.tensor(float(sig[c].detach()), dtype=torch.float64, device="cuda", requires_grad=True)
        c1 = torch.tensor(0.1, dtype=torch.float64, device="cuda", requires_grad=True)
        l1 = lqg_amd.BoundedActor(T=80, sigma_target=s1, action_cost=c1, device="cuda", dtype=torch.float64).log_likelihood(x)
        (l1 * w[c]).sum().backward()
        assert abs(float(s1.grad) - float(g_sig[c])) < 1e-9 * max(1.0, abs(float(s1.grad)))
        tot_cost += float(c1.grad)
    assert abs(tot_cost - g_cost) < 1e-9 * max(1.0, abs(g_cost))             # a shared parameter sums over systems


@gpu
def test_sigma0_gradient():
    import lqg_amd
    m = lqg_amd.SubjectiveActor(dim=1, T=40, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        x = m.simulate(2, n=3)
    S0 = torch.tensor([[2.0, 0.3, 0.1], [0.3, 1.5, 0.2], [0.1, 0.2, 1.0]], dtype=torch.float64, device="cuda", requires_grad=True)
    m.log_likelihood(x, Sigma0=S0).sum().backward()
    an = S0.grad.cpu().numpy()
    h = 1e-6
    for (i, j) in ((0, 0), (0, 1), (2, 1)):
        E = torch.zeros(3, 3, dtype=torch.float64, device="cuda")
        E[i, j] += 0.5
        E[j, i] += 0.5
        with torch.no_grad():
            fd = float(m.log_likelihood(x, Sigma0=S0.detach() + h * E).sum() - m.log_likelihood(x, Sigma0=S0.detach() - h * E).sum()) / (2 * h)
        assert abs(fd - 0.5 * (an[i, j] + an[j, i])) < 1e-6 * max(1.0, abs(fd))


@gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 5e-4)], ids=["f64", "f32"])
def test_time_varying_specs_get_one_bar_per_step(dtype, tol):
    """Every field time-varying, custom Sigma0 (golden case timevarying_T30): HIP per-step bars against the restatement,
    then through torch.autograd with the full [T, r, c] fields as leaves."""
    from gpu_common import system_from_golden
    from lqg_amd import grad as G
    g, actor, dyn = load_golden("timevarying_T30")
    x, S0 = g["x"], g["Sigma0"]
    w = np.linspace(0.5, 1.5, x.shape[0])
    _, ga, gd, S0b = ADJ.loglik_grad(actor, dyn, x, w, S0)
    s = system_from_golden(actor, dyn, dtype)
    xt = torch.as_tensor(x, dtype=dtype, device="cuda")
    S0t = torch.as_tensor(S0, dtype=dtype, device="cuda")
    ll, bars, _ = G.raw_grad(s.actor, s.dynamics, xt, g=torch.as_tensor(w, dtype=dtype, device="cuda"), Sigma0=S0t)
    assert np.abs(ll.double().cpu().numpy() - g["ll"]).max() < max(tol * 1e-1, 1e-10) * np.abs(g["ll"]).max()
    tot = {k: v.sum(1)[0].double().cpu().numpy() for k, v in bars.items()}
    assert tot["dA"].shape == (30, 2, 2) and tot["aQf"].shape == (3, 3)
    sym2 = lambda M: M + np.swapaxes(M, -1, -2)
    got = {"dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"], "dV": sym2(tot["dVV"]) @ dyn["V"], "dW": sym2(tot["dWW"]) @ dyn["W"],
           "aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"], "aF": tot["aF"], "aV": sym2(tot["aVV"]) @ actor["V"],
           "aW": sym2(tot["aWW"]) @ actor["W"], "aQ": tot["aQ"], "aR": tot["aR"], "aQf": tot["aQf"]}
    ref = {"d" + k: v for k, v in gd.items()}
    ref.update({"a" + k: v for k, v in ga.items()})
    scale = max(np.abs(v).max() for v in ref.values())
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() < tol * max(np.abs(r).max(), 1e-3 * scale), k
    assert np.abs(tot["aS0"] - S0b).max() < tol * max(np.abs(S0b).max(), 1e-3)
    if dtype == torch.float64:                       # autograd: leaves are the time-varying fields themselves
        A = s.actor.A.clone().requires_grad_(True)
        Wd = s.dynamics.W.clone().requires_grad_(True)
        sys2 = type(s)(actor=s.actor._replace(A=A), dynamics=s.dynamics._replace(W=Wd))
        (sys2.log_likelihood(xt, Sigma0=S0t) * torch.as_tensor(w, device="cuda")).sum().backward()
        assert np.abs(A.grad.cpu().numpy() - ga["A"]).max() < 1e-8 * np.abs(ga["A"]).max()
        assert np.abs(Wd.grad.cpu().numpy() - gd["W"]).max() < 1e-8 * max(np.abs(gd["W"]).max(), 1e-3)


# ---- the COOPERATIVE reverse-mode sweep (csrc/lqg_coop_adjoint.hip: one workgroup per system, run-time dims, fp64, bars summed
# over the trials): every shape without adjoint lane kernels — the reference's delay models — and, forced with
# LQG_COOP_ADJOINT=1, every golden case the lane kernels are pinned on
COOP_CASES = TI_CASES + ["delay1_bounded_T30", "delay2_bounded_T30", "hand2d_T40", "subjective2d_T60", "delay12_subjective1d_T30"]


def _got_and_ref(tot, ga, gd, actor, dyn, time_varying=False):
    sym2 = lambda M: M + np.swapaxes(M, -1, -2)
    V = (lambda sp, k: sp[k]) if time_varying else (lambda sp, k: sp[k][0])
    got = {"dA": tot["dA"], "dB": tot["dB"], "dF": tot["dF"], "dV": sym2(tot["dVV"]) @ V(dyn, "V"),
           "dW": sym2(tot["dWW"]) @ V(dyn, "W"), "aA": tot["aA"] + tot["aA2"], "aB": tot["aB"] + tot["aB2"],
           "aF": tot["aF"], "aV": sym2(tot["aVV"]) @ V(actor, "V"), "aW": sym2(tot["aWW"]) @ V(actor, "W"),
           "aQ": tot["aQ"], "aR": tot["aR"], "aQf": tot["aQf"]}
    red = (lambda v: v) if time_varying else (lambda v: v.sum(0) if v.ndim == 3 else v)
    ref = {"d" + k: red(v) for k, v in gd.items()}
    ref.update({"a" + k: red(v) for k, v in ga.items()})
    return got, ref


@gpu
@pytest.mark.parametrize("name", COOP_CASES)
def test_cooperative_adjoint_matches_the_restatement(name, monkeypatch):
    """fp64, 1e-8 of the NumPy restatement (round-3 review, item 4: `delay12_subjective1d_T30`, x = 26, b = 39, included)."""
    from gpu_common import system_from_golden
    from lqg_amd import _abi, grad as G
    monkeypatch.setenv("LQG_COOP_ADJOINT", "1")
    g, actor, dyn = load_golden(name)
    x = g["x"]
    w = np.linspace(0.5, 1.5, x.shape[0])
    ll_ref, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x, w)
    s = system_from_golden(actor, dyn, torch.float64)
    sw = G.Sweep(_ti(s.actor), _ti(s.dynamics), torch.as_tensor(x, dtype=torch.float64, device="cuda"))
    assert sw.per_sys == 1                                                 # bars summed over the trials
    ll = sw.forward()
    bars = sw.reverse(torch.as_tensor(w, dtype=torch.float64, device="cuda"))
    assert np.abs(ll.cpu().numpy() - g["ll"]).max() < (1e-7 if name == "pointmass_d4_T50" else 1e-10) * np.abs(g["ll"]).max()
    assert bars["dA"].shape[:2] == (1, 1)
    tot = {k: v.sum(1)[0].cpu().numpy() for k, v in bars.items()}
    got, ref = _got_and_ref(tot, ga, gd, actor, dyn)
    scale = max(np.abs(v).max() for v in ref.values())
    tol = 1e-6 if name == "pointmass_d4_T50" else 1e-8                     # (cond 5.6e8: the lane kernels' test allows the same)
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() < tol * max(np.abs(r).max(), 1e-3 * scale), k


@gpu
def test_cooperative_adjoint_time_varying_specs_sigma0_and_candidates(monkeypatch):
    """Per-step bars (every field time-varying, custom Sigma0: golden timevarying_T30) and a candidate axis with per-pair
    weights, cooperative sweep against the lane kernels' results and the restatement."""
    from gpu_common import system_from_golden
    from lqg_amd import grad as G
    import lqg_amd
    g, actor, dyn = load_golden("timevarying_T30")
    x, S0 = g["x"], g["Sigma0"]
    w = np.linspace(0.5, 1.5, x.shape[0])
    _, ga, gd, S0b = ADJ.loglik_grad(actor, dyn, x, w, S0)
    s = system_from_golden(actor, dyn, torch.float64)
    xt, S0t = torch.as_tensor(x, device="cuda"), torch.as_tensor(S0, device="cuda")
    monkeypatch.setenv("LQG_COOP_ADJOINT", "1")
    ll, bars, _ = G.raw_grad(s.actor, s.dynamics, xt, g=torch.as_tensor(w, device="cuda"), Sigma0=S0t)
    assert np.abs(ll.cpu().numpy() - g["ll"]).max() < 1e-10 * np.abs(g["ll"]).max()
    tot = {k: v.sum(1)[0].cpu().numpy() for k, v in bars.items()}
    assert tot["dA"].shape == (30, 2, 2) and tot["aQf"].shape == (3, 3)
    got, ref = _got_and_ref(tot, ga, gd, actor, dyn, time_varying=True)
    scale = max(np.abs(v).max() for v in ref.values())
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() < 1e-8 * max(np.abs(r).max(), 1e-3 * scale), k
    assert np.abs(tot["aS0"] - S0b).max() < 1e-8 * max(np.abs(S0b).max(), 1e-3)
    # candidates x trials with weights, through autograd: cooperative == lane kernels
    sig = torch.tensor([3.0, 6.0, 12.0], dtype=torch.float64, device="cuda")
    with torch.no_grad():
        xb = lqg_amd.BoundedActor(T=80, sigma_target=6.0, device="cuda", dtype=torch.float64).simulate(1, n=4)
    wt = torch.tensor([[1.0, 0.5, 2.0, 0.0], [0.3, 0.3, 0.3, 0.3], [1.0, -1.0, 1.0, -1.0]], dtype=torch.float64, device="cuda")
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LQG_COOP_ADJOINT", mode)
        sg = sig.clone().requires_grad_(True)
        cost = torch.tensor(0.1, dtype=torch.float64, device="cuda", requires_grad=True)
        ll = lqg_amd.BoundedActor(T=80, sigma_target=sg, action_cost=cost, device="cuda", dtype=torch.float64).log_likelihood(xb)
        (ll * wt).sum().backward()
        res[mode] = (ll.detach().clone(), sg.grad.clone(), float(cost.grad))
    assert torch.allclose(res["1"][0], res["0"][0], rtol=1e-11)
    assert torch.allclose(res["1"][1], res["0"][1], rtol=1e-8) and abs(res["1"][2] / res["0"][2] - 1) < 1e-8


@gpu
def test_reverse_mode_serves_the_delay_models_end_to_end():
    """`jax.grad` in the reference differentiates DelayedSubjectiveActor like any other model (lqg/tracking/delay.py:44-51,
    lqg/optim.py:142-147): lqg_grad_supported is true for (x, b) = (26, 39), value_and_grad(method="adjoint") agrees with
    central differences of the fp64 forward path, and an fp32 caller gets the fp64 image rounded once."""
    import ctypes as C
    from lqg_amd import _abi
    from lqg_amd.infer import gradient
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    lib = _abi.load()
    dm = _abi._dims_struct(dict(x=26, b=39, u=1, y=2, d=2))
    assert lib.lqg_grad_supported(_abi.F64, C.byref(dm)) == 1 and lib.lqg_kernel_supported(_abi.FAM_ADJOINT, C.byref(dm)) == 0
    with torch.no_grad():
        x = DelayedSubjectiveActor(T=40, device="cuda", dtype=torch.float64).simulate(4, n=6)[..., :2].contiguous()
    x = torch.cat([x, x[:, -1:]], dim=1)                              # lqg_model's convention: T rows = T - 1 steps
    p = dict(c=0.5, action_variability=0.5, subj_noise=1.0, subj_vel_noise=10.0, sigma_target=6.0, sigma_cursor=3.0)
    v1, g1 = gradient.value_and_grad(x, DelayedSubjectiveActor, p, method="adjoint")
    v2, g2 = gradient.value_and_grad(x, DelayedSubjectiveActor, p, method="fd")
    assert abs(v1 - v2) < 1e-9 * abs(v2)
    for k in p:
        assert abs(g1[k] - g2[k]) < 2e-5 * max(1.0, abs(g2[k])), (k, g1[k], g2[k])
    grads = {}
    for dt in (torch.float64, torch.float32):                          # the same model and data in both precisions
        sig = torch.tensor(6.0, dtype=dt, device="cuda", requires_grad=True)
        ll = DelayedSubjectiveActor(T=41, sigma_target=sig, device="cuda", dtype=dt).log_likelihood(x.to(dt))
        assert ll.dtype == dt
        ll.sum().backward()
        grads[dt] = float(sig.grad)
    assert abs(grads[torch.float64] - g1["sigma_target"]) < 1e-9 * max(1.0, abs(g1["sigma_target"]))
    assert abs(grads[torch.float32] - grads[torch.float64]) < 2e-3 * max(1.0, abs(grads[torch.float64]))


@gpu
@pytest.mark.parametrize("shared_x", [False, True])
def test_candidate_chunked_cooperative_adjoint_bounds_the_live_workspace(monkeypatch, shared_x):
    """the candidate chunking of the cooperative reverse sweep (grad._candidate_chunks / _one: 4096 DelayedSubjectiveActor
    candidates would ask for 172 GB in one piece) must bound the PEAK, under autograd too — every piece runs forward before any
    runs backward, so a piece gives its workspace back after its forward and re-runs the forward sweep inside its backward.
    Workspace limit pinned to two systems' worth -> 3 pieces of 6 candidates: same gradients as the unchunked run, peak memory
    below three systems' worth (unchunked: six).  shared_x: a [1, n, T+1, d] x is shared by the pieces, not sliced to nothing."""
    from lqg_amd import grad as G, plan as _plan
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    dev, dt, B = "cuda", torch.float64, 6
    with torch.no_grad():
        x = DelayedSubjectiveActor(T=120, device=dev, dtype=dt).simulate(4, n=12)[..., :2].contiguous()
    xin = x[None] if shared_x else x

    def run():
        sig = torch.linspace(5.0, 9.0, B, device=dev, dtype=dt).requires_grad_(True)
        m = DelayedSubjectiveActor(T=120, sigma_target=sig, device=dev, dtype=dt)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        ll = m.log_likelihood(xin)
        ll.sum().backward()
        torch.cuda.synchronize()
        return ll.detach().clone(), sig.grad.clone(), torch.cuda.max_memory_allocated() - base

    ll0, g0, peak0 = run()                                           # one piece
    sig = torch.linspace(5.0, 9.0, B, device=dev, dtype=dt)
    m = DelayedSubjectiveActor(T=120, sigma_target=sig, device=dev, dtype=dt)
    assert G._candidate_chunks(m, x, None) is None                   # fits the device's limit in one piece
    from lqg_amd import _hip, workload
    one = workload.slice_system(m, 0, 1)
    ln = _hip.Launch(one.actor, one.dynamics, d=2, n_trials=12)
    per = int(ln.require_gpu(_abi_mod().FAM_ADJOINT).lqg_grad_workspace_bytes(C_mod().byref(ln.p), 64))
    assert per > (4 << 20)                                           # the workspace dominates the other allocations of this test
    monkeypatch.setattr(_plan, "OPS_WORKSPACE_LIMIT", 2 * per + 1024)
    assert G._candidate_chunks(m, x, None) == [(0, 2), (2, 4), (4, 6)]
    ll1, g1, peak1 = run()
    assert torch.allclose(ll1, ll0, rtol=1e-13, atol=0) and torch.allclose(g1, g0, rtol=1e-12, atol=0)
    assert peak0 > 5.5 * per                                         # the unchunked run holds all six systems' state
    assert peak1 < 3.0 * per, (peak1 / per, peak0 / per)             # chunked: one piece of two at a time


def _abi_mod():
    from lqg_amd import _abi
    return _abi


def _C_mod_cache():
    import ctypes
    return ctypes


C_mod = _C_mod_cache


@gpu
def test_value_and_grad_adjoint_agrees_with_finite_difference_method():
    import lqg_amd
    from lqg_amd.infer import gradient
    with torch.no_grad():
        x = lqg_amd.BoundedActor(T=150, sigma_target=8.0, action_cost=0.2, device="cuda", dtype=torch.float64).simulate(4, n=10)
    p = dict(sigma_target=6.0, sigma_cursor=2.0, action_cost=0.3, action_variability=0.4)
    v1, g1 = gradient.value_and_grad(x, lqg_amd.BoundedActor, p, method="adjoint")
    v2, g2 = gradient.value_and_grad(x, lqg_amd.BoundedActor, p, method="fd")
    assert abs(v1 - v2) < 1e-9 * abs(v2)
    for k in p:
        assert abs(g1[k] - g2[k]) < 1e-5 * max(1.0, abs(g2[k])), k


@gpu
def test_max_likelihood_with_adjoint_gradients_follows_the_finite_difference_path():
    import lqg_amd
    from lqg_amd.infer import max_likelihood
    with torch.no_grad():
        x = lqg_amd.BoundedActor(T=100, sigma_target=12.0, action_variability=0.4, action_cost=0.1, sigma_cursor=2.0,
                                 device="cuda", dtype=torch.float64).simulate(11, n=12)
    kw = dict(steps=25, step_size=0.05, action_cost=0.1, sigma_cursor=2.0)
    p1, l1 = max_likelihood(x, lqg_amd.BoundedActor, method="adjoint", **kw)
    p2, l2 = max_likelihood(x, lqg_amd.BoundedActor, method="fd", **kw)
    assert torch.allclose(l1, l2, rtol=1e-6) and float(l1[-1]) < float(l1[0])
    for k in p1:
        assert abs(p1[k] - p2[k]) < 1e-4 * abs(p2[k])


@gpu
def test_fp32_gradients_track_fp64_on_the_decoupled_headline_model():
    """SubjectiveActor(dim=2) (two decoupled 1-D components), a batch of candidates: fp32 parameter gradients against fp64."""
    import lqg_amd
    names = ("sigma_target", "subj_noise", "subj_vel_noise", "action_cost")
    base = dict(sigma_target=[4.0, 8.0, 16.0, 30.0], subj_noise=[1.0, 1.5, 0.7, 1.2], subj_vel_noise=[0.5, 0.8, 0.4, 1.0],
                action_cost=[0.05, 0.2, 0.5, 1.0])
    with torch.no_grad():
        x = lqg_amd.SubjectiveActor(dim=2, T=200, device="cuda", dtype=torch.float64).simulate(9, n=6)
    grads = {}
    for dtype in (torch.float64, torch.float32):
        theta = {k: torch.tensor(base[k], dtype=dtype, device="cuda", requires_grad=True) for k in names}
        ll = lqg_amd.SubjectiveActor(dim=2, T=200, device="cuda", dtype=dtype, **theta).log_likelihood(x.to(dtype))
        assert ll.shape == (4, 6) and ll.dtype == dtype
        ll.sum().backward()
        grads[dtype] = {k: theta[k].grad.double().cpu().numpy() for k in names}
    for k in names:
        ref = grads[torch.float64][k]
        assert np.abs(grads[torch.float32][k] - ref).max() < 2e-3 * max(np.abs(ref).max(), 1.0), k


@pytest.mark.gpu
def test_leaf_matrices_get_their_off_block_gradients():
    """Round-1 advisor finding: a leaf matrix that merely HOLDS zeros off its blocks (A = I as an autograd leaf, a
    diagonal Sigma0 leaf) must not be split into components by the differentiable path — d ll / d A[0, 1] is not zero.
    Two decoupled 1-D trackers (x = b = u = y = 2): autograd against central finite differences of the fp64 forward
    path for an off-block and an in-block entry of A and of Sigma0."""
    import lqg_amd
    from lqg_amd import decouple
    dev, dt = "cuda", torch.float64
    t = lambda v: torch.tensor(v, dtype=dt, device=dev)
    A0 = torch.eye(2, dtype=dt, device=dev)
    Bm = t([[0.3, 0.0], [0.0, 0.2]])
    F, V, W = torch.eye(2, dtype=dt, device=dev), t([[1.0, 0.0], [0.0, 0.7]]), t([[2.0, 0.0], [0.0, 1.5]])
    Q, Rm = t([[1.0, 0.0], [0.0, 0.5]]), t([[0.1, 0.0], [0.0, 0.2]])
    S00 = t([[1.5, 0.0], [0.0, 0.8]])
    T = 25
    with torch.no_grad():
        x = lqg_amd.LQG(A0, Bm, F, V, W, Q, Rm, T=T).simulate(5, n=4)
        assert decouple.plan(lqg_amd.LQG(A0, Bm, F, V, W, Q, Rm, T=T), 2) is not None      # by VALUE it decouples

    def f(A, S0):
        return lqg_amd.LQG(A, Bm, F, V, W, Q, Rm, T=T).log_likelihood(x, Sigma0=S0).sum()

    A = A0.clone().requires_grad_(True)
    S0 = S00.clone().requires_grad_(True)
    f(A, S0).backward()
    h = 1e-6
    for (i, j) in ((0, 1), (1, 0), (0, 0)):
        with torch.no_grad():
            Ap, Am = A0.clone(), A0.clone()
            Ap[i, j] += h
            Am[i, j] -= h
            fd = float(f(Ap, S00) - f(Am, S00)) / (2 * h)
        assert abs(float(A.grad[i, j]) - fd) < 1e-5 * max(1.0, abs(fd)), (i, j, float(A.grad[i, j]), fd)
        if (i, j) == (0, 1):
            assert abs(fd) > 1e-3                                  # the off-block derivative really is non-zero
    with torch.no_grad():                                          # Sigma0: symmetric perturbation of the off-diagonal pair
        Sp, Sm = S00.clone(), S00.clone()
        Sp[0, 1] += h; Sp[1, 0] += h
        Sm[0, 1] -= h; Sm[1, 0] -= h
        fd = float(f(A0, Sp) - f(A0, Sm)) / (2 * h)
    assert abs(float(S0.grad[0, 1] + S0.grad[1, 0]) - fd) < 1e-5 * max(1.0, abs(fd))


@gpu
def test_grad_plan_is_the_persistent_form_of_the_autograd_path():
    """lqg_amd.grad.GradPlan (what bench.py times): decisions taken once, repeated runs are pure launches — same value as
    log_likelihood, same bars as a fresh Sweep of the merged component, run after run."""
    import lqg_amd
    from lqg_amd import grad as G, workload
    dev = torch.device("cuda")
    system, _ = workload.headline_system(96, 80, seed=3, device=dev, dtype=torch.float64)
    x = workload.pack_trials(workload.simulate_one_trial_each(system, seed=4))
    gp = G.GradPlan(system, x, events=True)
    assert len(gp.items) == 1 and gp.items[0]["group"] == 2 and gp.items[0]["sweep"].sp is not None, gp.description
    ll1, bars1 = gp.run()
    ll2, bars2 = gp.run()
    with torch.no_grad():
        ref = system.log_likelihood(x)
    assert torch.allclose(ll1, ref, rtol=1e-12) and torch.equal(ll1, ll2)
    for k in bars1[0]:
        assert torch.equal(bars1[0][k], bars2[0][k]), k
    ph = gp.phase_ms()
    assert all(v >= 0 for v in ph.values()) and ph["rev_system"] > 0
    assert gp.items[0]["sweep"].per_sys == 1
