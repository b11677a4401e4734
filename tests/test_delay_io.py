"""CPU: the formats either side of the hot path — temporal-delay augmentation
(lqg/tracking/delay.py:9-51) and the tracking-data loader (lqg/io.py:45-98) — against fixtures produced by the
reference's own source (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import scipy.io as spio
import torch

import lqg_amd
from lqg_amd import io as lio
from lqg_amd.tracking.delay import DelayedSubjectiveActor, TemporalDelayModel, delay_system
from conftest import GOLDEN_DIR, load_golden

BOUNDED = dict(T=30, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05)


@pytest.mark.parametrize("name,base,kw,delay", [
    ("delay1_bounded_T30", lqg_amd.BoundedActor, BOUNDED, 1),
    ("delay2_bounded_T30", lqg_amd.BoundedActor, BOUNDED, 2),
    ("delay1_subjective1d_T30", lqg_amd.SubjectiveActor, dict(dim=1, T=30, action_cost=0.5), 1),
])
def test_delay_system_builds_the_reference_matrices(name, base, kw, delay):
    g, actor, dyn = load_golden(name)
    m = TemporalDelayModel(base(device="cpu", dtype=torch.float64, **kw), delay=delay)
    for f in lqg_amd.LQGSpec._fields:
        a, d = getattr(m.actor, f).numpy(), getattr(m.dynamics, f).numpy()
        assert a.shape == actor[f].shape and d.shape == dyn[f].shape, f
        assert np.array_equal(a, actor[f]) and np.array_equal(d, dyn[f]), f
    assert m.actor.A.stride(0) == 0 and m.actor.V.stride(0) == 0          # time-invariant stays a stride-0 view
    assert m.xdim == dyn["A"].shape[1] and m.bdim == actor["A"].shape[1]


def test_delay_system_shift_register_semantics():
    base = lqg_amd.BoundedActor(T=5, device="cpu", dtype=torch.float64)
    s = delay_system(base.dynamics, 3)
    n = base.xdim
    assert s.A.shape == (5, 4 * n, 4 * n) and s.V.shape == (5, 4 * n, base.dynamics.V.shape[-1] + 3 * n)
    x = torch.arange(4.0 * n, dtype=torch.float64)
    nxt = s.A[0] @ x
    assert torch.equal(nxt[n:], x[:-n])                                    # every copy moves one slot down
    assert torch.equal(nxt[:n], base.dynamics.A[0] @ x[:n])
    assert torch.equal(s.F[0] @ x, base.dynamics.F[0] @ x[-n:])            # the oldest copy is observed
    assert torch.equal(s.R, base.dynamics.R) and torch.equal(s.W, base.dynamics.W)
    assert float(s.q.abs().sum() + s.P.abs().sum() + s.r.abs().sum()) == 0.0
    assert torch.equal(delay_system(base.dynamics, 0).A, base.dynamics.A)
    with pytest.raises(ValueError):
        delay_system(base.dynamics, -1)


def test_delay_system_keeps_candidate_axis_and_time_variation():
    sig = torch.tensor([3.0, 6.0, 9.0], dtype=torch.float64)
    m = TemporalDelayModel(lqg_amd.BoundedActor(T=7, sigma_target=sig, device="cpu"), delay=2)
    assert m.actor.A.shape == (3, 7, 6, 6) and m.actor.W.shape == (3, 7, 2, 2) and m.n_systems == 3
    one = TemporalDelayModel(lqg_amd.BoundedActor(T=7, sigma_target=6.0, device="cpu", dtype=torch.float64), delay=2)
    for f in lqg_amd.LQGSpec._fields:
        assert torch.equal(getattr(m.actor, f)[1], getattr(one.actor, f)), f
    tv = one.actor._replace(A=one.actor.A.clone() * torch.linspace(1, 2, 7, dtype=torch.float64)[:, None, None])
    base = lqg_amd.BoundedActor(T=7, sigma_target=6.0, device="cpu", dtype=torch.float64).actor
    base = base._replace(A=base.A * torch.linspace(1, 2, 7, dtype=torch.float64)[:, None, None])
    s = delay_system(base, 2)
    assert s.A.stride(0) != 0 and torch.equal(s.A[:, :2, :2], base.A) and torch.equal(s.A[3, 2:, :4], torch.eye(4, dtype=torch.float64))
    del tv


def test_delayed_subjective_actor_mirrors_the_reference_class():
    m = DelayedSubjectiveActor(T=20, device="cpu")
    assert (m.xdim, m.bdim, m.udim, m.ydim, m.T) == (2 * 13, 3 * 13, 1, 2, 20)


# ---- lqg/io.py

@pytest.fixture(scope="module")
def tracking_fixture(tmp_path_factory):
    g = np.load(os.path.join(GOLDEN_DIR, "io", "tracking_small.npz"))
    d = tmp_path_factory.mktemp("bonnen")
    spio.savemat(os.path.join(d, "data.mat"), {k: g[k] for k in ("sigma", "target", "response")})
    return g, str(d)


@pytest.mark.parametrize("tag,kw", [("default", dict()), ("nodelay", dict(delay=0, clip=50)),
                                    ("raw", dict(delay=5, clip=0, subtract_mean=False))])
def test_load_tracking_data_matches_the_reference_loader(tracking_fixture, tag, kw):
    g, path = tracking_fixture
    data, sigmas = lio.load_tracking_data(data_path=path, **kw)
    assert data.dtype == np.float32 and data.shape == g["data_" + tag].shape
    assert np.array_equal(data, g["data_" + tag])                          # bit-exact: same float32 operations
    assert np.array_equal(sigmas, g["sigmas_" + tag])


def test_load_tracking_data_properties(tracking_fixture):
    g, path = tracking_fixture
    data, sigmas = lio.load_tracking_data(data_path=path)
    S = g["target"].shape[1]
    assert data.shape == (4, 3, S - 120 - 12, 2)
    assert np.all(np.diff(sigmas) > 0) and np.array_equal(sigmas, np.unique(np.round(g["sigma"] * 1.32)))
    assert np.all(data[:, :, 0, 0] == 0)                                   # target starts at the origin
    raw, _ = lio.load_tracking_data(data_path=path, delay=7, clip=3, subtract_mean=False)
    width = np.round(g["sigma"] * 1.32)
    rows = np.where(width == sigmas[2])[0]
    t0 = g["target"][rows[1], 3].astype(np.float32)
    assert np.array_equal(raw[2, 1, :, 0], g["target"][rows[1], 3:-7].astype(np.float32) - t0)
    assert np.array_equal(raw[2, 1, :, 1], g["response"][rows[1], 10:].astype(np.float32) - t0)   # response shifted by delay


def test_load_tracking_data_rejects_ragged_conditions(tmp_path):
    spio.savemat(os.path.join(tmp_path, "data.mat"), {"sigma": np.array([10, 10, 12], dtype=np.uint8),
                                                      "target": np.zeros((3, 200)), "response": np.zeros((3, 200), np.uint16)})
    with pytest.raises(ValueError, match="trial counts"):
        lio.load_tracking_data(data_path=str(tmp_path))


def test_loadmat_flattens_structs(tmp_path):
    spio.savemat(os.path.join(tmp_path, "s.mat"), {"cfg": {"rate": 60.0, "inner": {"k": np.arange(3)}}})
    m = lio.loadmat(os.path.join(tmp_path, "s.mat"))
    assert m["cfg"]["rate"] == 60.0 and np.array_equal(m["cfg"]["inner"]["k"], np.arange(3))


def test_out_of_range_shapes_go_to_the_cooperative_kernels_never_to_a_compile():
    """The delay-12 model (x=26, b=39) is beyond the register-resident lane kernels: it must be routed to the
    run-time-dims cooperative kernels of the MAIN library at once — never start an hours-long on-demand hipcc build."""
    from lqg_amd import _abi, _hip
    big = DelayedSubjectiveActor(T=20, device="cpu")
    ln = _hip.Launch(big.actor, big.dynamics, d=2, n_trials=1)
    assert (ln.dims["x"], ln.dims["b"], ln.dims["u"], ln.dims["y"]) == (26, 39, 1, 2)
    assert _abi.library_for(ln.dims, n_sys=1 << 20) is _abi.load()       # whatever the batch: no lane kernels possible
    assert _abi.load().lqg_strategy(ln.p) == _abi.STRATEGY_COOP
    assert _abi.shape_available(26, 39, 1, 2, 2) and not _abi.shape_in_range(11, 4, 1, 2, 2)
    assert _abi.shape_in_range(10, 10, 2, 4, 4) and _abi.shape_in_range(6, 6, 1, 2, 2)
    with pytest.raises(_abi.LqgHipError, match="outside the dims"):      # u = 7: beyond both kernel families
        _abi.library_for(dict(x=30, b=30, u=7, y=2, d=2))


@pytest.mark.gpu
def test_delayed_model_value_and_gradient_through_batched_finite_differences():
    """The reference differentiates DelayedSubjectiveActor with jax.grad (lqg/infer/utils.py:14-41 runs NUTS on any model).
    Here shapes without adjoint lane kernels (m = 65) take central differences in log-space with the 2P + 1 perturbed
    systems as the candidate axis of ONE evaluation (every system is its own workgroup of the cooperative kernels): check
    it against differences of separately constructed models, one parameter at a time."""
    import math
    from lqg_amd.infer.gradient import value_and_grad
    from lqg_amd.infer.models import get_model_params
    from lqg_amd.tracking.delay import DelayedSubjectiveActor
    names = sorted(get_model_params(DelayedSubjectiveActor))
    assert names == sorted(["c", "action_variability", "subj_noise", "subj_vel_noise", "sigma_target", "sigma_cursor"])
    T = 60
    truth = DelayedSubjectiveActor(T=T, device="cuda", dtype=torch.float64)
    x = truth.simulate(3, n=6)[..., :2].contiguous()
    p = dict(c=0.4, action_variability=0.6, subj_noise=1.2, subj_vel_noise=8.0, sigma_target=5.0, sigma_cursor=2.5)
    v, g = value_and_grad(x, DelayedSubjectiveActor, p, method="fd")      # (x has T + 1 rows = T steps, lqg_model's convention)
    ll = lambda q: float(DelayedSubjectiveActor(T=T, device="cuda", dtype=torch.float64, **q).log_likelihood(x).sum())
    assert abs(v / ll(p) - 1) < 1e-12
    h = 1e-4
    for k in ("c", "sigma_target", "subj_vel_noise"):
        up, dn = dict(p), dict(p)
        up[k], dn[k] = p[k] * math.exp(h), p[k] * math.exp(-h)
        fd = (ll(up) - ll(dn)) / (2 * h) / p[k]
        assert abs(g[k] - fd) < 1e-6 * max(1.0, abs(fd)), (k, g[k], fd)
