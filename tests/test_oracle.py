"""CPU: pin the oracle (oracle/lqg_np.py and oracle/lqg_oracle.c) before trusting it.

(1) against the golden vectors produced by the reference's own source (oracle/gen_golden.py, run in the
    build container under oracle/jax_standin.py);
(2) against reference-independent pins: closed forms, the steady-state DARE solution, a brute-force joint
    Gaussian, and the metamorphic relations the reference's own tests use (tests/lqg_test.py:69-93).
The reference's tests hold no numeric golden vectors, so (1)+(2) are the whole pin.
"""
import numpy as np
import pytest
import scipy.linalg as sla

import lqg_np as O
from conftest import golden_names, load_golden, relerr

ILL = {"pointmass_d4_T50"}   # observed covariance block with condition number ~1e12


@pytest.mark.parametrize("name", golden_names())
def test_numpy_oracle_matches_golden(name):
    g, actor, dyn = load_golden(name)
    S0 = g.get("Sigma0")
    L, l, H = O.backward(actor)
    K = O.forward(actor, O.default_sigma0(actor) if S0 is None else S0)
    assert relerr(L, g["L"]) < 1e-12 and relerr(H, g["H"]) < 1e-12 and relerr(K, g["K"]) < 1e-12
    assert np.abs(l - g["l"]).max() < 1e-12
    mu, Sig = O.conditional_moments(actor, dyn, g["x"][0], S0)
    assert relerr(mu, g["mu"][0]) < 1e-11 and relerr(Sig, g["Sigma"][0]) < 1e-11
    if g["x"].shape[1] <= 101:
        ll = O.log_likelihood(actor, dyn, g["x"], S0)
        assert np.abs(ll / g["ll"] - 1).max() < (1e-9 if name in ILL else 1e-12)
    if g["sim_eps"].shape[1] <= 100:
        X, XH, Y, U = O.simulate(actor, dyn, g["sim_eps"], g["sim_eta"], x0=g.get("x0"), Sigma0=S0)
        assert relerr(X, g["sim_x"]) < 1e-12 and relerr(XH, g["sim_xhat"]) < 1e-12
        assert relerr(Y, g["sim_y"]) < 1e-12 and relerr(U, g["sim_u"]) < 1e-12


@pytest.mark.parametrize("dtype,tol_mat,tol_ll", [(np.float64, 1e-10, 1e-11), (np.float32, 2e-5, 2e-6)],
                         ids=["f64", "f32"])
@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_golden(oracle_lib, name, dtype, tol_mat, tol_ll):
    OC = oracle_lib
    g, actor, dyn = load_golden(name)
    S0 = g.get("Sigma0")
    L, l, H = OC.riccati_backward(actor, dtype=dtype)
    K = OC.kalman_forward(actor, S0, dtype=dtype)
    assert relerr(L, g["L"]) < tol_mat and relerr(H, g["H"]) < tol_mat and relerr(K, g["K"]) < tol_mat
    assert np.abs(l - g["l"]).max() < tol_mat
    X, XH, Y, U = OC.simulate(actor, dyn, g["sim_eps"], g["sim_eta"], x0=g.get("x0"), Sigma0=S0, dtype=dtype)
    stol = 1e-10 if dtype == np.float64 else 5e-4
    assert relerr(X, g["sim_x"]) < stol and relerr(XH, g["sim_xhat"]) < stol
    assert relerr(Y, g["sim_y"]) < stol and relerr(U, g["sim_u"]) < stol
    if name in ILL and dtype == np.float32:
        return   # the literal fp32 recursion cannot represent this case (NaN), see tests/test_gpu_parity.py
    mu, Sig = OC.conditional_moments(actor, dyn, g["x"], S0, dtype=dtype)
    ll = OC.log_likelihood(actor, dyn, g["x"], S0, dtype=dtype)
    loose = 1e3 if name in ILL else 1.0
    assert relerr(mu, g["mu"]) < tol_mat * loose and relerr(Sig, g["Sigma"][0]) < tol_mat * loose
    assert np.abs(ll / g["ll"] - 1).max() < tol_ll * (1e2 if name in ILL else 1.0)


def test_c_oracle_batched_matches_unbatched(oracle_lib):
    """Leading system axis + stride-0 (shared / time-invariant) fields go through the same code path."""
    OC = oracle_lib
    g, actor, dyn = load_golden("subjective1d_T50")
    B = 3
    ab = {k: np.broadcast_to(v, (B,) + v.shape) for k, v in actor.items()}
    db = {k: np.broadcast_to(v, (B,) + v.shape) for k, v in dyn.items()}
    ab["W"] = np.ascontiguousarray(ab["W"]) * np.array([1.0, 1.5, 2.0])[:, None, None, None]
    xb = np.broadcast_to(g["x"], (B,) + g["x"].shape)
    ll = OC.log_likelihood(ab, db, xb)
    assert ll.shape == (B, g["x"].shape[0])
    assert np.abs(ll[0] / g["ll"] - 1).max() < 1e-11
    for i in (1, 2):
        ai = {k: v[i] for k, v in ab.items()}
        assert np.allclose(OC.log_likelihood(ai, dyn, g["x"]), ll[i], rtol=1e-13)
    assert not np.allclose(ll[0], ll[1])


# ---------------------------------------------------------------- reference-independent pins

def bounded(T, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5, process_noise=1.0,
            dt=1.0 / 60):
    """BoundedActor matrices (lqg/tracking/basic.py:7-64) restated for the oracle."""
    A = np.eye(2)
    B = dt * np.array([[0.0], [1.0]])
    V = np.diag([process_noise, action_variability])
    W = np.diag([sigma_target, sigma_cursor])
    Q = np.array([[1.0, -1.0], [-1.0, 1.0]])
    R = np.eye(1) * action_cost
    return O.time_stack_spec(A, B, np.eye(2), V, W, Q, R, T)


def subjective(T, subj_noise, subj_vel_noise, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05,
               action_variability=0.5, process_noise=1.0, dt=1.0 / 60):
    """SubjectiveActor(dim=1) matrices (lqg/tracking/subjective.py:15-47), dims already in swapped order."""
    dyn = O.dynamics_spec(np.eye(2), np.array([[0.0], [dt]]), np.eye(2), np.diag([process_noise, action_variability]),
                          np.diag([sigma_target, sigma_cursor]), T)
    A = np.array([[1.0, 0.0, dt], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    B = np.array([[0.0], [dt], [0.0]])
    F = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    V = np.diag([subj_noise, action_variability, subj_vel_noise])
    Q = np.array([[1.0, -1.0, 0.0], [-1.0, 1.0, 0.0], [0.0, 0.0, 0.0]])
    act = O.time_stack_spec(A, B, F, V, np.diag([sigma_target, sigma_cursor]), Q, np.eye(1) * action_cost, T)
    return act, dyn


def test_closed_form_first_kalman_and_last_riccati_gain(oracle_lib):
    """K[0] = V V^T (V V^T + W W^T)^-1 after the first predict, L[T-1] from S = Qf = Q."""
    spec = bounded(100)
    for back, fwd in ((O.backward, O.forward), (lambda s: oracle_lib.riccati_backward(s), None)):
        L, _, _ = back(spec)
        c = (1.0 / 60) / (0.05 + (1.0 / 60) ** 2)
        assert np.allclose(L[-1], [[c, -c]], rtol=1e-12)
        assert np.allclose(L[0], [[4.30857106, -4.30857106]], rtol=1e-8)
    for K in (O.forward(spec, O.default_sigma0(spec)), oracle_lib.kalman_forward(spec)):
        assert np.allclose(K[0], np.diag([2.0 / 38.0, 0.5 / 1.5]), rtol=1e-12)
        assert np.allclose(K[99], np.diag([0.15335548, 0.3903882]), rtol=1e-7)


def test_kalman_gain_converges_to_dare(oracle_lib):
    spec = bounded(3000)
    A, F, V, W = spec["A"][0], spec["F"][0], spec["V"][0], spec["W"][0]
    P = sla.solve_discrete_are(A.T, F.T, V @ V.T, W @ W.T)          # steady-state *predicted* covariance
    Kss = P @ F.T @ np.linalg.inv(F @ P @ F.T + W @ W.T)
    K = oracle_lib.kalman_forward(spec)
    assert np.allclose(K[-1], Kss, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("model", ["bounded", "subjective"])
def test_loglik_equals_brute_force_joint_gaussian(oracle_lib, model):
    """The scan formulation equals log p(x_1..x_T | x_0) of the stacked linear-Gaussian system (T small)."""
    T = 6
    rng = np.random.default_rng(3)
    if model == "bounded":
        act = dyn = bounded(T)
    else:
        act, dyn = subjective(T, subj_noise=1.3, subj_vel_noise=0.6)
    eps, eta = rng.standard_normal((2, T, 2)), rng.standard_normal((2, T, 2))
    X, _, _, _ = O.simulate(act, dyn, eps, eta)
    for i in range(2):
        bf = O.brute_force_loglik(act, dyn, X[i])
        assert abs(O.log_likelihood(act, dyn, X[i:i + 1])[0] - bf) < 1e-10
        assert abs(oracle_lib.log_likelihood(act, dyn, X[i:i + 1])[0] - bf) < 1e-10


def test_bounded_equals_subjective_without_subjective_component(oracle_lib):
    """Likelihood analogue of the reference's tests/lqg_test.py:69-93 (exercises x != b)."""
    T = 80
    rng = np.random.default_rng(5)
    kw = dict(sigma_target=6.0, sigma_cursor=3.0, action_cost=0.1, action_variability=0.5)
    b = bounded(T, **kw)
    act, dyn = subjective(T, subj_noise=1.0, subj_vel_noise=0.0, **kw)
    eps, eta = rng.standard_normal((4, T, 2)), rng.standard_normal((4, T, 2))
    Xb, _, _, _ = oracle_lib.simulate(b, b, eps, eta)
    Xs, _, _, _ = oracle_lib.simulate(act, dyn, eps, eta)
    assert np.allclose(Xb, Xs, rtol=1e-10, atol=1e-12)              # tests/lqg_test.py:93
    assert np.allclose(oracle_lib.log_likelihood(b, b, Xb), oracle_lib.log_likelihood(act, dyn, Xb), rtol=1e-10)


def test_sigma_is_data_independent(oracle_lib):
    g, actor, dyn = load_golden("subjective2d_T60")
    assert relerr(g["Sigma"][0], g["Sigma"][1]) == 0.0              # the reference's own output
    mu, Sig = oracle_lib.conditional_moments(actor, dyn, g["x"])
    assert Sig.shape == g["Sigma"].shape[1:] and mu.shape == g["mu"].shape


def test_nan_propagates_for_singular_observed_noise(oracle_lib):
    """V with zero noise on an observed dim makes Sigma_oo singular; the reference
    yields NaN/inf silently, so does the oracle (no exception)."""
    spec = bounded(10, action_variability=0.0)
    x = np.zeros((1, 11, 2))
    assert not np.isfinite(oracle_lib.log_likelihood(spec, spec, x)).all()


_ASAN_CHILD = r'''
import sys
import numpy as np
sys.path[:0] = [ROOT + "/tests", ROOT + "/oracle", ROOT]
from conftest import golden_names, load_golden
import oracle as OC
assert OC.LIB.endswith("liblqg_oracle_asan.so"), OC.LIB
n = 0
for name in golden_names():
    g, actor, dyn = load_golden(name)
    S0 = g.get("Sigma0")
    for dtype in (np.float64, np.float32):
        OC.riccati_backward(actor, dtype=dtype)
        OC.kalman_forward(actor, S0, dtype=dtype)
        OC.simulate(actor, dyn, g["sim_eps"], g["sim_eta"], x0=g.get("x0"), Sigma0=S0, dtype=dtype)
        OC.conditional_moments(actor, dyn, g["x"], S0, dtype=dtype)
        ll = OC.log_likelihood(actor, dyn, g["x"], S0, dtype=dtype)
        n += 1
    assert abs(ll / g["ll"] - 1).max() < 1e-3 or name == "pointmass_d4_T50"
# a batch of systems across OpenMP threads (the cpu_baseline leg's shape of call)
g, actor, dyn = load_golden("subjective2d_T60")
rep = lambda d: {k: np.repeat(v[None], 24, axis=0) for k, v in d.items()}
OC.lib().lqg_oracle_set_threads(4)
llb = OC.log_likelihood(rep(actor), rep(dyn), np.repeat(g["x"][None], 24, axis=0))
assert np.allclose(llb, g["ll"][None], rtol=1e-9)
print("ASAN_OK", n)
'''


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers():
    """`make -C oracle asan` (-fsanitize=address,undefined, no recovery): every golden case, both dtypes, every entry point,
    plus a batch across OpenMP threads, in a child process with libasan preloaded.  A heap / stack overrun, a use after free
    or undefined behaviour in the restatement aborts the child.  (CPU only: no GPU sanitizers exist on this pool.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "liblqg_oracle_asan.so"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan):
        pytest.skip("gcc has no libasan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", LQG_ORACLE_LIB=os.path.join(odir, "liblqg_oracle_asan.so"),
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _ASAN_CHILD], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ASAN_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
