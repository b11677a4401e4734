"""ctypes front end of the C oracle (oracle/lqg_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.  It takes NumPy
arrays in the reference's layout — spec fields [T, ...] or, with a leading system axis, [B, T, ...];
trajectories x[n, T+1, d] or [B, n, T+1, d] — and calls the host twins of the C-ABI entry points with
the same argument structs (include/lqg_hip.h) the HIP library takes.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lqg_amd import _abi  # noqa: E402  (struct definitions only; no HIP library is loaded)

# LQG_ORACLE_LIB: another build of the same sources (the sanitizer build `make -C oracle asan`, tests/test_oracle.py)
LIB = os.environ.get("LQG_ORACLE_LIB") or os.path.join(HERE, "liblqg_oracle.so")
_lib = None


def build(force=False):
    src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("lqg_oracle.c", "lqg_oracle_body.inc"))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < src_m:
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", os.path.basename(LIB)])
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = _abi.declare(C.CDLL(LIB), prefix="lqg_oracle_", with_stream=False)
        _lib.lqg_oracle_max_threads.restype = C.c_int
        _lib.lqg_oracle_set_threads.argtypes = [C.c_int]
    return _lib


def _es(a):
    return [s // a.itemsize for s in a.strides]


def _field_view(a, batched, has_time, vector):
    return _abi.mat_view(a.ctypes.data, a.shape, _es(a), batched, has_time, vector)


_VEC = {"q", "qf", "r"}
_NOTIME = {"Qf", "qf"}


class Prob:
    """Holds the ctypes Problem plus the arrays it points into."""

    def __init__(self, actor, dyn, d=None, n_trials=1, Sigma0=None, dtype=np.float64, eps=1e-8):
        dt = np.dtype(dtype)
        self.np_dtype = dt
        batched = actor["A"].ndim == 4
        self.batched = batched
        conv = lambda a: np.asarray(a).astype(dt, copy=False) if np.asarray(a).dtype != dt else np.asarray(a)
        self.actor = {k: conv(v) for k, v in actor.items() if v is not None}
        self.dyn = {k: conv(v) for k, v in dyn.items() if v is not None}
        A = self.actor["A"]
        self.B = A.shape[0] if batched else 1
        self.T = A.shape[1] if batched else A.shape[0]
        b = A.shape[-1]
        u = self.actor["B"].shape[-1]
        y = self.actor["F"].shape[-2]
        x = self.dyn["A"].shape[-1]
        self.dims = dict(x=x, b=b, u=u, y=y, d=(x if d is None else d), nva=self.actor["V"].shape[-1],
                         nwa=self.actor["W"].shape[-1], nvd=self.dyn["V"].shape[-1], nwd=self.dyn["W"].shape[-1])
        p = _abi.Problem()
        p.dtype = _abi.F64 if dt == np.float64 else _abi.F32
        p.T, p.n_sys, p.n_trials, p.eps = self.T, self.B, n_trials, eps
        p.dims = _abi.Dims(**self.dims)
        for name, spec, dst in (("actor", self.actor, p.actor), ("dyn", self.dyn, p.dynamics)):
            for f in _abi.SPEC_FIELDS:
                if f in spec:
                    setattr(dst, f, _field_view(spec[f], batched, f not in _NOTIME, f in _VEC))
                else:
                    setattr(dst, f, _abi.NULL_VIEW)
        if Sigma0 is not None:
            self.Sigma0 = conv(Sigma0)
            p.Sigma0 = _field_view(self.Sigma0, self.Sigma0.ndim == 3, False, False)
        else:
            p.Sigma0 = _abi.NULL_VIEW
        self.p = p

    def out(self, *shape):
        lead = (self.B,) if self.batched else ()
        return np.zeros(lead + shape, dtype=self.np_dtype)

    def view(self, a, vector=False, has_time=True):
        return _field_view(a, self.batched, has_time, vector)

    def traj(self, a):
        return _abi.traj_view(a.ctypes.data, a.shape, _es(a), self.batched)


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def riccati_backward(actor, dtype=np.float64, eps=1e-8):
    pr = Prob(actor, actor, dtype=dtype, eps=eps)
    dm = pr.dims
    L, l, H = pr.out(pr.T, dm["u"], dm["b"]), pr.out(pr.T, dm["u"]), pr.out(pr.T, dm["u"], dm["u"])
    _ck(lib().lqg_oracle_riccati_backward(C.byref(pr.p), pr.view(L), pr.view(l, vector=True), pr.view(H)), "riccati")
    return L, l, H


def kalman_forward(actor, Sigma0=None, dtype=np.float64):
    pr = Prob(actor, actor, Sigma0=Sigma0, dtype=dtype)
    K = pr.out(pr.T, pr.dims["b"], pr.dims["y"])
    _ck(lib().lqg_oracle_kalman_forward(C.byref(pr.p), pr.view(K)), "kalman")
    return K


def _prep_x(pr, x):
    x = np.ascontiguousarray(np.asarray(x, dtype=pr.np_dtype))
    return x


def conditional_moments(actor, dyn, x, Sigma0=None, dtype=np.float64, eps=1e-8):
    """x[n,T+1,d] (or [B,n,T+1,d]) -> mu[(B,)n,T,m], Sigma[(B,)T,m,m]."""
    xa = np.asarray(x)
    n, d = xa.shape[-3], xa.shape[-1]
    pr = Prob(actor, dyn, d=d, n_trials=n, Sigma0=Sigma0, dtype=dtype, eps=eps)
    xa = _prep_x(pr, xa)
    m = pr.dims["x"] + pr.dims["b"]
    mu, Sig = pr.out(n, pr.T, m), pr.out(pr.T, m, m)
    _ck(lib().lqg_oracle_conditional_moments(C.byref(pr.p), pr.traj(xa), pr.traj(mu), pr.view(Sig)), "moments")
    return mu, Sig


def log_likelihood(actor, dyn, x, Sigma0=None, dtype=np.float64, eps=1e-8):
    """x[n,T+1,d] (or [B,n,T+1,d]) -> ll[(B,)n]."""
    xa = np.asarray(x)
    n, d = xa.shape[-3], xa.shape[-1]
    pr = Prob(actor, dyn, d=d, n_trials=n, Sigma0=Sigma0, dtype=dtype, eps=eps)
    xa = _prep_x(pr, xa)
    ll = pr.out(n)
    _ck(lib().lqg_oracle_log_likelihood(C.byref(pr.p), pr.traj(xa), ll.ctypes.data, n if pr.batched else 0, 1),
        "log_likelihood")
    return ll


def simulate(actor, dyn, eps_noise, eta_noise, x0=None, xhat0=None, Sigma0=None, dtype=np.float64, eps=1e-8):
    """eps_noise[(B,)n,T,x], eta_noise[(B,)n,T,y] -> x[(B,)n,T+1,x], xhat[(B,)n,T+1,b], y, u."""
    n = np.asarray(eps_noise).shape[-3]
    pr = Prob(actor, dyn, n_trials=n, Sigma0=Sigma0, dtype=dtype, eps=eps)
    dm = pr.dims
    L, l, _ = riccati_backward(actor, dtype=dtype, eps=eps)
    K = kalman_forward(actor, Sigma0=Sigma0, dtype=dtype)
    e1 = np.ascontiguousarray(np.asarray(eps_noise, dtype=pr.np_dtype))
    e2 = np.ascontiguousarray(np.asarray(eta_noise, dtype=pr.np_dtype))
    xs, xh = pr.out(n, pr.T + 1, dm["x"]), pr.out(n, pr.T + 1, dm["b"])
    ys, us = pr.out(n, pr.T, dm["y"]), pr.out(n, pr.T, dm["u"])

    def init_view(v, k):
        if v is None:
            return _abi.NULL_VIEW, None
        a = np.ascontiguousarray(np.asarray(v, dtype=pr.np_dtype))
        if a.ndim == 1:
            return _abi.View(a.ctypes.data, 0, 0, 1, 0), a
        return _abi.View(a.ctypes.data, a.shape[1], 0, 1, 0), a

    v0, k0 = init_view(x0, dm["x"])
    vh, kh = init_view(xhat0, dm["b"])
    _ck(lib().lqg_oracle_simulate(C.byref(pr.p), pr.view(L), pr.view(l, vector=True), pr.view(K), pr.traj(e1),
                                  pr.traj(e2), v0, vh, pr.traj(xs), pr.traj(xh), pr.traj(ys), pr.traj(us)),
        "simulate")
    return xs, xh, ys, us
