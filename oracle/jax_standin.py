"""NumPy-backed stand-in for the handful of `jax` / `numpyro` names the reference's hot path touches.

TEST INFRASTRUCTURE ONLY.  Nothing under `lqg_amd/` may import this module.

Why it exists: RothkopfLab/lqg (mounted read-only at /root/reference in the build container)
is pure Python on JAX + NumPyro; neither is installed here and there is no network.  The reference's
hot-path source (lqg/control/lqr.py, lqg/belief/kf.py, lqg/system.py, lqg/tracking/*.py) only
uses a small slice of those APIs, all of which have exact NumPy/SciPy (LAPACK, fp64) counterparts.
`install()` registers fake modules in `sys.modules` so that the reference's *own source lines*
execute unmodified on NumPy fp64; `oracle/gen_golden.py` uses that to emit the golden vectors in
`tests/golden/`.  The reference never travels to the GPU box; only the vectors do.

What is substituted (third-party arithmetic, pinned in the reference's uv.lock as jax 0.6.2 /
numpyro 0.19.0; not under /root/reference):
  jnp.linalg.eigh / solve / inv  -> numpy.linalg (LAPACK syevd / gesv / getri)
  lax.scan, jax.vmap             -> Python loops
  numpyro MultivariateNormal     -> Cholesky-based log-density (same formula NumPyro uses)
Same algorithm classes, results equal to rounding (fp64), not bitwise.
"""
import importlib.metadata
import sys
import types

import numpy as np
import scipy.linalg as sla


class _AtSetter:
    def __init__(self, arr, idx=None):
        self._arr, self._idx = arr, idx

    def __getitem__(self, idx):
        return _AtSetter(self._arr, idx)

    def set(self, value):
        out = np.array(self._arr, copy=True)
        out[self._idx] = value
        return out.view(AtArray)


class AtArray(np.ndarray):
    """ndarray with the functional-update `.at[idx].set(v)` accessor of jax arrays."""

    @property
    def at(self):
        return _AtSetter(self)


def _wrap(a):
    return np.asarray(a).view(AtArray)


def _tree_map(f, tree):
    if isinstance(tree, tuple):
        if hasattr(tree, "_fields"):
            return type(tree)(*[_tree_map(f, t) for t in tree])
        return tuple(_tree_map(f, t) for t in tree)
    if isinstance(tree, list):
        return [_tree_map(f, t) for t in tree]
    if tree is None:
        return None
    return f(tree)


def _tree_leaves(tree):
    if isinstance(tree, (tuple, list)):
        out = []
        for t in tree:
            out.extend(_tree_leaves(t))
        return out
    if tree is None:
        return []
    return [tree]


def _tree_stack(trees):
    first = trees[0]
    if isinstance(first, tuple):
        parts = [_tree_stack([t[i] for t in trees]) for i in range(len(first))]
        if hasattr(first, "_fields"):
            return type(first)(*parts)
        return tuple(parts)
    if first is None:
        return None
    return np.stack([np.asarray(t) for t in trees])


def scan(f, init, xs, length=None, reverse=False):
    """lax.scan over tuple pytrees; stacked outputs are in forward time order also for reverse."""
    leaves = _tree_leaves(xs)
    n = int(leaves[0].shape[0]) if leaves else int(length)
    order = range(n - 1, -1, -1) if reverse else range(n)
    carry, ys = init, [None] * n
    for i in order:
        carry, y = f(carry, _tree_map(lambda a: a[i], xs))
        ys[i] = y
    return carry, _tree_stack(ys)


def vmap(f, in_axes=0, out_axes=0):
    def wrapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        n = None
        for a, ax in zip(args, axes):
            if ax is not None:
                n = _tree_leaves(a)[0].shape[ax]
                break
        outs = []
        for i in range(n):
            call = [a if ax is None else _tree_map(lambda v: np.take(v, i, axis=ax), a)
                    for a, ax in zip(args, axes)]
            outs.append(f(*call))
        return _tree_stack(outs)

    return wrapped


def _prngkey(seed):
    return np.array([0, int(seed)], dtype=np.uint32)


def _split(key, num=2):
    ss = np.random.SeedSequence([int(v) for v in np.asarray(key).ravel()])
    return np.stack([child.generate_state(2).astype(np.uint32) for child in ss.spawn(num)])


def _normal(key, shape=()):
    # NOT jax's threefry stream: distributional use only.
    return np.random.default_rng([int(v) for v in np.asarray(key).ravel()]).standard_normal(shape)


class Distribution:
    def __init__(self, batch_shape=(), event_shape=(), validate_args=None):
        self._batch_shape, self._event_shape = tuple(batch_shape), tuple(event_shape)

    @property
    def batch_shape(self):
        return self._batch_shape

    @property
    def event_shape(self):
        return self._event_shape

    def shape(self, sample_shape=()):
        return tuple(sample_shape) + self._batch_shape + self._event_shape


class MultivariateNormal(Distribution):
    """log-density as numpyro.distributions.MultivariateNormal computes it: Cholesky of the covariance,
    triangular solve, -0.5*(k log 2pi + |L^-1 (v-mu)|^2) - sum log diag L."""

    def __init__(self, loc=0.0, covariance_matrix=None):
        self.loc = np.asarray(loc)
        self.covariance_matrix = np.asarray(covariance_matrix)
        self.scale_tril = np.linalg.cholesky(self.covariance_matrix)
        batch = np.broadcast_shapes(self.loc.shape[:-1], self.covariance_matrix.shape[:-2])
        super().__init__(batch_shape=batch, event_shape=self.loc.shape[-1:])
        self._reinterpreted = 0

    def to_event(self, n):
        self._reinterpreted = n
        return self

    def shape(self, sample_shape=()):
        return tuple(sample_shape) + self._batch_shape + self._event_shape

    def log_prob(self, value):
        diff = np.asarray(value) - self.loc
        Lc = np.broadcast_to(self.scale_tril, diff.shape[:-1] + self.scale_tril.shape[-2:])
        z = np.empty_like(diff)
        for idx in np.ndindex(*diff.shape[:-1]):
            z[idx] = sla.solve_triangular(Lc[idx], diff[idx], lower=True)
        k = diff.shape[-1]
        half_logdet = np.log(np.diagonal(Lc, axis1=-2, axis2=-1)).sum(-1)
        lp = -0.5 * (k * np.log(2 * np.pi) + (z ** 2).sum(-1)) - half_logdet
        for _ in range(self._reinterpreted):
            lp = lp.sum(-1)
        return lp


def install(reference_root="/root/reference"):
    """Register the fake modules and put the reference on sys.path."""
    if "jax" in sys.modules and getattr(sys.modules["jax"], "__lqg_standin__", False):
        return
    jnp = types.ModuleType("jax.numpy")
    for name in dir(np):
        if not name.startswith("_"):
            setattr(jnp, name, getattr(np, name))
    jnp.ndarray = np.ndarray
    jnp.array = lambda a, dtype=None: _wrap(np.array(a, dtype=np.float64 if dtype is None else dtype))
    jnp.zeros = lambda shape, dtype=None: _wrap(np.zeros(shape, dtype=np.float64))
    jnp.eye = lambda n, m=None, k=0, dtype=None: _wrap(np.eye(n, m, k, dtype=np.float64))

    def _clip(x, min=None, max=None):
        return np.clip(x, min, max)

    jnp.clip = _clip
    jlinalg = types.ModuleType("jax.numpy.linalg")
    jlinalg.inv, jlinalg.solve, jlinalg.eigh = np.linalg.inv, np.linalg.solve, np.linalg.eigh
    jnp.linalg = jlinalg

    lax = types.ModuleType("jax.lax")
    lax.scan = scan

    jrandom = types.ModuleType("jax.random")
    jrandom.PRNGKey, jrandom.split, jrandom.normal = _prngkey, _split, _normal

    jsl = types.ModuleType("jax.scipy.linalg")
    jsl.block_diag = lambda *a: _wrap(sla.block_diag(*a))
    jsl.expm = lambda a: _wrap(sla.expm(a))
    jsl.cholesky = lambda a, lower=False: _wrap(sla.cholesky(a, lower=lower))
    jscipy = types.ModuleType("jax.scipy")
    jscipy.linalg = jsl

    jax = types.ModuleType("jax")
    jax.__lqg_standin__ = True
    jax.numpy, jax.lax, jax.random, jax.scipy = jnp, lax, jrandom, jscipy
    jax.vmap, jax.Array = vmap, np.ndarray

    dist = types.ModuleType("numpyro.distributions")
    dist.Distribution, dist.MultivariateNormal = Distribution, MultivariateNormal
    numpyro = types.ModuleType("numpyro")
    numpyro.distributions = dist

    sys.modules.update({
        "jax": jax, "jax.numpy": jnp, "jax.numpy.linalg": jlinalg, "jax.lax": lax,
        "jax.random": jrandom, "jax.scipy": jscipy, "jax.scipy.linalg": jsl,
        "numpyro": numpyro, "numpyro.distributions": dist,
    })

    _orig_version = importlib.metadata.version

    def _version(name):
        if name == "lqg":
            return "0.2.13"
        return _orig_version(name)

    importlib.metadata.version = _version
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
