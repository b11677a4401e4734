"""lqg.system — System / Dynamics / Actor / LQG with the reference's call surface (lqg/system.py:12-376),
computing on MI355X through liblqg_hip.so.

Shapes follow the reference exactly (`x[n, T+1, d]` in, `[n]` / `[n, T, ...]` out).  One extension: a spec
may carry a leading axis of B systems (parameter candidates); results then gain a leading `[B]` axis and
`x` may be `[n, T+1, d]` (the same trials scored under every candidate — the reference's
`vmap(ll)(sigmas)`, notebooks/Tutorial.ipynb cell 38) or `[B, n, T+1, d]`.
"""
import torch

from lqg_amd import _hip
from lqg_amd.belief import kf
from lqg_amd.control import lqr
from lqg_amd.spec import LQGSpec
from lqg_amd.utils import time_stack_spec


class TrajectoryNormal:
    """Gaussian over trajectories with per-step moments — the slice of numpyro's MultivariateNormal the
    reference uses (lqg/system.py:244,257): `.loc`, `.covariance_matrix`, `.log_prob`, `.shape()`,
    `.sample`, `.to_event`.  `Sigma` is stored once per system ([T,k,k]); `covariance_matrix` broadcasts it
    over trials without copying (under vmap the reference holds n identical copies)."""

    def __init__(self, loc, Sigma, lo, hi, event_dims=0, dtype=None):
        """dtype: the caller's dtype when the moments are held wider (System.conditional_distribution keeps the moments
        of an fp32 problem in fp64: a mean of magnitude ~50 rounded to fp32 loses the digits of an innovation of ~0.1, and
        log_prob(x[:, 1:]) — the reference's log_likelihood, lqg/system.py:246-248 — would miss 1e-6).  `.loc`,
        `.covariance_matrix`, `.sample` and the result of `.log_prob` are of that dtype."""
        self._mu, self._Sigma, self._lo, self._hi = loc, Sigma, lo, hi
        self._event_dims = event_dims
        self._dtype = loc.dtype if dtype is None else dtype

    @property
    def loc(self):
        return self._mu[..., self._lo:self._hi].to(self._dtype)

    mean = loc

    @property
    def covariance_matrix(self):
        S = self._Sigma[..., self._lo:self._hi, self._lo:self._hi].to(self._dtype)
        return S.unsqueeze(-4).expand(*self._mu.shape[:-1], S.shape[-2], S.shape[-1])

    def to_event(self, n=1):
        return TrajectoryNormal(self._mu, self._Sigma, self._lo, self._hi, self._event_dims + n, self._dtype)

    def shape(self, sample_shape=()):
        return tuple(sample_shape) + tuple(self.loc.shape)

    @property
    def batch_shape(self):
        s = tuple(self.loc.shape[:-1])
        return s[:len(s) - self._event_dims]

    @property
    def event_shape(self):
        s = tuple(self.loc.shape)
        return s[len(s) - 1 - self._event_dims:]

    def log_prob(self, value):
        """sum over time of log N(value_t; loc_t, cov_t): numpyro MultivariateNormal(...).to_event(1).log_prob
        (lqg/system.py:244,248).  Runs lqg_gaussian_logprob on the GPU."""
        if self._event_dims != 1:
            raise NotImplementedError("per-step log_prob: use .to_event(1) (the only form the reference uses)")
        k = self._hi - self._lo
        if self._lo == 0:
            return _hip.gaussian_logprob(value, self._mu, self._Sigma, k).to(self._dtype)
        mu = self._mu[..., self._lo:self._hi].contiguous()
        S = self._Sigma[..., self._lo:self._hi, self._lo:self._hi].contiguous()
        return _hip.gaussian_logprob(value, mu, S, k).to(self._dtype)

    def sample(self, key=None, sample_shape=()):
        """Draw from the per-step Gaussians (torch RNG; `key` = int seed or torch.Generator)."""
        gen = _generator(key, self._mu.device)
        S = self._Sigma[..., self._lo:self._hi, self._lo:self._hi].to(self._dtype)
        Lc = torch.linalg.cholesky(S)
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        z = torch.randn(shape, dtype=self._dtype, device=self._mu.device, generator=gen)
        return self.loc + torch.einsum("...tij,...ntj->...nti", Lc, z)


def _generator(key, device):
    if key is None or isinstance(key, torch.Generator):
        return key
    g = torch.Generator(device=device)
    g.manual_seed(int(key))
    return g


class System:
    def __init__(self, actor: LQGSpec, dynamics: LQGSpec):
        self.actor = actor
        self.dynamics = dynamics

    # ---- dimensions (lqg/system.py:16-60)
    @property
    def T(self):
        """Length of trajectory: number of time steps"""
        return self.dynamics.A.shape[-3]

    @property
    def xdim(self):
        """State dimensionality"""
        return self.dynamics.A.shape[-1]

    @property
    def ydim(self):
        """Observation dimensionality"""
        return self.dynamics.F.shape[-2]

    @property
    def bdim(self):
        """Belief dimensionality"""
        return self.actor.A.shape[-1]

    @property
    def udim(self):
        """Action dimensionality"""
        return self.dynamics.B.shape[-1]

    @property
    def n_systems(self):
        """Number of systems (parameter candidates) on the leading axis, or None when unbatched."""
        for sp in (self.actor, self.dynamics):
            if sp.A.dim() == 4:
                return sp.A.shape[0]
        return None

    def to(self, *args, **kwargs):
        """Move / cast both specs (torch `.to` semantics)."""
        from lqg_amd.utils import mark_zero

        def move(t):
            # keep stride-0 (expanded) axes expanded: convert one slice and re-expand, never materialise T copies
            idx = tuple(slice(0, 1) if (st == 0 and sz > 1) else slice(None) for st, sz in zip(t.stride(), t.shape))
            return t[idx].to(*args, **kwargs).expand(t.shape)

        def conv(spec):
            out = {}
            for f in LQGSpec._fields:
                t = getattr(spec, f)
                t2 = move(t)
                if getattr(t, "_lqg_zero", False):
                    mark_zero(t2)
                out[f] = t2
            return LQGSpec(**out)

        new = System.__new__(type(self))
        new.__dict__.update(self.__dict__)
        new.__dict__.pop("_lqg_decouple", None)      # cached sub-systems hold tensors of the old dtype / device
        same = self.actor is self.dynamics
        new.__dict__["actor"] = conv(self.actor)     # (not through __setattr__: a cast keeps the structure)
        new.__dict__["dynamics"] = new.__dict__["actor"] if same else conv(self.dynamics)
        return new

    def __setattr__(self, name, value):
        """Replacing a spec AFTER construction voids everything derived from the old one: the class-level structure of a
        zoo model (sparsity pattern, identical axes) and the cached decoupling / specialisation decisions."""
        if name in ("actor", "dynamics") and name in self.__dict__:
            for k in ("_zoo_structure", "_lqg_decouple", "_lqg_patterns", "_lqg_groups"):
                self.__dict__.pop(k, None)
        object.__setattr__(self, name, value)

    # ---- simulation (lqg/system.py:62-140)
    def simulate(self, rng_key=None, n=1, x0=None, xhat0=None, Sigma0=None, return_all=False):
        """Simulate n trials.

        rng_key: int seed or torch.Generator (the reference takes a jax PRNGKey; the stream of normal draws
        necessarily differs, the recursion does not).  An int seed (or None = 0) draws IN THE KERNEL (counter-based
        Philox keyed by the seed, counter = (trial, system, step, block), csrc/lqg_rng.hpp: trial k of system s gets the same
        draws whatever the number of systems or trials around it — shards of one study on different ranks must therefore
        use different seeds — and no [n, T, x + y] noise arrays pass through HBM; fp64 problems get fp32 normals, widened);
        a torch.Generator supplies the draws from torch's stream.
        Returns x[n, T+1, xdim]; with return_all also x_hat[n, T+1, bdim], y[n, T, ydim], u[n, T, udim]."""
        # (the gains only travel from the two sweeps to the simulate kernel: system-fastest storage, coalesced both ways)
        L_, l_, H_ = _hip.riccati_backward(self.actor, system_fastest=True)
        gains = lqr.Gains(L=L_, l=l_, H=H_)
        K = _hip.kalman_forward(self.actor, Sigma0=Sigma0, system_fastest=True)
        dev, dt = self.actor.A.device, self.actor.A.dtype
        if not isinstance(rng_key, torch.Generator):
            x, x_hat, y, u = _hip.simulate(self.actor, self.dynamics, gains.L, gains.l, K, x0=x0, xhat0=xhat0,
                                           return_all=return_all, seed=0 if rng_key is None else int(rng_key), n=n)
            return (x, x_hat, y, u) if return_all else x
        gen = _generator(rng_key, dev)
        lead = () if self.n_systems is None else (self.n_systems,)
        eps = torch.randn(lead + (n, self.T, self.xdim), dtype=dt, device=dev, generator=gen)
        eta = torch.randn(lead + (n, self.T, self.ydim), dtype=dt, device=dev, generator=gen)
        x, x_hat, y, u = _hip.simulate(self.actor, self.dynamics, gains.L, gains.l, K, eps, eta, x0=x0, xhat0=xhat0,
                                       return_all=return_all)
        if return_all:
            return x, x_hat, y, u
        return x

    # ---- likelihood (lqg/system.py:142-257)
    def conditional_moments(self, x, Sigma0=None):
        """p(x_{t+1}, xhat_{t+1} | x_{1:t}) for ONE trajectory x[T+1, d] -> mu[T, m], Sigma[T, m, m]
        (lqg/system.py:142-235).  With B systems: x[B, T+1, d] -> mu[B, T, m], Sigma[B, T, m, m]."""
        mu, Sig = self._moments(x.unsqueeze(-3), Sigma0)
        return mu.squeeze(-3), Sig

    def _moments(self, x, Sigma0, keep_wide=False):
        """mu[(B,) n, T, m], Sigma[(B,) T, m, m].  An fp32 problem is evaluated over an fp64 image of its specs and data when
        its observed noise block is ill-conditioned (plan.F32_MAX_COND) or when the caller keeps the moments wide
        (keep_wide: conditional_distribution); the moments are rounded to fp32 unless keep_wide."""
        if self.actor.A.dtype == torch.float32 and self.actor.A.is_cuda:
            from lqg_amd import plan
            if keep_wide or plan.f32_needs_wide(self, x.shape[-1]):
                wide = self.to(torch.float64)
                S0 = None if Sigma0 is None else Sigma0.to(torch.float64)
                mu, Sig = _hip.conditional_moments(wide.actor, wide.dynamics, x.to(torch.float64), Sigma0=S0, system=wide)
                return (mu, Sig) if keep_wide else (mu.to(torch.float32), Sig.to(torch.float32))
        return _hip.conditional_moments(self.actor, self.dynamics, x, Sigma0=Sigma0, system=self)

    def conditional_distribution(self, x, Sigma0=None):
        """x[n, T+1, d] -> Gaussian over x[:, 1:] with event shape (T, d) (lqg/system.py:237-244).  (fp32: the moments are
        held in fp64 inside the distribution, TrajectoryNormal.__init__.)"""
        d = x.shape[-1]
        mu, Sig = self._moments(x, Sigma0, keep_wide=True)
        return TrajectoryNormal(mu, Sig, 0, d, dtype=self.actor.A.dtype).to_event(1)

    def log_likelihood(self, x, Sigma0=None):
        """log p(x_{1:T} | x_0) per trial: x[n, T+1, d] -> [n] (lqg/system.py:246-248).  Fused HIP path."""
        from lqg_amd import grad
        if grad.needs_grad(self, Sigma0):              # jax.grad(ll): reverse-mode HIP sweep behind torch.autograd
            return grad.log_likelihood(self, x, Sigma0)
        from lqg_amd.plan import LogLikelihoodPlan
        return LogLikelihoodPlan(self, x, Sigma0=Sigma0).run()

    def decoupled(self, d, Sigma0=None, for_grad=False, eps=1e-8):
        """Independent components of this system for data with d observed dims ([(sub_system, data columns, belief
        dims)], lqg_amd/decouple.py), or None when it does not decouple (or LQG_NO_DECOUPLE=1).  for_grad: the
        differentiable evaluation's variant (decouple.plan)."""
        from lqg_amd import _abi, decouple, options
        if options.flag("NO_DECOUPLE"):
            return None
        parts = decouple.plan(self, d, Sigma0, for_grad=for_grad)
        if parts is None:
            return None
        if not decouple.floor_provably_inactive(self, eps):
            return None      # an active eigenvalue floor (lqr.py:27-28) couples the components: solve the joint problem
        # every component must be solvable by a generic library too (several trials, moments, ...)
        try:
            ok = all(_abi.shape_available(sub.xdim, sub.bdim, sub.udim, sub.ydim, len(cols)) for sub, cols, _ in parts)
        except _abi.LqgHipError:
            return parts
        if not ok:
            return None
        return parts

    def belief_tracking_distribution(self, x, Sigma0=None):
        """Distribution of the actor's belief given the observed trajectory (lqg/system.py:250-257):
        batch shape (n, T), event shape (bdim,)."""
        d = self.xdim
        mu, Sig = self._moments(x, Sigma0)
        return TrajectoryNormal(mu, Sig, d, d + self.bdim)

    def to_numpyro(self, Sigma0=None, xdim=None):
        return TrajectoryLQG(self, Sigma0=Sigma0, xdim=xdim)


class TrajectoryLQG:
    """The reference's NumpyroLQG adapter (lqg/system.py:358-376) without the NumPyro base class
    (NumPyro is not installed here): same `log_prob` / `sample` / `event_shape` protocol."""

    def __init__(self, system: System, xdim=None, Sigma0=None):
        self.system = system
        self.Sigma0 = Sigma0
        xdim = system.xdim if xdim is None else xdim
        self.event_shape = (system.T + 1, xdim)
        self.batch_shape = ()

    def log_prob(self, x):
        return self.system.log_likelihood(x, Sigma0=self.Sigma0)

    def sample(self, key, sample_shape=()):
        if len(sample_shape) == 0:
            return self.system.simulate(key, n=1, Sigma0=self.Sigma0)[0]
        return self.system.simulate(key, n=sample_shape[0], Sigma0=self.Sigma0)


def Dynamics(A, B, F, V, W, T=1000):
    """lqg/system.py:331-344: a spec whose cost matrices Q, R are zero."""
    A = torch.as_tensor(A)
    B = torch.as_tensor(B, dtype=A.dtype, device=A.device)
    xdim, udim = A.shape[-1], B.shape[-1]
    return time_stack_spec(A=A, B=B, F=F, V=V, W=W, Q=torch.zeros((xdim, xdim), dtype=A.dtype, device=A.device),
                           R=torch.zeros((udim, udim), dtype=A.dtype, device=A.device), T=T)


def Actor(A, B, F, V, W, Q, R, T=1000):
    """lqg/system.py:347-348"""
    return time_stack_spec(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)


class LQG(System):
    """lqg/system.py:351-355: actor model == true dynamics."""

    def __init__(self, A, B, F, V, W, Q, R, T=1000):
        spec = time_stack_spec(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
        super().__init__(actor=spec, dynamics=spec)
