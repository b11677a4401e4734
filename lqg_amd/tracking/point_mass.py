"""PointMassBoundedActor — mirrors lqg/tracking/point_mass.py:7-144: a damped point mass driven through a
first-order muscle filter, discretised exactly (zero-order hold via the matrix exponential, point_mass.py:50-79)
with Van Loan process-noise discretisation (:82-110) made positive definite by eigenvalue clipping (:130-144)."""
import torch

from lqg_amd import options
from lqg_amd.system import Actor, System
from lqg_amd.tracking import _build as bd


def discretize_linear_system(A, B, dt):
    """x' = A x + B u  ->  (Ad, Bd) by expm([[A, B],[0, 0]] dt)              point_mass.py:50-79"""
    n, m = A.shape[-1], B.shape[-1]
    M = torch.zeros(*A.shape[:-2], n + m, n + m, dtype=A.dtype, device=A.device)
    M[..., :n, :n] = A
    M[..., :n, n:] = B
    E = torch.linalg.matrix_exp(M * dt)
    return E[..., :n, :n], E[..., :n, n:]


def van_loan_discretization(A, G, dt, Qc=None):
    """Discrete process-noise covariance block of expm([[A, G Qc G^T],[0, -A^T]] dt)   point_mass.py:82-110"""
    n = A.shape[-1]
    if Qc is None:
        Qc = torch.eye(G.shape[-1], dtype=A.dtype, device=A.device)
    Q = G @ Qc @ G.transpose(-1, -2)
    M = torch.zeros(*A.shape[:-2], 2 * n, 2 * n, dtype=A.dtype, device=A.device)
    M[..., :n, :n] = A
    M[..., :n, n:] = Q
    M[..., n:, n:] = -A.transpose(-1, -2)
    return torch.linalg.matrix_exp(M * dt)[..., :n, n:]


def make_psd(M, eps=1e-6):
    """Symmetrise and clip eigenvalues from below                              point_mass.py:130-144"""
    Ms = (M + M.transpose(-1, -2)) / 2
    w, U = torch.linalg.eigh(Ms)
    return U @ torch.diag_embed(w.clamp(min=eps)) @ U.transpose(-1, -2)


def _setup_on_gpu(damping, m, tau, action_variability, dt, eps=1e-6):
    """The same matrices from ONE kernel (csrc/lqg_setup.hip, `lqg_point_mass_setup`): no rocSOLVER call, no host
    synchronisation, capturable into a hipGraph.  No-grad route only."""
    import ctypes as C
    from lqg_amd import _abi
    lib = _abi.load()
    lead = damping.shape
    n = max(1, damping.numel())
    vec = lambda t: t.to(torch.float64).expand(lead).reshape(n).contiguous()
    dmp, mm, ta, av = vec(damping), vec(m), vec(tau), vec(action_variability)
    A = torch.empty(n, 3, 3, dtype=torch.float64, device=damping.device)
    B = torch.empty(n, 3, 1, dtype=torch.float64, device=damping.device)
    V = torch.empty(n, 3, 3, dtype=torch.float64, device=damping.device)
    with torch.cuda.device(damping.device):
        _abi.check(lib.lqg_point_mass_setup(n, dmp.data_ptr(), mm.data_ptr(), ta.data_ptr(), av.data_ptr(), float(dt), float(eps),
                                            A.data_ptr(), B.data_ptr(), V.data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream(damping.device).cuda_stream)),
                   "lqg_point_mass_setup")
    shape = tuple(lead)
    return A.reshape(shape + (3, 3)), B.reshape(shape + (3, 1)), V.reshape(shape + (3, 3))


def point_mass_dynamics_matrices(damping, m, tau, action_variability, dt):
    """Continuous point mass + muscle filter, discretised                      point_mass.py:113-127
    (setup arithmetic is done in float64 on the host device of the parameters and cast by the caller)."""
    import os
    if damping.is_cuda and not any(t.requires_grad for t in (damping, m, tau, action_variability)) \
            and isinstance(dt, float) and options.flag("SETUP_KERNEL"):
        return _setup_on_gpu(damping, m, tau, action_variability, dt)
    z, o = torch.zeros_like(damping), torch.ones_like(damping)
    A_c = torch.stack([torch.stack([z, o, z], -1), torch.stack([z, -damping / m, o / m], -1),
                       torch.stack([z, z, -o / tau], -1)], -2)
    B_c = torch.stack([z, z, o / tau], -1).unsqueeze(-1)
    A, B = discretize_linear_system(A_c, B_c, dt)
    G = (1e-2 * action_variability)[..., None, None] * B_c
    # jax.scipy.linalg.cholesky defaults to the UPPER factor (point_mass.py:123)
    V = torch.linalg.cholesky(make_psd(van_loan_discretization(A_c, G, dt)), upper=True)
    return A, B, V


class PointMassBoundedActor(System):
    def __init__(self, process_noise=1.0, action_variability=1e-3, sigma_target=6.0, sigma_cursor=6.0,
                 action_cost=0.01, dt=1.0 / 60.0, T=1000, damping=0.1, m=1.0, tau=0.0015, device=None, dtype=None):
        device, dtype = bd.resolve(device, dtype, process_noise, action_variability, sigma_target, sigma_cursor,
                                   action_cost, damping, m, tau)
        # matrix exponential / eigh / Cholesky of the setup run in float64 (host-side, once per candidate)
        (pn, av, st, sc, ac, dmp, mm, ta), lead = bd.params(device, torch.float64, process_noise, action_variability,
                                                            sigma_target, sigma_cursor, action_cost, damping, m, tau)
        A3, B3, V3 = point_mass_dynamics_matrices(damping=dmp, m=mm, tau=ta, action_variability=av, dt=dt)
        one = bd.const([[1.0]], lead, device, torch.float64)
        A = bd.block_diag(one, A3)                 # target position as a constant state   point_mass.py:24
        B = torch.cat([torch.zeros_like(B3[..., :1, :]), B3], dim=-2)
        V = bd.block_diag(bd.diag([pn]), V3)
        F = bd.const([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0]], lead, device, torch.float64)   # eye(3, 4)
        W = bd.diag([st, sc, sc])
        Q = bd.const([[1.0, -1.0, 0.0, 0.0], [-1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]],
                     lead, device, torch.float64)
        R = bd.diag([ac * dt])
        A, B, V, F, W, Q, R = (t.to(dtype) for t in (A, B, V, F, W, Q, R))
        spec = Actor(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
        super().__init__(actor=spec, dynamics=spec)
        self._zoo_structure = {}
