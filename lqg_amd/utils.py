"""time_stack / time_stack_spec — mirrors lqg/utils.py:6-35.

The reference materialises T copies of every constant matrix (`jnp.stack((A,) * T)`); here the time axis
is a stride-0 `expand` view, so a time-invariant spec costs one copy of each matrix in HBM and the
kernels see `st == 0` (load once per system).  Shapes and values are identical to the reference's.
"""
import torch

from lqg_amd.spec import LQGSpec


def _as_tensor(a, like=None):
    if isinstance(a, torch.Tensor):
        return a
    dtype = like.dtype if like is not None else torch.get_default_dtype()
    device = like.device if like is not None else None
    return torch.as_tensor(a, dtype=dtype, device=device)


def mark_zero(t):
    """Tag a tensor as known-all-zero so that the launch glue can pass NULL (lqg_hip.h: 'ptr NULL = zero')."""
    t._lqg_zero = True
    return t


def time_stack(A: torch.Tensor, T: int):
    """lqg/utils.py:6-7.  A[r, c] -> [T, r, c] (or [B, r, c] -> [B, T, r, c]) as a stride-0 view."""
    out = A.unsqueeze(-3).expand(*A.shape[:-2], T, *A.shape[-2:])
    out._lqg_base = A                  # lets lqg_amd.grad differentiate w.r.t. the matrix without a T-fold zero-fill
    return out


def time_stack_spec(A, B, F, V, W, Q, R, T: int) -> LQGSpec:
    """lqg/utils.py:10-35: replicate the constant matrices T times; q, P, r = 0; Qf = Q[-1]; qf = 0.

    Each matrix may carry a leading axis of systems ([B, r, c]); the result then has [B, T, r, c] fields."""
    A = _as_tensor(A)
    B, F, V, W, Q, R = (_as_tensor(m, A) for m in (B, F, V, W, Q, R))
    state_dim, action_dim = Q.shape[-1], R.shape[-1]
    z = torch.zeros((), dtype=A.dtype, device=A.device)
    q = mark_zero(z.expand(T, state_dim))
    qf = mark_zero(z.expand(state_dim))
    P = mark_zero(z.expand(T, action_dim, state_dim))
    r = mark_zero(z.expand(T, action_dim))
    return LQGSpec(A=time_stack(A, T), B=time_stack(B, T), F=time_stack(F, T), V=time_stack(V, T),
                   W=time_stack(W, T), Q=time_stack(Q, T), R=time_stack(R, T), q=q, Qf=Q, qf=qf, P=P, r=r)
