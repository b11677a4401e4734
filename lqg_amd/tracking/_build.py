"""Batched matrix builders shared by the model zoo.

Every model parameter may be a Python scalar or a 1-D tensor of B candidate values; matrices come out as
[r, c] or [B, r, c] accordingly (what the reference obtains by tracing the constructor under jax.vmap,
notebooks/Tutorial.ipynb cell 38).  This is O(n^2) host-side setup per candidate, not part of the hot path.
"""
import torch


def default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def resolve(device, dtype, *params):
    """Pick device/dtype: explicit > taken from the first tensor parameter > (cuda if present, float32)."""
    for p in params:
        if isinstance(p, torch.Tensor):
            device = p.device if device is None else device
            dtype = (p.dtype if p.dtype.is_floating_point else None) if dtype is None else dtype
            break
    device = default_device() if device is None else torch.device(device)
    dtype = torch.get_default_dtype() if dtype is None else dtype
    return device, dtype


def params(device, dtype, *values):
    """Scalars / [B] tensors -> list of tensors broadcast to a common shape () or [B]."""
    ts = [torch.as_tensor(v, dtype=dtype, device=device) for v in values]
    shape = torch.broadcast_shapes(*[t.shape for t in ts])
    if len(shape) > 1:
        raise ValueError(f"model parameters must be scalars or 1-D candidate vectors, got shape {tuple(shape)}")
    return [t.expand(shape) for t in ts], tuple(shape)


def const(rows, lead, device, dtype):
    """Constant matrix from nested lists, broadcast over the candidate axis (stride 0)."""
    m = torch.tensor(rows, dtype=dtype, device=device)
    return m.expand(*lead, *m.shape) if lead else m


def diag(entries):
    """[..., k] stacked diagonal entries -> [..., k, k]."""
    v = torch.stack(list(entries), dim=-1)
    return torch.diag_embed(v)


def block_diag(*mats):
    """Block-diagonal of matrices with common leading (candidate) shape."""
    lead = torch.broadcast_shapes(*[m.shape[:-2] for m in mats])
    rows = sum(m.shape[-2] for m in mats)
    cols = sum(m.shape[-1] for m in mats)
    out = torch.zeros(*lead, rows, cols, dtype=mats[0].dtype, device=mats[0].device)
    r = c = 0
    for m in mats:
        out[..., r:r + m.shape[-2], c:c + m.shape[-1]] = m
        r += m.shape[-2]
        c += m.shape[-1]
    return out
