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


# Python scalars and constant matrices as device tensors, kept per (value, dtype, device): a host-to-device copy per
# constant per constructor call costs ~10 us each, and is not allowed at all while a hipGraph is being captured
# (lqg_amd/infer/graphed.py captures the whole constructor).  Entries are never handed out for writing: scalars are only
# read inside the constructors, constant matrices are cloned.
_CACHE_MAX = 512
_scalar_cache, _const_cache = {}, {}


def cached_tensors():
    """Every tensor the two caches hold right now.  A captured hipGraph of a constructor reads these by ADDRESS: the
    evaluator keeps this list alive for as long as its graph exists (lqg_amd/infer/graphed.py), so that a later wholesale
    clear() of a full cache drops the dictionary entries but never frees memory a graph still replays from."""
    return list(_scalar_cache.values()) + list(_const_cache.values())


def _scalar(v, dtype, device):
    if isinstance(v, torch.Tensor):
        return torch.as_tensor(v, dtype=dtype, device=device)
    key = (float(v), dtype, str(device))
    t = _scalar_cache.get(key)
    if t is None:
        if len(_scalar_cache) >= _CACHE_MAX:
            _scalar_cache.clear()
        t = _scalar_cache[key] = torch.as_tensor(float(v), dtype=dtype, device=device)
    return t


def params(device, dtype, *values):
    """Scalars / [B] tensors -> list of tensors broadcast to a common shape () or [B]."""
    ts = [_scalar(v, dtype, device) for v in values]
    shape = torch.broadcast_shapes(*[t.shape for t in ts])
    if len(shape) > 1:
        raise ValueError(f"model parameters must be scalars or 1-D candidate vectors, got shape {tuple(shape)}")
    return [t.expand(shape) for t in ts], tuple(shape)


def const(rows, lead, device, dtype):
    """Constant matrix from nested lists, broadcast over the candidate axis (stride 0)."""
    key = (repr(rows), dtype, str(device))
    master = _const_cache.get(key)
    if master is None:
        if len(_const_cache) >= _CACHE_MAX:
            _const_cache.clear()
        master = _const_cache[key] = torch.tensor(rows, dtype=dtype, device=device)
    m = master.clone()
    return m.expand(*lead, *m.shape) if lead else m


def take(t, idx, axis):
    """t[..., idx, :] (axis = -2) or t[..., :, idx] (axis = -1) for a Python list of indices: index_select with a cached
    device index tensor (indexing with the list itself builds the index on the host and copies it over on every call)."""
    key = ("idx", tuple(idx), str(t.device))
    i = _const_cache.get(key)
    if i is None:
        i = _const_cache[key] = torch.tensor(list(idx), dtype=torch.long, device=t.device)
    return torch.index_select(t, axis, i)


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
