"""scipy.optimize.minimize with gradients from torch.autograd — the role of lqg/optim.py:14-169 (`minimize`: the
reference wraps scipy with `jit(grad(fun))`, optim.py:142-147).

`fun(x, *args)` is written with torch ops on a parameter "tree" x (a tensor, or a dict / list / tuple of tensors or
floats) and returns a scalar tensor; for a likelihood it would build a model from x and return
`-model.log_likelihood(data).sum()`, which is differentiable through the HIP adjoint sweep (lqg_amd/grad.py).  With an
initialised process group and trial-sharded data, all-reduce inside `fun` is not needed: pass `group=` and the value and
gradient are summed over ranks here (one all-reduce of 1 + P numbers per evaluation, lqg_amd/dist.py).
"""
import numpy as np
import scipy.optimize
import torch


def _flatten(tree):
    """tree -> (list of leaf tensors, rebuild(list of tensors) -> tree)."""
    if isinstance(tree, torch.Tensor):
        return [tree], lambda leaves: leaves[0]
    if isinstance(tree, dict):
        keys = list(tree)
        parts = [_flatten(tree[k]) for k in keys]
        sizes = [len(p[0]) for p in parts]

        def rebuild(leaves):
            out, i = {}, 0
            for k, (_, rb), n in zip(keys, parts, sizes):
                out[k] = rb(leaves[i:i + n])
                i += n
            return out
        return [l for p in parts for l in p[0]], rebuild
    if isinstance(tree, (list, tuple)):
        parts = [_flatten(v) for v in tree]
        sizes = [len(p[0]) for p in parts]

        def rebuild(leaves):
            out, i = [], 0
            for (_, rb), n in zip(parts, sizes):
                out.append(rb(leaves[i:i + n]))
                i += n
            return type(tree)(out)
        return [l for p in parts for l in p[0]], rebuild
    return _flatten(torch.as_tensor(float(tree), dtype=torch.float64))


def minimize(fun, x0, method=None, args=(), bounds=None, constraints=(), tol=None, callback=None, options=None,
             group=None):
    """scipy.optimize.minimize(fun, x0, jac=autograd) on a tree of tensors; same arguments and result as the reference's
    wrapper (lqg/optim.py:14): `res.x` has the structure of x0, `res.fun`, `res.success`, `res.message` as scipy's."""
    leaves0, rebuild = _flatten(x0)
    shapes = [tuple(l.shape) for l in leaves0]
    sizes = [int(np.prod(s)) if s else 1 for s in shapes]
    dev = leaves0[0].device
    x0_flat = np.concatenate([l.detach().double().cpu().reshape(-1).numpy() for l in leaves0])

    def unravel(flat, requires_grad):
        out, i = [], 0
        for s, n, l0 in zip(shapes, sizes, leaves0):
            t = torch.as_tensor(np.asarray(flat[i:i + n]), dtype=torch.float64, device=dev).reshape(s)
            out.append(t.requires_grad_(requires_grad))
            i += n
        return out

    cache = {}

    def value_and_grad(flat, *a):
        key = flat.tobytes()
        if cache.get("key") != key:
            leaves = unravel(flat, True)
            val = fun(rebuild(leaves), *a)
            grads = torch.autograd.grad(val, leaves, allow_unused=True)
            vec = torch.cat([val.detach().double().reshape(1)] +
                            [(torch.zeros_like(l) if g is None else g).double().reshape(-1) for g, l in zip(grads, leaves)])
            if group is not None:
                from lqg_amd import dist as ld
                vec = ld.all_reduce_sum(vec, group=group)
            vec = vec.cpu().numpy()
            cache.update(key=key, val=float(vec[0]), grad=vec[1:].copy())
        return cache["val"], cache["grad"]

    def cb(flat, *a):
        if callback is not None:
            return callback(rebuild(unravel(flat, False)), *a)

    res = scipy.optimize.minimize(lambda f, *a: value_and_grad(f, *a)[0], x0_flat, args=args, method=method,
                                  jac=lambda f, *a: value_and_grad(f, *a)[1], callback=cb, bounds=bounds,
                                  constraints=constraints, tol=tol, options=options)
    res["x"] = rebuild(unravel(res["x"], False))
    return res
