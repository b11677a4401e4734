"""Multi-GPU sharding of the objective: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The path shards embarrassingly: every (candidate, trial) pair is independent, so there is no
collective inside the sweeps.  The single exchange is the objective the reference's inference drivers minimise —
sum over trials of the log-likelihood per candidate (lqg/infer/models.py:34; notebooks/Tutorial.ipynb cell 38;
the scalar `fun` of lqg/optim.py:137-147): per-rank fp64 partial sums (lqg_sum_trials, fixed reduction tree) are
all-reduced once per evaluation.  The message is C x 8 bytes (8 B ... 32 kB): latency-bound, ring/bandwidth
considerations do not apply.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_trials(x, rank=None, world=None, group=None):
    """This rank's block of trials of x[n, T+1, d] (a view)."""
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    lo, hi = shard_bounds(x.shape[-3], rank, world)
    return x[..., lo:hi, :, :]


def _default_local_sum(system, x_local, Sigma0=None):
    from lqg_amd import _hip
    ll = system.log_likelihood(x_local, Sigma0=Sigma0)          # HIP path, [(B,) n_local]
    return _hip.sum_trials(ll)                                   # fp64 [(B,)]


def log_likelihood_sum(system, x_local, Sigma0=None, group=None, local_sum=None):
    """Objective sum_n log p(x_n | theta_c) for every candidate c, trials sharded over the ranks of `group`.

    x_local is THIS rank's shard of the trials ([n_local, T+1, d]); every rank holds all candidates (specs are a
    few kB).  Returns fp64 [(B,)] identical on all ranks.  `local_sum(system, x_local, Sigma0)` computes the
    per-rank partial sums; the default runs the HIP path, the CPU tests inject the oracle."""
    part = (local_sum or _default_local_sum)(system, x_local, Sigma0)
    part = torch.as_tensor(part, dtype=torch.float64)
    if x_local.shape[-3] == 0:
        part = torch.zeros_like(part)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def all_reduce_sum(t, group=None):
    """Sum a small fp64 vector (objective and its gradient) over the ranks of `group`; identity without a group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def gather_candidates(local_values, n_total, group=None):
    """Candidate-sharded evaluation: every rank scored its block of candidates (shard_bounds) against all trials;
    concatenate the [n_local] blocks into [n_total] on every rank (all_gather of padded blocks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_values
    world = dist.get_world_size(group)
    width = (n_total + world - 1) // world
    pad = torch.zeros(width, dtype=local_values.dtype, device=local_values.device)
    pad[: local_values.shape[0]] = local_values
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts)
