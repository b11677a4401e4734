"""CPU, world_size 2, gloo: the multi-GPU sharding logic (trial-sharded objective with one all-reduce;
candidate-sharded evaluation with one all-gather).  The per-rank evaluation is injected with the CPU oracle —
on the GPU box the same functions run the HIP path per rank (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as OC
    from lqg_amd import dist as ld

    g, actor, dyn = load_golden("subjective1d_T50")
    x = np.concatenate([g["x"], g["x"][::-1] * 0.9, g["x"] * 1.1])           # 9 trials
    # three candidates: scale the observation noise
    scales = np.array([1.0, 1.5, 2.0])
    ab = {k: np.broadcast_to(v, (3,) + v.shape) for k, v in actor.items()}
    db = {k: np.broadcast_to(v, (3,) + v.shape) for k, v in dyn.items()}
    ab["W"] = np.ascontiguousarray(ab["W"]) * scales[:, None, None, None]
    db["W"] = np.ascontiguousarray(db["W"]) * scales[:, None, None, None]

    def local_sum(system, x_local, Sigma0):
        if x_local.shape[0] == 0:
            return torch.zeros(3, dtype=torch.float64)
        xl = np.broadcast_to(x_local.numpy(), (3,) + tuple(x_local.shape))
        return torch.from_numpy(OC.log_likelihood(ab, db, xl).sum(-1))

    xt = torch.from_numpy(x)
    mine = ld.shard_trials(xt)
    total = ld.log_likelihood_sum(None, mine, local_sum=local_sum)
    full = OC.log_likelihood(ab, db, np.broadcast_to(x, (3,) + x.shape)).sum(-1)
    assert np.allclose(total.numpy(), full, rtol=1e-12), (total, full)

    # candidate sharding: each rank scores its block of candidates on all trials, then all-gather
    lo, hi = ld.shard_bounds(3, rank, world)
    a_loc = {k: v[lo:hi] for k, v in ab.items()}
    d_loc = {k: v[lo:hi] for k, v in db.items()}
    loc = OC.log_likelihood(a_loc, d_loc, np.broadcast_to(x, (hi - lo,) + x.shape)).sum(-1) if hi > lo else np.zeros(0)
    allv = ld.gather_candidates(torch.from_numpy(np.ascontiguousarray(loc)), 3)
    assert np.allclose(allv.numpy(), full, rtol=1e-12)
    # objective + gradient with the trials sharded (lqg_amd.infer.value_and_grad, method="adjoint"): every rank
    # differentiates its shard (here with the NumPy restatement of the adjoint sweep), ONE all-reduce of [1 + P] numbers
    import lqg_adjoint_np as ADJ
    xm = mine.numpy()
    if xm.shape[0]:
        ll_loc, ga, gd, _ = ADJ.loglik_grad(actor, dyn, xm)
        vec = torch.tensor([ll_loc.sum(), ga["W"].sum(0)[0, 0], gd["V"].sum(0)[1, 1]], dtype=torch.float64)
    else:
        vec = torch.zeros(3, dtype=torch.float64)
    vec = ld.all_reduce_sum(vec)
    ll_all, ga, gd, _ = ADJ.loglik_grad(actor, dyn, x)
    ref = np.array([ll_all.sum(), ga["W"].sum(0)[0, 0], gd["V"].sum(0)[1, 1]])
    assert np.allclose(vec.numpy(), ref, rtol=1e-10), (vec, ref)
    # lqg_amd.optim.minimize with a process group (row a12): each rank holds a shard of the objective's data; value and
    # gradient are all-reduced inside the driver, so every rank walks the same L-BFGS path to the same optimum
    from lqg_amd.optim import minimize
    data = torch.arange(1.0, 9.0, dtype=torch.float64)                       # 8 "trials"
    lo8, hi8 = ld.shard_bounds(8, rank, world)
    mine8 = data[lo8:hi8]
    res = minimize(lambda p: ((mine8 - p["m"]) ** 2).sum() + 0.0 * p["s"], dict(m=torch.zeros((), dtype=torch.float64),
                   s=torch.ones((), dtype=torch.float64)), method="L-BFGS-B", group=dist.group.WORLD)
    assert abs(float(res.x["m"]) - float(data.mean())) < 1e-6 and abs(res.fun - float(((data - data.mean()) ** 2).sum())) < 1e-8
    # the (Nc, N, T, d) shared-parameter objective (lqg/infer/models.py:67-130) with the TRIAL axis split over the ranks:
    # every rank scores its shard of each condition's trials (the C oracle injected as the likelihood, CPU models), the
    # [C, Nc] table of partial sums is all-reduced once and equals the single-process value
    import lqg_amd
    from lqg_amd.infer import shared_params_objective
    rng = np.random.default_rng(3)
    xc = torch.from_numpy(np.cumsum(rng.standard_normal((2, 5, 41, 2)), axis=2))          # 2 conditions x 5 trials

    def oracle_ll(model, data):
        from lqg_amd import workload
        B_ = model.n_systems

        def f(spec):                       # every field with the system axis (the zoo shares what does not depend on a parameter)
            out = {}
            for k in lqg_amd.LQGSpec._fields:
                t = getattr(spec, k)
                if B_ is not None and t.dim() < workload._batched_ndim(k):
                    t = t.expand(B_, *t.shape)
                out[k] = np.ascontiguousarray(t.double().numpy())
            return out
        a_, d_ = f(model.actor), f(model.dynamics)
        dn = data.numpy()
        if dn.shape[-3] == 0:
            return torch.zeros(a_["A"].shape[:-3] + (0,), dtype=torch.float64)
        if a_["A"].ndim == 4 and dn.ndim == 3:
            dn = np.broadcast_to(dn, (a_["A"].shape[0],) + dn.shape)
        return torch.from_numpy(np.asarray(OC.log_likelihood(a_, d_, np.ascontiguousarray(dn))))

    pars = dict(sigma_target=torch.tensor([6.0, 12.0], dtype=torch.float64), action_cost=torch.tensor([0.1, 0.5, 2.0], dtype=torch.float64))
    kw = dict(shared_params=["action_cost", "action_variability", "sigma_cursor"], per_condition=True, _log_likelihood=oracle_ll)
    lo5, hi5 = ld.shard_bounds(5, rank, world)
    part = shared_params_objective(xc[:, lo5:hi5], lqg_amd.BoundedActor, pars, group=dist.group.WORLD, **kw)
    # (with a process group initialised every evaluation is all-reduced over it: all data on both ranks counts twice)
    whole = shared_params_objective(xc, lqg_amd.BoundedActor, pars, **kw) / world
    assert part.shape == (3, 2) and torch.allclose(part, whole, rtol=1e-12), (part, whole)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), total.numpy())
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from lqg_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 1024, 1067):
        for w in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 8])
def test_trial_and_candidate_sharding(tmp_path, oracle_lib, world):
    """world 2 and world 8 (the node the driver scales to).  Nothing divides evenly at 8: 9 trials -> shards of 2,1,1,...;
    3 candidates and the 5 trials per condition -> EMPTY shards on the upper ranks (they contribute zeros to the all-reduce
    and zero-length blocks to the all-gather); the result equals the single-process value on every rank."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    assert all(np.array_equal(rs[0], r) for r in rs[1:])            # identical on every rank after the all-reduce


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` run bare must start two ranks as child processes (round-1 finding: it ran one rank and
    printed n_gpus 1).  Here there is no GPU: both children must fail loudly (no CPU fallback), each naming its rank
    environment, and the parent must relay the failure — which also shows the parent never touched the GPU itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    assert out.count("bench.py needs an MI355X") >= 2 or "local_rank: 1" in out or "rank      : 1" in out, out[-2000:]


def _host_reduce_worker(rank, world, port, out_dir):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    hr = bench.HostReduce(torch, dist)
    assert hr.get_world_size() == world
    t = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
    hr.all_reduce(t)                                         # in place, like dist.all_reduce on a device tensor
    assert t.tolist() == [sum(1.0 + r for r in range(world)), sum(10.0 * (r + 1) for r in range(world))]
    mine = torch.tensor([float(rank) + 0.25], dtype=torch.float64)
    outs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    hr.all_gather(outs, mine)
    assert [float(o) for o in outs] == [r + 0.25 for r in range(world)]
    hr.barrier()
    np.save(os.path.join(out_dir, f"hr{rank}.npy"), t.numpy())
    hr.destroy_process_group()


def test_share_gpu_host_reduce_shim_world2(tmp_path):
    """bench.py --share-gpu routes its three collectives through gloo on host tensors (bench.HostReduce) with the call
    shapes of the RCCL path; here on CPU tensors, world size 2 (the GPU test runs the whole flow with HIP kernels)."""
    world, port = 2, _free_port()
    mp.spawn(_host_reduce_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "hr0.npy"), np.load(tmp_path / "hr1.npy"))
