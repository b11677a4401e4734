"""CPU: the host-side decisions that are cached or compiled on demand must stay correct when the inputs change under
them (round-1 advisor findings): the decoupling plan depends on the sparsity pattern of Sigma0, on whether the
evaluation is differentiable, and on in-place edits of the spec tensors; on-demand compiles must be safe when several
processes (the ranks of one node) ask for the same library at once."""
import multiprocessing as mp
import os
import shutil

import pytest
import torch

import lqg_amd
from lqg_amd import decouple, specialize


def _bounded2(**kw):
    return lqg_amd.BoundedActor(dim=2, T=10, device="cpu", dtype=torch.float64, **kw)


def test_decoupling_plan_depends_on_the_pattern_of_sigma0():
    m = _bounded2()
    diag = torch.eye(4, dtype=torch.float64)
    dense = torch.full((4, 4), 0.1, dtype=torch.float64) + torch.eye(4, dtype=torch.float64)
    assert len(decouple.plan(m, 4, diag)) == 2
    assert decouple.plan(m, 4, dense) is None          # a cross-block Sigma0 couples the axes: same instance, new answer
    assert len(decouple.plan(m, 4, diag)) == 2
    fresh = _bounded2()
    assert decouple.plan(fresh, 4, dense) is None and len(decouple.plan(fresh, 4, diag)) == 2


def test_decoupled_parts_follow_the_autograd_mode_of_the_call():
    sig = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    m = _bounded2(sigma_target=sig)
    with torch.no_grad():
        parts = decouple.plan(m, 4)
    assert len(parts) == 2
    again = decouple.plan(m, 4, for_grad=True)         # differentiable call AFTER a cached no_grad one
    assert again[0][0].actor.W.requires_grad           # the gathers are on the current graph
    (again[0][0].actor.W.sum() + again[1][0].actor.W.sum()).backward()
    assert sig.grad is not None and float(sig.grad) != 0.0
    third = decouple.plan(m, 4, for_grad=True)         # and are rebuilt per call: a freed graph is never reused
    assert third[0][0].actor.W is not again[0][0].actor.W


def test_leaf_matrices_are_not_split_by_their_current_zeros():
    """A = eye(4) as an autograd LEAF: its off-block derivatives are not zero, so a differentiable evaluation must keep
    the joint problem; the value-only evaluation may still split it."""
    eye = lambda n, m=None: torch.eye(n, m or n, dtype=torch.float64)
    A = eye(4).requires_grad_(True)
    B = torch.tensor([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 1.0]], dtype=torch.float64)
    Q = torch.diag(torch.tensor([1.0, 1.0, 1.0, 1.0], dtype=torch.float64))
    Q[0, 1] = Q[1, 0] = -1.0
    Q[2, 3] = Q[3, 2] = -1.0
    m = lqg_amd.LQG(A=A, B=B, F=eye(4), V=eye(4), W=eye(4), Q=Q, R=eye(2), T=5)
    with torch.no_grad():
        assert decouple.plan(m, 4) is not None           # by value: two blocks {0,1}, {2,3}
    assert decouple.plan(m, 4, for_grad=True) is None    # differentiable: A is structurally full
    S0 = eye(4).requires_grad_(True)
    m2 = _bounded2()
    assert len(decouple.plan(m2, 4, eye(4), for_grad=True)) == 2
    assert decouple.plan(m2, 4, S0, for_grad=True) is None      # a Sigma0 leaf couples everything it could touch


def test_in_place_edit_of_a_spec_tensor_voids_the_cached_structure():
    eye = lambda n: torch.eye(n, dtype=torch.float64)
    A = eye(4).clone()
    B = torch.tensor([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 1.0]], dtype=torch.float64)
    Q = eye(4).clone()
    Q[0, 1] = Q[1, 0] = Q[2, 3] = Q[3, 2] = -1.0
    m = lqg_amd.LQG(A=A, B=B, F=eye(4), V=eye(4), W=eye(4), Q=Q, R=eye(2), T=5)
    assert decouple.plan(m, 4) is not None
    k0 = specialize.system_pattern(m, 4)[2]
    A[0, 2] = 0.3                                         # couples the blocks; the time-stacked view shares storage and version
    assert specialize.system_pattern(m, 4)[2] != k0
    assert decouple.plan(m, 4) is None


def _compile_worker(pat_dir, q):
    os.environ["LQG_PAT_DIR"] = pat_dir
    import importlib
    import ctypes as C
    from lqg_amd import specialize as sp
    importlib.reload(sp)
    # count compiler invocations: a wrapper that logs one line per call, then runs the real hipcc as its child
    wrapper = os.path.join(pat_dir, "hipcc_counted.sh")
    sp._build.HIPCC = wrapper
    dims, masks, key = sp.class_pattern(lqg_amd.BoundedActor, 2, dim=1)
    so = sp.compile_pattern(key + "_race", dims, masks)
    lib = C.CDLL(so)
    q.put((so, lib.lqg_log_likelihood_sp is not None, os.stat(so).st_ino))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("nproc", [2, 8])
def test_processes_compiling_the_same_cold_pattern_compile_once_and_load_one_file(tmp_path, nproc):
    """The ranks of one node (2, and the 8 the driver scales to) ask for the same COLD pattern at once: the inter-process lock
    lets exactly ONE of them run the compiler; all of them load the same finished file (same path, same inode); no temporaries
    remain."""
    pat_dir = str(tmp_path / "pat")
    os.makedirs(pat_dir)
    log = os.path.join(pat_dir, "compiles.log")
    wrapper = os.path.join(pat_dir, "hipcc_counted.sh")
    with open(wrapper, "w") as f:
        f.write(f'#!/bin/bash\necho "$$" >> {log}\n/opt/rocm/bin/hipcc "$@"\n')
    os.chmod(wrapper, 0o755)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_compile_worker, args=(pat_dir, q)) for _ in range(nproc)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert len({r[0] for r in res}) == 1 and len({r[2] for r in res}) == 1 and all(r[1] for r in res)
    assert len(open(log).read().split()) == 1, open(log).read()      # exactly one compile
    left = [f for f in os.listdir(pat_dir) if ".tmp." in f]
    assert not left, left                                  # no half-written temporaries remain
    shutil.rmtree(pat_dir, ignore_errors=True)
