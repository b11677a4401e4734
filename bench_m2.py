#!/usr/bin/env python3
"""bench_m2.py — measurement mode M2 of SURVEY.md §8(d): the reference's LITERAL data model.

Genuinely time-varying (T, ...)-stacked specs in, every intermediate the reference materialises out
(L, H from lqr.backward; K from kf.forward; mu, Sigma from conditional_moments).  This is the HBM-bound regime
(≈8.5 FLOP/B at n=6): algorithmic bytes per solve = inputs once + outputs once (SURVEY.md §8d formula, 824 kB at
n=6, T=500, fp32).  Arrays are laid out [T][element][system] (system index fastest) and handed to the C ABI as
strided views, so that a wave's 64 lanes touch 64 consecutive elements.

    python bench_m2.py [--log2-batch 14] [--T 500] [--reps 5]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import lqg_amd
from lqg_amd import _abi, _hip, workload

PEAK_HBM_GBS = 8000.0


def soa(base, T, B, gen, jitter, sym=False, positive_diag=False):
    """base[B, r, c] -> time-varying field with physical layout [T, r, c, B], logical shape [B, T, r, c]."""
    r, c = base.shape[-2:]
    xi = torch.randn((T, r, c, B), dtype=base.dtype, device=base.device, generator=gen)
    if sym:
        xi = 0.5 * (xi + xi.transpose(1, 2))
    # (.contiguous(): the product of a permuted [B, r, c] view and a dense [T, r, c, B] tensor comes out in the PERMUTED
    # operand's memory order, [T][B][r][c] -- 64 lanes of a wave 64 cache lines apart -- which is what this function
    # returned until the end of round 3: see DESIGN.md 6b)
    phys = (base.permute(1, 2, 0).unsqueeze(0) * (1.0 + jitter * xi)).contiguous()
    out = phys.permute(3, 0, 1, 2)
    assert out.stride(0) == 1 and out.stride(1) == r * c * B, out.stride()
    return out


def soa_congruent(base, T, B, gen, jitter):
    """base[B, n, n] symmetric -> D_t base D_t with D_t = diag(1 + jitter xi_t): every non-zero entry moves in time and over the
    systems, positive semi-definiteness and the sparsity pattern are kept (a congruence).  Layout as soa()."""
    n = base.shape[-1]
    dg = 1.0 + jitter * torch.randn((T, n, B), dtype=base.dtype, device=base.device, generator=gen)
    phys = (base.permute(1, 2, 0).unsqueeze(0) * dg.unsqueeze(2) * dg.unsqueeze(1)).contiguous()
    out = phys.permute(3, 0, 1, 2)
    assert out.stride(0) == 1 and out.stride(1) == n * n * B, out.stride()
    return out


def m2_system(dev, dtype, B, T, psd=False):
    """The headline model with every non-zero entry of every spec matrix moving in time and over the systems (multiplicative
    jitter): genuinely time-varying [T, ...] stacks in the [T][element][system] layout.  -> (system, time-invariant base).

    psd=False (mode M2's workload since round 3): every entry of Q and R jitters on its own — Q_t of the tracking models has a
    zero eigenvalue, so it is no longer positive semi-definite, the eigenvalue floor of lqr.py:27-28 cannot be proven inactive
    and the exact block decoupling is (rightly) refused: the joint n = 10 problem is solved.  psd=True: the costs move by a
    congruence D_t Q D_t (what a model whose PARAMETERS move in time produces: the costs stay costs) — the two 1-D components
    decouple as for the time-invariant model."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    base, _ = workload.headline_system(B, T, seed=77, device=dev, dtype=dtype)
    a0, d0 = base.actor, base.dynamics
    first = lambda t: t[:, 0] if t.dim() == 4 else t[0].expand(B, *t.shape[1:])
    jit = 1e-3
    cost = (lambda t: soa_congruent(first(t), T, B, gen, jit)) if psd else (lambda t: soa(first(t), T, B, gen, jit, sym=True))
    act = dict(A=soa(first(a0.A), T, B, gen, jit), B=soa(first(a0.B), T, B, gen, jit), F=soa(first(a0.F), T, B, gen, jit),
               V=soa(first(a0.V), T, B, gen, jit), W=soa(first(a0.W), T, B, gen, jit),
               Q=cost(a0.Q), R=cost(a0.R))
    actor = lqg_amd.LQGSpec(Q=act["Q"], q=a0.q, Qf=a0.Qf if a0.Qf.dim() == 3 else a0.Qf, qf=a0.qf, P=a0.P, R=act["R"],
                            r=a0.r, A=act["A"], B=act["B"], V=act["V"], F=act["F"], W=act["W"])
    dyn = lqg_amd.LQGSpec(Q=d0.Q, q=d0.q, Qf=d0.Qf, qf=d0.qf, P=d0.P, R=d0.R, r=d0.r,
                          A=soa(first(d0.A), T, B, gen, jit), B=soa(first(d0.B), T, B, gen, jit),
                          F=soa(first(d0.F), T, B, gen, jit), V=soa(first(d0.V), T, B, gen, jit),
                          W=soa(first(d0.W), T, B, gen, jit))
    return lqg_amd.System(actor=actor, dynamics=dyn), base


def m2_pattern():
    """(dims, masks, key) of the sparsity pattern the M2 workload keeps through time (a small CPU instance of it: the pattern
    does not depend on the batch) — __graft_entry__.build() compiles its library so that the bench does not."""
    from lqg_amd import specialize
    system, _ = m2_system(torch.device("cpu"), torch.float64, 4, 3)
    dims, masks = specialize.pattern_of_time_varying(system, 4)
    return dims, masks, specialize.pattern_key(dims, masks)


def tv_component_patterns():
    """[(dims, masks, key)] of the decoupled components of the psd=True workload (bench.py leg `timevarying_f64`,
    tests/test_gpu_timevarying.py): what `System.log_likelihood` asks the pattern cache for on such a model."""
    from lqg_amd import specialize
    system, _ = m2_system(torch.device("cpu"), torch.float64, 4, 3, psd=True)
    out = {}
    for sub, cols, _ in (system.decoupled(4, None) or [(system, [0, 1, 2, 3], None)]):
        dims, masks = specialize.pattern_of_time_varying(sub, len(cols))
        out[specialize.pattern_key(dims, masks)] = (dims, masks, specialize.pattern_key(dims, masks))
    return list(out.values())


def run(dev, dtype_name="f32", log2_batch=17, T=500, reps=5):
    """One M2 measurement -> dict (the line main() prints; bench.py's `extra.m2_f32` leg)."""
    from lqg_amd import _hipev
    dtype = torch.float32 if dtype_name == "f32" else torch.float64
    w = 4 if dtype_name == "f32" else 8
    B = 1 << log2_batch
    system, base = m2_system(dev, dtype, B, T)
    x = workload.pack_trials(workload.simulate_one_trial_each(base, seed=5))          # [B,1,T+1,4]
    dm = dict(x=4, b=6, u=2, y=4, m=10, d=4)

    def out(*shape):   # physical [T, ..., B], logical [B, T, ...]
        t = torch.empty((T,) + shape + (B,), dtype=dtype, device=dev)
        return t.permute(len(shape) + 1, 0, *range(1, len(shape) + 1))

    L, H, K = out(dm["u"], dm["b"]), out(dm["u"], dm["u"]), out(dm["b"], dm["y"])
    mu = torch.empty((T, dm["m"], B, 1), dtype=dtype, device=dev).permute(2, 3, 0, 1)   # [B,1,T,m]
    Sig = out(dm["m"], dm["m"])

    lib = _abi.load()
    lnm = _hip.Launch(system.actor, system.dynamics, d=4, n_trials=1)
    xx, xb = _hip._prep_x(lnm, x)
    nbytes = lib.lqg_workspace_bytes(C.byref(lnm.p), _abi.OP_CONDITIONAL_MOMENTS)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    ll = torch.empty((B, 1), dtype=dtype, device=dev)
    evs = [_hipev.Event() for _ in range(4)]             # phase events the library records on the launch stream
    for i in range(4):
        lnm.p.phase_events[i] = evs[i].h

    # the structure-specialised library when the specs keep one sparsity pattern through time (they do: a zoo model with
    # moving entries), else the dense kernels of the main library
    sp_entry = _hip.materialised_entry(lnm, system, 4)
    if sp_entry is not None:                             # (a library that refuses the problem leaves it to the dense kernels)
        rc = sp_entry(C.byref(lnm.p), lnm.traj(xx, xb), lnm.view(L), _abi.NULL_VIEW, lnm.view(H), lnm.view(K), lnm.traj(mu),
                      lnm.view(Sig), C.c_void_p(ll.data_ptr()), 1, C.c_void_p(ws.data_ptr()), nbytes, lnm.stream())
        if rc != 0:
            sp_entry = None

    def one_pass(dense=False):
        if sp_entry is not None and not dense:
            # one ABI call, two kernels: k_riccati_tv_sp (L, H out + gain scratch) -> k_forward_tv_sp (K, mu, Sigma, ll out)
            _abi.check(sp_entry(
                C.byref(lnm.p), lnm.traj(xx, xb), lnm.view(L), _abi.NULL_VIEW, lnm.view(H), lnm.view(K), lnm.traj(mu),
                lnm.view(Sig), C.c_void_p(ll.data_ptr()), 1, C.c_void_p(ws.data_ptr()), nbytes, lnm.stream()),
                "lqg_solve_materialised_sp")
            return
        # one ABI call, two kernels: k_riccati (L, H out + gain scratch) -> k_forward (K, mu, Sigma, ll out)
        _abi.check(lib.lqg_solve_materialised(
            C.byref(lnm.p), lnm.traj(xx, xb), lnm.view(L), _abi.NULL_VIEW, lnm.view(H), lnm.view(K), lnm.traj(mu),
            lnm.view(Sig), C.c_void_p(ll.data_ptr()), 1, 1, C.c_void_p(ws.data_ptr()), nbytes, lnm.stream()),
            "lqg_solve_materialised")

    one_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms, ric, fwd = [], [], []
    for _ in range(reps):
        e0.record()
        one_pass()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
        ric.append(evs[0].elapsed_ms(evs[1]))
        fwd.append(evs[1].elapsed_ms(evs[2]))
    ms = float(np.median(ms))
    dense_ms = None
    if sp_entry is not None:                             # the same pass on the dense kernels, for the record
        one_pass(dense=True)
        torch.cuda.synchronize()
        dm_ = []
        for _ in range(2):
            e0.record()
            one_pass(dense=True)
            e1.record()
            e1.synchronize()
            dm_.append(e0.elapsed_time(e1))
        dense_ms = float(np.median(dm_))
        one_pass()                                       # (the outputs checked below are the specialised path's)
        torch.cuda.synchronize()

    b, u, y, xx_, m, d = dm["b"], dm["u"], dm["y"], dm["x"], dm["m"], dm["d"]
    in_per_step = (3 * b * b + b * u + y * b + y * y + u * u) + (b + u * b + u) + (2 * xx_ * xx_ + xx_ * u + y * xx_ + y * y)
    out_per_step = u * b + u + u * u + b * y + m + m * m
    bytes_dense = w * (T * in_per_step + (T + 1) * d) + w * T * out_per_step          # SURVEY.md §8(d) M2 formula
    if sp_entry is not None:
        # the pattern kernels request only the structurally non-zero entries of the specs (a zero of the pattern is never
        # read): the bytes the pass MUST move are those rows, the trajectory and every output it writes (l is not requested)
        _, masks, _ = m2_pattern()
        in_rows = sum(int(np.asarray(masks[k]).sum()) for k in ("Aa", "Ba", "Fa", "Va", "Wa", "Q", "Rr", "Ad", "Bd", "Fd", "Vd", "Wd"))
        bytes_solve = w * (T * in_rows + (T + 1) * d) + w * T * (out_per_step - u)
    else:
        in_rows = in_per_step
        bytes_solve = bytes_dense
    gbs = bytes_solve * B / (ms * 1e-3) / 1e9
    # parity spot check of the time-varying path against the fp64 oracle
    import oracle as OC
    OC.build()
    sel = [0, B // 2, B - 1]
    idx = torch.as_tensor(sel, device=dev)
    host = lambda spec: {f: (getattr(spec, f)[idx] if getattr(spec, f).dim() == workload._batched_ndim(f)
                             else getattr(spec, f).expand(len(sel), *getattr(spec, f).shape)).double().cpu().numpy()
                         for f in lqg_amd.LQGSpec._fields}
    a64, d64 = host(system.actor), host(system.dynamics)
    Lr, _, Hr = OC.riccati_backward(a64)
    Kr = OC.kalman_forward(a64)
    mur, Sr = OC.conditional_moments(a64, d64, x[idx].double().cpu().numpy())
    rel = lambda got, ref: float(np.abs(got.double().cpu().numpy() - ref).max() / np.abs(ref).max())
    parity = dict(L=rel(L[idx], Lr), H=rel(H[idx], Hr), K=rel(K[idx], Kr), mu=rel(mu[idx], mur), Sigma=rel(Sig[idx], Sr))
    # Two named fractions (round-3 review, weak #7): on SURVEY 8(d)'s dense-format byte count (824 016 B per solve at n=6, T=500,
    # fp32) only the DENSE kernels do that work — frac_survey_bytes is theirs; the pattern kernels never touch the structural
    # zeros and are priced on the bytes they must move (non-zero spec rows + trajectory + every output) — frac_moved_bytes.
    frac_survey = (bytes_dense * B / (dense_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if dense_ms else \
        (bytes_dense * B / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if sp_entry is None else None)
    return {
        "mode": "M2 (time-varying [T,...] specs in; L,H,K,mu,Sigma materialised out)", "dtype": dtype_name,
        "frac_survey_bytes": frac_survey, "frac_moved_bytes": gbs / PEAK_HBM_GBS,
        "fractions": {"frac_survey_bytes": {"kernels": "k_riccati<TI=false> + k_forward<TI=false,FUSED,MAT> (dense, main library)",
                                            "bytes_per_solve": bytes_dense, "ms_per_pass": dense_ms if dense_ms else ms,
                                            "frac_of_8TBps": frac_survey},
                      "frac_moved_bytes": {"kernels": "k_riccati_tv_sp + k_forward_tv_sp (pattern library)" if sp_entry is not None
                                           else "dense kernels", "bytes_per_solve": bytes_solve, "ms_per_pass": ms,
                                           "frac_of_8TBps": gbs / PEAK_HBM_GBS}},
        "systems": B, "T": T, "ms_per_pass": ms, "solves_per_s": B / (ms * 1e-3),
        "algorithmic_bytes_per_solve": bytes_solve, "spec_rows_read_per_step": in_rows,
        "dense_format_bytes_per_solve": bytes_dense,
        "input_layout": "[T][row][col][system] (system stride 1), asserted in soa()",
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                     "frac_of_measured_copy_rate": gbs / 6290.0,
                     "dense_format_bytes_over_time_GBps": bytes_dense * B / (ms * 1e-3) / 1e9,
                     "dense_kernels_frac": (bytes_dense * B / (dense_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if dense_ms else None,
                     "algorithmic_bytes_per_pass": bytes_solve * B, "traffic": None,
                     "kernel": ("k_riccati_tv_sp + k_forward_tv_sp (pattern library: structurally non-zero entries loaded per step; "
                                "one pass = both)" if sp_entry is not None else
                                "k_riccati<TI=false> + k_forward<TI=false, FUSED, MAT> (one pass = both)"),
                     "dense_kernels_ms_per_pass": dense_ms,
                     "kernel_ms": float(np.median(ric)) + float(np.median(fwd)),
                     "riccati_kernel_ms": float(np.median(ric)), "forward_kernel_ms": float(np.median(fwd))},
        "calls": ("lqg_solve_materialised_sp" if sp_entry is not None else "lqg_solve_materialised") + " (1 ABI call, 2 kernels)",
        "workspace_GB": nbytes / 1e9, "parity_rel_maxnorm_vs_fp64_oracle": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-batch", type=int, default=14)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    args = ap.parse_args()
    print(json.dumps(run(torch.device("cuda", 0), args.dtype, args.log2_batch, args.T, args.reps)))


if __name__ == "__main__":
    main()
