"""fp32 point mass at long horizons: error of every fp32 route against the fp64 path on the fp64 image of the same fp32 inputs
(the quantity tests/test_gpu_parity.py::test_fp32_candidate_ranges_point_mass bounds), and the cost of the hi + lo pass of the MIXED
mode (lqg_kernels_sp.hpp: LQG_HILO_MIN) on a many-trials shape.  Run on the GPU box: python scripts/pointmass_hilo.py"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import lqg_amd                                   # noqa: E402
from lqg_amd import options, workload            # noqa: E402

dev = torch.device("cuda")
names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost")
out = {}
for T in (500, 1067, 2000, 4000):
    for seed in (5, 11):
        B, n, d = 256, 8, 2
        gen = torch.Generator(device=dev); gen.manual_seed(seed)
        kw = {k: workload.log_uniform(B, *workload.RANGES[k], gen, dev, torch.float32) for k in names}
        m32 = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32, **kw)
        x = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32).simulate(3, n=n)[..., :d].contiguous()
        ref = m32.to(torch.float64).log_likelihood(x.double())
        scale = ref.abs().clamp_min(float(T * d))
        row = {}
        for name, ov in (("default", {}), ("mixed", dict(F32_WIDE=0)), ("fp32", dict(F32_WIDE=0, MIXED=0))):
            with options.override(**ov):
                ll = m32.log_likelihood(x).double()
            row[name] = float(((ll - ref).abs() / scale).max())
        out[f"T{T}_seed{seed}"] = row
        print(T, seed, {k: "%.2e" % v for k, v in row.items()}, flush=True)

# the one-pass sweeps (many trials per candidate): 64-lane workgroups and the 256 x 2 geometry
for B, n in ((300, 500), (256, 800)):
    T, d = 1067, 2
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    kw = {k: workload.log_uniform(B, *workload.RANGES[k], gen, dev, torch.float32) for k in names}
    m32 = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32, **kw)
    x = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32).simulate(4, n=n)[..., :d].contiguous()
    ref = m32.to(torch.float64).log_likelihood(x.double())
    scale = ref.abs().clamp_min(float(T * d))
    row = {}
    for name, ov in (("default", {}), ("mixed", dict(F32_WIDE=0)), ("fp32", dict(F32_WIDE=0, MIXED=0))):
        with options.override(**ov):
            ll = m32.log_likelihood(x).double()
        row[name] = float(((ll - ref).abs() / scale).max())
    out[f"onepass_{B}x{n}"] = row
    print("one-pass", B, n, {k: "%.2e" % v for k, v in row.items()}, flush=True)

# cost: 1024 candidates x 1024 trials, T = 1067, MIXED route
for model, d in (("PointMassBoundedActor", 2), ("BoundedActor", 2)):
    B, n, T = 1024, 1024, 1067
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    cls = getattr(lqg_amd, model)
    ks = names
    kw = {k: workload.log_uniform(B, *workload.RANGES[k], gen, dev, torch.float32) for k in ks}
    m = cls(T=T, device=dev, dtype=torch.float32, **kw)
    x = workload.pack_trials(cls(T=T, device=dev, dtype=torch.float32).simulate(3, n=n)[..., :d].contiguous())
    with options.override(F32_WIDE=0):
        for _ in range(2):
            m.log_likelihood(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.log_likelihood(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
    out[f"ms_{model}_{B}x{n}_T{T}_mixed"] = ms
    print(model, "%.2f ms" % ms, flush=True)
print(json.dumps(out))
