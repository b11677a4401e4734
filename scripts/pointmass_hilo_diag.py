"""Is the MIXED route's error on the fp32 point mass the operator rounding?  On the GPU box: the per-candidate error of the MIXED
route at T = 1067 (inputs of tests/test_gpu_parity.py::test_fp32_candidate_ranges_point_mass), then the NumPy emulation
(scripts/pointmass_hilo_emulation.py) of the worst candidates on the SAME data.  LQG_PAT_DIR selects an A/B pattern directory."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import lqg_amd                                   # noqa: E402
from lqg_amd import options, workload            # noqa: E402
import pointmass_hilo_emulation as emu           # noqa: E402

dev = torch.device("cuda")
names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost")
T, B, n, d = 1067, 256, 8, 2
gen = torch.Generator(device=dev); gen.manual_seed(5)
kw = {k: workload.log_uniform(B, *workload.RANGES[k], gen, dev, torch.float32) for k in names}
m32 = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32, **kw)
x = lqg_amd.PointMassBoundedActor(T=T, device=dev, dtype=torch.float32).simulate(3, n=n)[..., :d].contiguous()
ref = m32.to(torch.float64).log_likelihood(x.double())
scale = ref.abs().clamp_min(float(T * d))
with options.override(F32_WIDE=0):
    ll = m32.log_likelihood(x).double()
err = ((ll - ref).abs() / scale)
per = err.max(dim=1).values.cpu().numpy()
order = np.argsort(-per)
print("MIXED route: max %.2e, candidates above 1e-6: %d of %d, above 3e-7: %d" % (per.max(), (per > 1e-6).sum(), B, (per > 3e-7).sum()))
print("|x| max", float(x.abs().max()))
if len(sys.argv) > 1 and sys.argv[1] == "emulate":
    xs = x.double().cpu().numpy()
    for c in order[:3]:
        one = lqg_amd.PointMassBoundedActor(T=T, device="cpu", dtype=torch.float32, **{k: float(v[c]) for k, v in kw.items()})
        ops = emu.build_ops(one, d)
        fmax = max(np.abs(o[0]).max() for o in ops[:-1])
        r64 = emu.sweep(ops, xs, "f64")
        sc = np.maximum(np.abs(r64), T * d)
        row = {k: float((np.abs(emu.sweep(ops, xs, k) - r64) / sc).max()) for k in ("f32", "hilo", "ops64")}
        print("candidate", int(c), {k: float(v[c]) for k, v in kw.items()}, "GPU err %.2e" % per[c],
              "emulation f64 vs GPU fp64 %.1e" % float((np.abs(r64 - ref[c].cpu().numpy()) / sc).max()),
              {k: "%.2e" % v for k, v in row.items()}, "max|F - I| %.1f" % fmax, flush=True)
