"""Value + gradient through torch.autograd on the two shapes the review names (GPU box):
  headline   SubjectiveActor(dim=2), T=500, B candidates x 1 trial each (own trajectory)
  config3    BoundedActor, T=1067, C candidates x N shared trials
Prints one JSON line per (shape, dtype): ms per value+grad, solves+grad/s, forward-only ms beside it."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lqg_amd
from lqg_amd import workload


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def leaf_params(names, n, seed, dev, dt):
    p = workload.sample_params(names, n, seed, dev, dt)
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def headline(B, T, dt, dev, steps, warmup):
    names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost", "subj_noise", "subj_vel_noise")
    with torch.no_grad():
        sys0, _ = workload.headline_system(B, T, 1234, dev, dt)
        x = workload.pack_trials(workload.simulate_one_trial_each(sys0, seed=3))
    p = leaf_params(names, B, 1234, dev, dt)

    def vg():
        for v in p.values():
            v.grad = None
        m = lqg_amd.SubjectiveActor(dim=2, T=T, process_noise=1.0, dt=1.0 / 60, device=dev, dtype=dt, **p)
        m.log_likelihood(x).sum().backward()

    def fwd():
        with torch.no_grad():
            m = lqg_amd.SubjectiveActor(dim=2, T=T, process_noise=1.0, dt=1.0 / 60, device=dev, dtype=dt, **p)
            m.log_likelihood(x).sum()

    return timed(vg, steps, warmup), timed(fwd, steps, warmup)


def config3(C, N, T, dt, dev, steps, warmup):
    names = ("action_variability", "sigma_target", "sigma_cursor", "action_cost")
    with torch.no_grad():
        truth = lqg_amd.BoundedActor(T=T, sigma_target=20.0, sigma_cursor=3.0, action_cost=0.3, device=dev, dtype=dt)
        x = workload.pack_trials(truth.simulate(17, n=N))
    p = leaf_params(names, C, 1234, dev, dt)

    def vg():
        for v in p.values():
            v.grad = None
        m = lqg_amd.BoundedActor(T=T, process_noise=1.0, dt=1.0 / 60, device=dev, dtype=dt, **p)
        m.log_likelihood(x).sum().backward()

    def fwd():
        with torch.no_grad():
            m = lqg_amd.BoundedActor(T=T, process_noise=1.0, dt=1.0 / 60, device=dev, dtype=dt, **p)
            m.log_likelihood(x).sum()

    return timed(vg, steps, warmup), timed(fwd, steps, warmup)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-batch", type=int, default=16)
    ap.add_argument("--cands", type=int, default=4096)
    ap.add_argument("--trials", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtypes", default="f32,f64")
    ap.add_argument("--shapes", default="headline,config3")
    ap.add_argument("--sp", default="1", help="LQG_ADJOINT_SP: 1 = split sweep on the pattern libraries (round 5), 0 = round-1 lane kernels")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from lqg_amd import options
    options.set("ADJOINT_SP", int(a.sp))
    for dn in a.dtypes.split(","):
        dt = torch.float32 if dn == "f32" else torch.float64
        if "headline" in a.shapes:
            B = 1 << a.log2_batch
            try:
                vg, fw = headline(B, 500, dt, dev, a.steps, a.warmup)
                print(json.dumps({"shape": "headline", "dtype": dn, "B": B, "T": 500, "value_and_grad_ms": vg, "forward_ms": fw,
                                  "solves_with_grad_per_s": B / vg * 1e3, "adjoint_sp": int(a.sp)}), flush=True)
            except Exception as e:
                print(json.dumps({"shape": "headline", "dtype": dn, "error": repr(e)[:300]}), flush=True)
        if "config3" in a.shapes:
            try:
                vg, fw = config3(a.cands, a.trials, 1067, dt, dev, a.steps, a.warmup)
                print(json.dumps({"shape": "config3", "dtype": dn, "C": a.cands, "N": a.trials, "T": 1067, "value_and_grad_ms": vg,
                                  "forward_ms": fw, "adjoint_sp": int(a.sp)}), flush=True)
            except Exception as e:
                print(json.dumps({"shape": "config3", "dtype": dn, "error": repr(e)[:300]}), flush=True)


if __name__ == "__main__":
    main()
