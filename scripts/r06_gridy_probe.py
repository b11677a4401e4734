import sys, torch
sys.path.insert(0, '/root/repo')
import lqg_amd
dev = torch.device("cuda")
print(torch.cuda.get_device_properties(0).multi_processor_count)
for B, n, dt in ((70000, 3, torch.float32), (70000, 3, torch.float64), (140000, 5, torch.float32), (70000, 1000, torch.float32)):
    sig = torch.linspace(3.0, 40.0, B, device=dev, dtype=dt)
    m = lqg_amd.BoundedActor(T=50, sigma_target=sig, device=dev, dtype=dt)
    x = lqg_amd.BoundedActor(T=50, device=dev, dtype=dt).simulate(5, n=n).contiguous()
    try:
        ll = m.log_likelihood(x)
        torch.cuda.synchronize()
        ref = lqg_amd.BoundedActor(T=50, sigma_target=sig[-3:], device=dev, dtype=dt).log_likelihood(x)
        print(B, n, dt, "ok", bool(torch.isfinite(ll).all()), float((ll[-3:] - ref).abs().max()))
    except Exception as e:
        print(B, n, dt, "FAILED", repr(e)[:300])
    # gradient
    try:
        s2 = sig.clone().requires_grad_(True)
        m2 = lqg_amd.BoundedActor(T=50, sigma_target=s2, device=dev, dtype=dt)
        m2.log_likelihood(x).sum().backward()
        torch.cuda.synchronize()
        print("   grad ok", bool(torch.isfinite(s2.grad).all()))
    except Exception as e:
        print("   grad FAILED", repr(e)[:300])
