"""Kernel timeline of ONE replay of the graphed one-vector value + gradient (BoundedActor, T = 500, 50 trials, fp64):
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_fdg -o p -- python3 scripts/fd_graph_timeline.py run
python3 scripts/fd_graph_timeline.py report"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch, lqg_amd
    from lqg_amd.infer import gradient
    true = dict(sigma_target=25.0, sigma_cursor=1.0, action_cost=0.05, action_variability=0.5)
    with torch.no_grad():
        x = lqg_amd.BoundedActor(T=500, device="cuda", dtype=torch.float64, **true).simulate(0, n=50)
    p0 = dict(sigma_target=20.0, sigma_cursor=2.0, action_cost=0.1, action_variability=0.4)
    for i in range(30):
        gradient.value_and_grad(x, lqg_amd.BoundedActor, dict(p0, sigma_target=20.0 + 1e-3 * i), method="fd")
    torch.cuda.synchronize()
else:
    import csv, glob
    f = glob.glob("gpurun_out/prof_fdg/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "build_rk" in r["Kernel_Name"]]
    a, b = starts[-2], starts[-1]                      # the last complete replay (from one build_rk to the next)
    # the replay begins with the constructor kernels that precede build_rk: walk back while the gaps stay small
    while a > 0 and int(rows[a]["Start_Timestamp"]) - int(rows[a - 1]["End_Timestamp"]) < 20000:
        a -= 1
    while b > 0 and int(rows[b]["Start_Timestamp"]) - int(rows[b - 1]["End_Timestamp"]) < 20000:
        b -= 1
    seq = rows[a:b]
    t0 = int(seq[0]["Start_Timestamp"])
    busy = 0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print("%8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    print("# span us %.1f busy us %.1f kernels %d" % ((int(seq[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, len(seq)))
