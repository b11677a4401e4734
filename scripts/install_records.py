"""Copy the records `bash scripts/final_records.sh <tag>` left under gpurun_out/ into profiles/ under their committed names
(profiles/ does not travel back from the GPU box by itself):  python scripts/install_records.py r03_w [old_tag_to_remove]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
old = sys.argv[2] if len(sys.argv) > 2 else None
g, p = "gpurun_out", "profiles"
pairs = {f"{tag}_bench.json": f"{tag}_bench_f32.json", f"{tag}_configs.jsonl": f"{tag}_bench_configs.jsonl",
         f"{tag}_grad_f64.jsonl": f"{tag}_bench_grad_f64.jsonl", f"prof_{tag}_kt/p_kernel_stats.csv": f"{tag}_kernel_stats.csv",
         f"prof_{tag}64_kt/p_kernel_stats.csv": f"{tag}64_kernel_stats.csv", f"prof_{tag}m2_kt/p_kernel_stats.csv": f"{tag}m2_kernel_stats.csv",
         f"m2_{tag}m2.json": f"{tag}m2_m2_f32.json", f"prof_{tag}_kt.json": f"{tag}_bench_under_rocprof.json",
         f"{tag}_pmc_traffic.json": "r03_pmc_traffic.json"}
for f in ("small_batch_f32.txt", "small_batch_f64.txt", "delay12.txt", "fp32_tail.txt", "timeline_config2.txt", "delay12_timeline.txt",
          "delay12_kernel_stats.csv"):
    pairs[f"{tag}_{f}"] = f"{tag}_{f}"
for src, dst in pairs.items():
    shutil.copy(os.path.join(g, src), os.path.join(p, dst))
# M2 traffic from its PMC passes (full-size launches only: the parity sample's small launches are excluded)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{g}/prof_{tag}m2_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = ("k_riccati_tv_sp" if "k_riccati_tv_sp" in k else "k_forward_tv_sp" if "k_forward_tv_sp" in k else
                    "k_riccati (dense, for comparison)" if "k_riccati<" in k else
                    "k_forward (dense, for comparison)" if "k_forward<" in k else None)
            if name and r["Counter_Name"] == c:
                acc[name][c].append(float(r["Counter_Value"]))
m2 = json.loads(open(f"{g}/m2_{tag}m2.json").read().strip().splitlines()[-1])
out = {"mode": "M2, fp32, 2^17 solves, T=500 (bench_m2.py)",
       "note": "HBM bytes per launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes; "
               "averaged over the full-size launches of the PMC passes", "launches_averaged": {}, "kernels": {}}
for name, d in acc.items():
    big = lambda v: [x for x in v if x > 0.9 * max(v)]
    F, W = big(d["FETCH_SIZE"]), big(d["WRITE_SIZE"])
    fs, ws = sum(F) / len(F), sum(W) / len(W)
    out["launches_averaged"][name] = len(F)
    out["kernels"][name] = {"fetch_size_kib_raw": fs, "write_size_kib_raw": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024}
rf = m2["roofline"]
timed = [n for n in out["kernels"] if "comparison" not in n] if any("tv_sp" in n for n in out["kernels"]) else list(out["kernels"])
out["hbm_bytes_per_pass"] = sum(out["kernels"][n]["hbm_bytes_per_launch"] for n in timed)
out["bench_m2"] = {"ms_per_pass": m2["ms_per_pass"], "frac": rf["frac"], "achieved_GBs": rf["achieved"]}
out["algorithmic_bytes_per_pass"] = rf["achieved"] * 1e9 * m2["ms_per_pass"] * 1e-3
out["traffic_over_algorithmic"] = out["hbm_bytes_per_pass"] / out["algorithmic_bytes_per_pass"]
json.dump(out, open(f"{p}/{tag}m2_pmc.json", "w"), indent=1)
if old and old != tag:
    for f in glob.glob(f"{p}/{old}_*") + glob.glob(f"{p}/{old}64_*") + glob.glob(f"{p}/{old}m2_*"):
        os.remove(f)
print("installed", len(pairs) + 1, "records for", tag, "| M2 traffic / algorithmic %.3f" % out["traffic_over_algorithmic"])
