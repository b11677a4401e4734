"""gpurun_out/prof_<tag>_{FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU}/ -> one record per hot-path kernel in
profiles/r03_pmc_traffic.json (what bench.py's `roofline.traffic` reads) + the trimmed kernel-stats table.

HBM bytes per launch = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B requests at 64 B —
/opt/skills/guides/MI355X_MICROARCH.md "HBM"; calibrated in round 1 on k_riccati's exactly-known byte count) + WRITE_SIZE,
both in KiB in the rocprofv3 tables.  Records are stamped with the hashes of the sources the profiled libraries were built
from; bench.py refuses a record whose stamps differ from the running build.
    python scripts/make_pmc_record.py <tag> <dtype> <log2_batch> [T]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(tag, counter):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{counter}", "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, counter, pat):
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and re.search(pat, r["Kernel_Name"])]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    tag, dtype, log2b = sys.argv[1], sys.argv[2], int(sys.argv[3])
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    from lqg_amd import build, specialize
    bench = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_kt.json")))
    path = bench["config"]["path"]
    merged = re.search(r"(\d+) identical decoupled components", path)
    sp = "structure-specialised" in path
    import lqg_amd
    key = specialize.class_pattern(lqg_amd.SubjectiveActor, 2, dim=1)[2] if sp else "generic"
    out_path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {"records": []}
    fetch, write, valu = table(tag, "FETCH_SIZE"), table(tag, "WRITE_SIZE"), table(tag, "SQ_INSTS_VALU")
    for kern, pat in (("forward", r"k_forward_sp<" if sp else r"k_forward<"), ("riccati", r"k_riccati_sp<" if sp else r"k_riccati<")):
        f, nf = per_kernel(fetch, "FETCH_SIZE", pat)
        w, nw = per_kernel(write, "WRITE_SIZE", pat)
        v, nv = per_kernel(valu, "SQ_INSTS_VALU", pat)
        if f is None:
            continue
        name = (("k_forward_sp" if sp else "k_forward") if kern == "forward" else ("k_riccati_sp" if sp else "k_riccati")) + \
            (f"<merged{merged.group(1)}>" if merged else "") + "x1"
        waves = (1 << log2b) / 64
        rec = dict(kernel=name, dtype=dtype, log2_batch=log2b, pattern_key=key, source_hash=build.source_hash(),
                   sp_headers_hash=specialize._headers_hash(), fetch_size_kib_raw=f, write_size_kib_raw=w,
                   hbm_bytes_per_launch=(2.0 * f + (w or 0.0)) * 1024.0, launches_averaged=nf,
                   valu_wave_insts_per_launch=v, valu_insts_per_step_per_wave=(v / waves / T if v else None),
                   profile=f"profiles/r03_{tag}_*  (rocprofv3 --pmc passes of `python bench.py --dtype {dtype} --log2-batch {log2b}`)",
                   bench_value_under_rocprof=bench["value"])
        doc["records"] = [r for r in doc["records"] if not (r["kernel"] == name and r["dtype"] == dtype and r["log2_batch"] == log2b)]
        doc["records"].append(rec)
        print(json.dumps(rec))
    json.dump(doc, open(out_path, "w"), indent=1)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_kt", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        os.system(f"python {ROOT}/profiles/summarize.py {stats[0]} {ROOT}/profiles/r03_{tag}_kernel_stats.csv 10")
        os.system(f"cp {ROOT}/gpurun_out/prof_{tag}_kt.json {ROOT}/profiles/r03_{tag}_bench_under_rocprof.json")


if __name__ == "__main__":
    main()
