"""gpurun_out/pmc_<tag>_<leg>_{FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU}/ -> gpurun_out/<tag>_pmc.json: one record per (bench leg,
kernel family) — what bench.py's `roofline.traffic` / VALU fractions read once the file is installed as profiles/r04_pmc.json.

HBM bytes per launch = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B requests at 64 B — /opt/skills/guides/
MI355X_MICROARCH.md "HBM"; calibrated in round 1 on k_riccati's exactly-known byte count) + WRITE_SIZE, both in KiB in the
rocprofv3 tables; averaged over the FULL-SIZE launches of a pass (a leg's parity sample and warm-up shapes launch the same
kernels on tiny batches: launches below half the largest value are excluded).  Records are stamped with the hashes of the
sources the profiled libraries were built from; bench.py refuses a record whose stamps differ from the running build.
    python scripts/pmc_records.py <tag> [leg ...]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# leg -> {kernel family: regex on the kernel name}
KERNELS = {
    "headline_f32": {"forward": r"k_forward_sp<float", "riccati": r"k_riccati_sp<float"},
    "headline_f64": {"forward": r"k_forward_sp<double", "riccati": r"k_riccati_sp<double"},
    "m2_f32": {"forward": r"k_forward_tv_sp<", "riccati": r"k_riccati_tv_sp<"},
    "config3": {"trial": r"k_trial_sp", "forward": r"k_forward_sp<", "riccati": r"k_riccati_sp<"},
    "config5_one_system": {"trial": r"k_trial_sp"},
}


def table(tag, leg, counter):
    rows = []
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{leg}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return rows


def full_size_mean(rows, pat):
    vals = [float(r["Counter_Value"]) for r in rows if re.search(pat, r["Kernel_Name"])]
    if not vals:
        return None, 0
    big = [v for v in vals if v > 0.5 * max(vals)]
    return sum(big) / len(big), len(big)


def main():
    tag = sys.argv[1]
    legs = sys.argv[2:] or list(KERNELS)
    from lqg_amd import build, specialize
    doc = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "tag": tag, "records": []}
    for leg in legs:
        fetch, write, valu = (table(tag, leg, c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"))
        for fam, pat in KERNELS[leg].items():
            f, nf = full_size_mean(fetch, pat)
            w, nw = full_size_mean(write, pat)
            v, nv = full_size_mean(valu, pat)
            if f is None and v is None:
                continue
            names = sorted({r["Kernel_Name"][:110] for r in (fetch or valu) if re.search(pat, r["Kernel_Name"])})
            rec = dict(leg=leg, kernel=fam, kernel_names=names[:3], source_hash=build.source_hash(),
                       sp_headers_hash=specialize._headers_hash(), fetch_size_kib_raw=f, write_size_kib_raw=w,
                       hbm_bytes_per_launch=((2.0 * f + (w or 0.0)) * 1024.0 if f is not None else None),
                       launches_averaged={"FETCH_SIZE": nf, "WRITE_SIZE": nw, "SQ_INSTS_VALU": nv},
                       valu_wave_insts_per_launch=v,
                       profile=f"profiles/r04_pmc.json <- rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --only {leg} "
                               f"--steps 3 --warmup 1 (scripts/pmc_legs.sh {tag}; one counter per pass)")
            if leg.startswith("headline") and v:
                rec["valu_insts_per_step_per_wave"] = v / ((1 << 20) / 64) / 500
            doc["records"].append(rec)
            print(json.dumps({k: rec[k] for k in ("leg", "kernel", "hbm_bytes_per_launch", "valu_wave_insts_per_launch",
                                                  "launches_averaged")}))
    out = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json")
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, len(doc["records"]), "records")


if __name__ == "__main__":
    main()
