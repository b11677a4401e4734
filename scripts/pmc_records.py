"""gpurun_out/pmc_<tag>_<leg>_{FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU,GRBM_GUI_ACTIVE}/ -> gpurun_out/<tag>_pmc.json: one record per
(bench leg, kernel family) — what bench.py's `roofline.traffic` / VALU fractions / sustained clock read once the file is installed
as profiles/r06_pmc.json.

HBM bytes per launch = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B requests at 64 B — /opt/skills/guides/
MI355X_MICROARCH.md "HBM"; calibrated in round 1 on k_riccati's exactly-known byte count) + WRITE_SIZE, both in KiB in the
rocprofv3 tables.  A family may hold several kernels (the time-chunked per-trial sweep is four): per distinct kernel name the
FULL-SIZE launches of a pass are averaged (a leg's parity sample and warm-up shapes launch the same kernels on tiny batches:
launches below half the largest value are excluded), and the family is the SUM over its kernels.  Sustained clock =
GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same dispatches (MI355X_MICROARCH.md "DVFS give-back").  Records are stamped with the
hashes of the sources the profiled libraries were built from; bench.py refuses a record whose stamps differ from the running build.
    python scripts/pmc_records.py <tag> [leg ...]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

F32 = r"(<float|If)"
# leg -> {kernel family: regex on the (demangled or mangled) kernel name}
KERNELS = {
    "headline_f32": {"forward": r"k_forward_sp<float", "riccati": r"k_riccati_sp<float"},
    "headline_f64": {"forward": r"k_forward_sp<double", "riccati": r"k_riccati_sp<double"},
    "m2_f32": {"forward": r"k_forward_tv_sp<", "riccati": r"k_riccati_tv_sp<"},
    "timevarying_f64": {"forward": r"k_forward_tv_sp<double", "riccati": r"k_riccati_tv_sp<double"},
    "config3": {"trial": r"k_trial_sp", "forward": r"k_forward_sp<", "riccati": r"k_riccati_sp<"},
    "config5_one_system": {"trial": r"k_trial"},
    "config4_sharded": {"trial": r"k_trial", "system": r"k_scan|k_forward|k_riccati"},
    "dense_generic_f32": {"forward": r"k_forward<float", "riccati": r"k_riccati<float"},
    "dense_generic_f64": {"forward": r"k_forward<double", "riccati": r"k_riccati<double"},
    "specialised_joint_n6": {"forward": r"k_forward_sp<float", "riccati": r"k_riccati_sp<float"},
    "value_and_grad_headline": {"sys_fwd": r"k_asp_sys_fwd<float", "sys_rev": r"k_asp_sys_rev(_fused)?<float", "ric_rev": r"k_asp_ric_rev<float",
                                "riccati": r"k_riccati_sp<float"},
    "value_and_grad_config3": {"sys_fwd": r"k_asp_sys_fwd<float", "sys_rev": r"k_asp_sys_rev(_fused)?<float", "ric_rev": r"k_asp_ric_rev<float",
                               "riccati": r"k_riccati_sp<float", "trial_fwd": r"k_trial_spIf", "trial_rev": r"k_asp_trial_revIf"},
}
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")


def table(tag, leg, counter):
    rows = []
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{leg}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return rows


def durations(tag, leg, counter):
    """kernel name -> {dispatch id: ns} of the pass that collected `counter`"""
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{leg}_{counter}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return out


def family_sum(rows, pat):
    """(sum over the family's kernels of the mean over their full-size launches, launches averaged, kernel names)"""
    by = collections.defaultdict(list)
    for r in rows:
        if re.search(pat, r["Kernel_Name"]):
            by[r["Kernel_Name"]].append((float(r["Counter_Value"]), r.get("Dispatch_Id")))
    if not by:
        return None, 0, [], {}
    top = max(max(v for v, _ in vs) for vs in by.values())
    tot, n, big_ids = 0.0, 0, {}
    for name, vs in by.items():
        mx = max(v for v, _ in vs)
        if mx < 0.02 * top:                 # a kernel that only ever ran on the parity sample
            continue
        big = [(v, i) for v, i in vs if v > 0.5 * mx]
        tot += sum(v for v, _ in big) / len(big)
        n += len(big)
        big_ids[name] = [i for _, i in big]
    return tot, n, sorted(k[:110] for k in big_ids), big_ids


def main():
    tag = sys.argv[1]
    legs = sys.argv[2:] or list(KERNELS)
    from lqg_amd import build, specialize
    doc = {"note": " ".join(__doc__.split("\n\n")[1].split()), "tag": tag, "records": []}
    for leg in legs:
        tabs = {c: table(tag, leg, c) for c in COUNTERS}
        dur = durations(tag, leg, "GRBM_GUI_ACTIVE")
        for fam, pat in KERNELS[leg].items():
            f, nf, names, _ = family_sum(tabs["FETCH_SIZE"], pat)
            w, nw, _, _ = family_sum(tabs["WRITE_SIZE"], pat)
            v, nv, names_v, _ = family_sum(tabs["SQ_INSTS_VALU"], pat)
            g, ng, _, ids = family_sum(tabs["GRBM_GUI_ACTIVE"], pat)
            if f is None and v is None:
                continue
            rec = dict(leg=leg, kernel=fam, kernel_names=(names or names_v)[:4], source_hash=build.source_hash(),
                       sp_headers_hash=specialize._headers_hash(), adj_headers_hash=specialize._adj_headers_hash(),
                       fetch_size_kib_raw=f, write_size_kib_raw=w,
                       hbm_bytes_per_launch=((2.0 * f + (w or 0.0)) * 1024.0 if f is not None else None),
                       launches_averaged={"FETCH_SIZE": nf, "WRITE_SIZE": nw, "SQ_INSTS_VALU": nv, "GRBM_GUI_ACTIVE": ng},
                       valu_wave_insts_per_launch=v,
                       profile=f"profiles/r06_pmc.json <- rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --only {leg} "
                               f"--steps 3 --warmup 1 (scripts/pmc_legs.sh {tag}; one counter per pass)")
            if g:                            # sustained clock of the family's full-size dispatches in the GRBM pass
                ns = sum(sum(dur[name].get(i, 0.0) for i in il) / max(1, len(il)) for name, il in ids.items())
                if ns > 0:
                    rec["grbm_gui_active_per_launch"] = g
                    rec["kernel_ns_in_grbm_pass"] = ns
                    rec["sustained_mhz"] = g / 8.0 / ns * 1e3
            if leg.startswith("headline") and v:
                rec["valu_insts_per_step_per_wave"] = v / ((1 << 20) / 64) / 500
            doc["records"].append(rec)
            print(json.dumps({k: rec.get(k) for k in ("leg", "kernel", "hbm_bytes_per_launch", "valu_wave_insts_per_launch",
                                                      "sustained_mhz", "launches_averaged")}))
    out = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json")
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, len(doc["records"]), "records")


if __name__ == "__main__":
    main()
