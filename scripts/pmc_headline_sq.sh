#!/bin/bash
# SQ wait / issue counters and the sustained clock (GRBM_GUI_ACTIVE / 8 / kernel time, MI355X_MICROARCH.md "DVFS give-back") of the
# headline kernels (python bench.py, fp32).  One counter group per pass, --kernel-trace only (no other trace domain).
# Writes gpurun_out/r05_headline_sq.txt (copy to profiles/).
#   bash scripts/pmc_headline_sq.sh [OUT] [BENCH_ARGS] [KERNEL_SUBSTRINGS, comma separated]
# e.g. the per-trial kernels of the config-3 value + gradient leg:
#   bash scripts/pmc_headline_sq.sh gpurun_out/r05_vg_config3_sq.txt "--only value_and_grad_config3" k_trial_sp,k_asp_trial_rev
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r05_headline_sq.txt}
BARGS=${2:---no-extra --no-cpu-baseline --steps 10 --warmup 2}
export LQG_SQ_KERNELS=${3:-k_forward_sp,k_riccati_sp}
export LQG_SQ_BARGS="$BARGS"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmch_$tag
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmch_$tag -o p -- python3 bench.py $BARGS > /dev/null 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
subs = os.environ["LQG_SQ_KERNELS"].split(",")
want = lambda k: any(q in k for q in subs)
def short(k):                       # a readable, distinguishing prefix of the (possibly mangled) kernel name
    for q in subs:
        if q in k:
            return q + " " + k[k.index(q) + len(q):][:34]
    return k[:40]
for f in glob.glob("gpurun_out/pmch_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want(k):
            a = acc[short(k)][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
# kernel durations of the GRBM pass itself (the clock is cycles / time of the SAME dispatches)
for f in glob.glob("gpurun_out/pmch_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want(k):
            d = dur[short(k)]
            d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(sys.argv[1], "w") as out:
    def P(s):
        print(s); out.write(s + "\n")
    P("# scripts/pmc_headline_sq.sh: python3 bench.py %s, one counter group per pass" % os.environ["LQG_SQ_BARGS"])
    for k, d in acc.items():
        P(k)
        for c, (m, v) in sorted(d.items()):
            P("   %-24s %.4e per call (%d calls)" % (c, v / m, m))
        per = lambda c: d[c][1] / d[c][0] if d[c][0] else float("nan")
        w = per("SQ_WAVE_CYCLES")
        P("   wait_any/wave_cycles %.3f   wait_inst_any/wave_cycles %.3f   active_valu/busy(4 SIMD-cycles) %.3f" % (
            per("SQ_WAIT_ANY") / w, per("SQ_WAIT_INST_ANY") / w, per("SQ_ACTIVE_INST_VALU") / max(per("SQ_BUSY_CYCLES"), 1)))
        if dur[k][0] and d["GRBM_GUI_ACTIVE"][0]:
            ns = dur[k][1] / dur[k][0]
            mhz = per("GRBM_GUI_ACTIVE") / 8.0 / ns * 1e3
            P("   kernel %.4f ms (in the GRBM pass)   sustained clock %.0f MHz (GRBM_GUI_ACTIVE / 8 / time)" % (ns * 1e-6, mhz))
            # VALU issue: wave-instructions / (1024 SIMDs x clock x time / 2 cycles per wave64 fp32 instruction on the SIMD-32)
            P("   valu_issue_frac_at_sustained_clock %.3f   at 2400 MHz %.3f" % (
                per("SQ_INSTS_VALU") / (1024 * mhz * 1e6 * ns * 1e-9 / 2.0), per("SQ_INSTS_VALU") / (1024 * 2.4e9 * ns * 1e-9 / 2.0)))
PY
