#!/bin/bash
# kernel-level timeline of one evaluation of DelayedSubjectiveActor (m = 65, T = 500) on the time-parallel sweeps
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_delay12 -o p -- python3 scripts/delay12_time.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_delay12/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "scan" in r["Kernel_Name"] or "trial" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "build_rk" in r["Kernel_Name"]] + [len(rows)]
evals = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
span = lambda q: int(q[-1]["End_Timestamp"]) - int(q[0]["Start_Timestamp"])
for which in (4, len(evals) - 2):                     # one fp32 and one fp64 evaluation
    seq = evals[which]
    t0 = int(seq[0]["Start_Timestamp"])
    print("# evaluation %d of %d: start us, duration us, grid, kernel" % (which, len(evals)))
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f  %6s x %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Kernel_Name"][:80]))
    print("# span us", span(seq) / 1e3)
PY
find gpurun_out/prof_delay12 -type f ! -name "*stats*.csv" -delete
