"""The contract line of bench.py: ONE JSON object on the last stdout line, small enough for the driver to parse and to keep whole.

Round 5 inlined 16 secondary legs into that line (24.6 kB) and the driver's record came back `parsed: null`.  The line is now
the headline only — `metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline, parity,
collective` — plus `extra_summary = {leg: [value, unit, frac]}`; the full legs are written to `gpurun_out/bench_extra.json`
(scratch; copied to `profiles/` when kept).  No GPU and no torch in this module: `tests/test_bench_line.py` runs it on a
canned dict.
"""
import json
import os

LINE_LIMIT = 8000          # characters of the final stdout line (the driver's record keeps an 8 kB tail of stdout)
EXTRA_FILE = os.path.join("gpurun_out", "bench_extra.json")

# keys the driver's contract + the judge's brief name; never dropped when the line is squeezed
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "riccati_kernel_ms",
                 "algorithmic_bytes_per_solve", "algorithmic_bytes_per_launch", "kernel_launches_per_step", "whole_step_frac",
                 "limits")
LIMITS_KEEP = ("bound", "hbm_frac_of_peak", "hbm_frac_of_copy_rate", "valu_issue_frac", "valu_issue_frac_at_sustained_clock",
               "valu_insts_per_step_per_wave", "sustained_mhz", "profile")
CPU_KEEP = ("value", "unit", "cores", "kind", "sample", "cpu_model", "single_thread", "other_dtype", "port")


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _leg_value(leg):
    """(value, unit) of one secondary leg: its throughput when it states one, else its time."""
    if "value" in leg:
        return leg["value"], leg.get("unit")
    for key, unit in (("solves_per_s", "solves/s"), ("evaluations_per_s", "evaluations/s"), ("ms_per_pass", "ms/pass"),
                      ("ms_per_value_and_grad", "ms"), ("ms_per_step", "ms/step"), ("wall_ms", "ms")):
        if key in leg:
            return leg[key], unit
    for sub in ("time_parallel", "trials_1", "candidates_64", "sequential_cooperative"):       # nested timing records
        if isinstance(leg.get(sub), dict):
            v, u = _leg_value(leg[sub])
            if v is not None:
                return v, u if u else sub
    return None, leg.get("unit")


def summarise_leg(leg):
    """`[value, unit, frac]` of a secondary leg (frac = its own roofline fraction, None when the leg is latency-bound and
    states no roofline); `["error", message, None]` for a leg that raised."""
    if not isinstance(leg, dict):
        return [None, None, None]
    if "error" in leg:
        return ["error", _clip(str(leg["error"]), 120), None]
    v, u = _leg_value(leg)
    roof = leg.get("roofline") if isinstance(leg.get("roofline"), dict) else {}
    frac = roof.get("frac", leg.get("frac_hbm_algorithmic"))
    r = lambda a: round(a, 4) if isinstance(a, float) else a
    if isinstance(v, float):
        v = float(f"{v:.6g}")
    return [v, _clip(u, 48), r(frac)]


def _headline(out):
    """The headline dict trimmed to the contract's keys + the brief's: long prose clipped, diagnostics kept in the side file."""
    line = {}
    for k, v in out.items():
        if k in ("extra", "extra_summary"):
            continue
        line[k] = v
    roof = out.get("roofline")
    if isinstance(roof, dict):
        r = {k: roof[k] for k in ROOFLINE_KEEP if k in roof}
        if isinstance(r.get("limits"), dict):
            r["limits"] = {k: _clip(r["limits"][k], 160) for k in LIMITS_KEEP if k in r["limits"]}
        r["kernel"] = _clip(r.get("kernel"), 200)
        line["roofline"] = r
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = {k: cpu[k] for k in CPU_KEEP if k in cpu}
        for k in ("sample", "port"):
            if k in c:
                c[k] = _clip(c[k], 240)
        line["cpu_baseline"] = c
    cfg = out.get("config")
    if isinstance(cfg, dict):
        line["config"] = {k: _clip(v, 240) for k, v in cfg.items()}
    return line


def _strict(o):
    """NaN / inf are not JSON: a non-finite float becomes null (the line's `all_finite` says so separately)."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {str(k): _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    return o


def compact(out, extra=None, limit=LINE_LIMIT):
    """Return the dict to print as the last stdout line: the trimmed headline + `extra_summary`.  Squeezes optional keys until
    `len(json.dumps(line)) < limit`; raises if the REQUIRED keys alone do not fit (a bug to see here, not in the driver)."""
    line = _strict(_headline(out))
    if extra is not None:
        extra = _strict(extra)
        if isinstance(extra, dict) and set(extra) == {"error"}:
            line["extra_summary"] = {"error": _clip(str(extra["error"]), 200)}
        else:
            line["extra_summary"] = {k: summarise_leg(v) for k, v in extra.items()}
        line["extra_file"] = EXTRA_FILE
    # squeeze, least important first
    droppable = ["per_rank_solves_per_s", "per_rank_objective", "allreduce_us_percentiles", "ms_per_step_min",
                 "ms_per_step_median", "share_gpu", "world_size", "objective_sum", "all_finite", "allreduce_us",
                 "extra_file", "collective", "parity", "extra_summary"]
    while len(json.dumps(line)) >= limit and droppable:
        line.pop(droppable.pop(0), None)
    if len(json.dumps(line)) >= limit:
        raise ValueError(f"bench line of {len(json.dumps(line))} characters with only the required keys: limit {limit}")
    return line


def write_extra(out, extra, path=EXTRA_FILE):
    """The full record (headline untrimmed + every leg) to the side file; never fatal (a read-only tree costs only the file)."""
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        full = _strict(dict(out))
        full["extra"] = _strict(extra)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return path
    except OSError:
        return None


def check_line(text, limit=LINE_LIMIT):
    """What the driver must be able to do with bench.py's stdout: exactly one line starting with '{', it is the LAST non-empty
    line, strict JSON, shorter than `limit`, with every required key and both added objects.  Returns the parsed dict."""
    lines = [l for l in text.splitlines() if l.strip()]
    objs = [l for l in lines if l.lstrip().startswith("{")]
    if len(objs) != 1:
        raise AssertionError(f"{len(objs)} lines start with '{{' (want exactly 1)")
    if lines[-1] is not objs[0]:
        raise AssertionError("the JSON line is not the last stdout line")
    if len(objs[0]) >= limit:
        raise AssertionError(f"JSON line of {len(objs[0])} characters (limit {limit})")

    def no_const(c):
        raise AssertionError(f"non-strict JSON constant {c}")
    d = json.loads(objs[0], parse_constant=no_const)
    missing = [k for k in REQUIRED if k not in d]
    if missing:
        raise AssertionError(f"missing keys {missing}")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        if k not in d["roofline"]:
            raise AssertionError(f"roofline.{k} missing")
    for k in ("value", "unit", "cores", "kind", "sample"):
        if k not in d["cpu_baseline"]:
            raise AssertionError(f"cpu_baseline.{k} missing")
    if "workload" not in d["config"]:
        raise AssertionError("config.workload missing")
    return d
