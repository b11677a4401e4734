"""The contract line of bench.py (bench_line.py) on canned dicts: round 5's real 24.6 kB record must come out as ONE strict-JSON
line under 8000 characters that still carries every key the driver and the judge read (task 1)."""
import json
import os

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _r05():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_f32.json")))


def test_round5_record_compacts_under_the_limit():
    full = _r05()
    assert len(json.dumps(full)) > 20000                      # the line that came back `parsed: null`
    extra = full.pop("extra")
    line = bench_line.compact(full, extra)
    text = json.dumps(line)
    assert len(text) < bench_line.LINE_LIMIT
    d = bench_line.check_line("some earlier log line\n" + text + "\n")
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["roofline"]["whole_step_frac"] == full["roofline"]["whole_step_frac"]
    assert d["roofline"]["limits"]["bound"].startswith("VALU")
    assert d["cpu_baseline"]["single_thread"]["value"] == full["cpu_baseline"]["single_thread"]["value"]
    assert d["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-6
    assert set(d["extra_summary"]) == set(extra)
    v, unit, frac = d["extra_summary"]["specialised_joint_n6"]
    assert unit == "solves/s" and 7e7 < v < 9e7 and 0.1 < frac < 0.2
    assert d["extra_summary"]["value_and_grad_headline"][1] == "solves+gradient/s"
    assert all(len(s) == 3 for s in d["extra_summary"].values())
    assert all(s[0] is not None for s in d["extra_summary"].values()), d["extra_summary"]


def test_non_finite_values_and_failed_legs_stay_strict_json():
    full = _r05()
    extra = full.pop("extra")
    full["objective_sum"] = float("nan")
    extra["dense_generic_f64"] = {"error": "RuntimeError('x' * 1000)" + "x" * 1000}
    extra["config3"]["value"] = float("inf")
    text = json.dumps(bench_line.compact(full, extra))
    d = bench_line.check_line(text)
    assert d["objective_sum"] is None
    assert d["extra_summary"]["dense_generic_f64"][0] == "error" and len(d["extra_summary"]["dense_generic_f64"][1]) <= 120
    assert d["extra_summary"]["config3"][0] is None


def test_squeeze_keeps_required_keys():
    full = _r05()
    extra = full.pop("extra")
    full["per_rank_objective"] = [1.0] * 2000                # e.g. many ranks: optional keys go first
    d = bench_line.check_line(json.dumps(bench_line.compact(full, extra)))
    assert "per_rank_objective" not in d and "extra_summary" in d


def test_check_line_rejects_what_broke_round5():
    full = _r05()
    with pytest.raises(AssertionError, match="characters"):
        bench_line.check_line(json.dumps(full))
    extra = full.pop("extra")
    good = json.dumps(bench_line.compact(full, extra))
    with pytest.raises(AssertionError, match="exactly 1"):
        bench_line.check_line(good + "\n" + good)
    with pytest.raises(AssertionError, match="last"):
        bench_line.check_line(good + "\ntrailing log")
    with pytest.raises(AssertionError, match="missing"):
        bench_line.check_line(json.dumps({k: v for k, v in json.loads(good).items() if k != "roofline"}))


def test_write_extra_round_trips(tmp_path):
    full = _r05()
    extra = full.pop("extra")
    p = bench_line.write_extra(full, extra, str(tmp_path / "sub" / "bench_extra.json"))
    back = json.load(open(p))
    assert set(back["extra"]) == set(extra) and back["value"] == full["value"]
