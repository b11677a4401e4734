"""The driver's own bench command, unabridged (VERDICT r05 task 1): `python3 bench.py --gpus 1 --steps 20 --warmup 5` — no
`--no-extra`, no `--no-cpu-baseline` — must end in ONE strict-JSON line under 8000 characters carrying the contract keys,
`roofline` and `cpu_baseline`, with the secondary legs summarised in it and written in full to gpurun_out/bench_extra.json."""
import json
import os
import subprocess
import sys

import pytest

import bench_line

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_command_prints_one_parseable_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    side = os.path.join(ROOT, bench_line.EXTRA_FILE)
    if os.path.exists(side):
        os.remove(side)
    r = subprocess.run(["python3", "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT, capture_output=True,
                       text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = bench_line.check_line(r.stdout)                      # one '{' line, last, < 8000 chars, strict JSON, required keys
    # the tail the driver's record keeps (8 kB of stdout) holds the whole line
    assert r.stdout[-8192:].lstrip().startswith("{") or "\n{" in r.stdout[-8192:]
    assert d["metric"].startswith("LQG solves/sec") and d["unit"] == "solves/s" and d["n_gpus"] == 1
    assert d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "n=6" in d["metric"] and d["config"]["T"] == 500 and d["config"]["solves_per_gpu"] == 2 ** 20
    assert d["value"] > 1e8 and abs(d["value"] * d["ms_per_step"] * 1e-3 / 2 ** 20 - 1) < 1e-9
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    # (limits.bound / traffic come from the committed PMC record and are present only when it was taken on exactly this build)
    assert roof["frac"] > 0.3 and 0 < roof["whole_step_frac"] < roof["frac"] and "limits" in roof and "traffic" in roof
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["single_thread"]["value"] > 0 and cpu["cpu_model"]
    assert d["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-6 and d["all_finite"]
    legs = d["extra_summary"]
    for leg in ("value_and_grad_headline", "value_and_grad_config3", "headline_f64", "specialised_joint_n6", "config3",
                "config4_sharded", "config5_one_system", "dense_generic_f64", "m2_f32"):
        assert leg in legs and len(legs[leg]) == 3 and legs[leg][0] != "error" and legs[leg][0] is not None, (leg, legs.get(leg))
    full = json.load(open(side))
    assert set(full["extra"]) == set(legs) and full["value"] == d["value"]
    assert "roofline" in full["extra"]["config3"]
