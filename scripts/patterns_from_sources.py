"""Turn generated pattern sources (a pattern directory that a GPU test run filled: scripts/r06_pattern_probe.sh) back into
(dims, masks[, live]) records — lqg_amd/csrc/test_patterns.json — so that __graft_entry__.build() compiles them here and a fresh
GPU box does not (40 s of hipcc apiece, on every box).
    python scripts/patterns_from_sources.py gpurun_out/pat_probe key [key ...]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lqg_amd import specialize

MASK = re.compile(r"static constexpr (?:lqg::)?Mask<(\d+), (\d+)> (\w+)\{\{([^}]*)\}\};")
LIVE = re.compile(r"static constexpr bool live_(\w+) = (true|false);")


def masks_of(text):
    out = {}
    for r, c, name, body in MASK.findall(text):
        vals = [v.strip() == "true" for v in body.split(",")]
        out[name] = np.array(vals, dtype=bool).reshape(int(r), int(c))
    return out


def main():
    pdir, keys = sys.argv[1], sys.argv[2:]
    dest = os.path.join(ROOT, "lqg_amd", "csrc", "test_patterns.json")
    doc = json.load(open(dest)) if os.path.exists(dest) else {"forward": [], "adjoint": []}
    have = {(r["key"]) for r in doc["forward"]} | {r["key"] for r in doc["adjoint"]}
    for key in keys:
        if key in have:
            continue
        if key.startswith("padj_"):
            k = key[5:]
            text = open(os.path.join(pdir, f"padj_{k}_pat.hpp")).read()
            inst = open(os.path.join(pdir, f"padj_{k}_f0.hip")).read()
            m = re.search(r"run_asp<float, lqg::Pat_\w+, (\d+), (\d+), (\d+), (\d+), (\d+),", inst)
            dims = dict(zip("xbuyd", map(int, m.groups())))
            masks = masks_of(text)
            live = {n: v == "true" for n, v in LIVE.findall(text)}
            base = k.split("_")[0]
            assert specialize.pattern_key(dims, masks) == base, (key, specialize.pattern_key(dims, masks))
            assert specialize.adjoint_key(base, live) == k, (key, specialize.adjoint_key(base, live))
            doc["adjoint"].append({"key": key, "dims": dims, "live": live, "masks": {n: v.astype(int).tolist() for n, v in masks.items()}})
        else:
            k = key[4:] if key.startswith("pat_") else key
            text = open(os.path.join(pdir, f"pat_{k}.hip")).read()
            m = re.search(r"log_likelihood_sp<Pat, (\d+), (\d+), (\d+), (\d+), (\d+)>", text)
            dims = dict(zip("xbuyd", map(int, m.groups())))
            masks = masks_of(text)
            assert specialize.pattern_key(dims, masks) == k, (key, specialize.pattern_key(dims, masks))
            doc["forward"].append({"key": "pat_" + k, "dims": dims, "masks": {n: v.astype(int).tolist() for n, v in masks.items()}})
    json.dump(doc, open(dest, "w"))
    print("wrote", dest, len(doc["forward"]), "forward,", len(doc["adjoint"]), "adjoint")


if __name__ == "__main__":
    main()
