"""Build liblqg_hip.so for gfx950 with hipcc: one translation unit per kernel instantiation of
csrc/lqg_dims.def, compiled in parallel, linked into lqg_amd/csrc/liblqg_hip.so (in-tree, so that it
travels to the GPU box).  Incremental: an object is rebuilt only when a source it includes is newer.

    python -m lqg_amd.build [-j N] [--force]
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "liblqg_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
HEADERS = ["lqg_small.hpp", "lqg_kernels.hpp", "lqg_launch.hpp", "lqg_dims.def", "../../include/lqg_hip.h"]


def dims_lists():
    """Parse the X-macro lists of lqg_dims.def -> {family: [tuple, ...]}."""
    text = open(os.path.join(CSRC, "lqg_dims.def")).read().replace("\\\n", " ")
    out = {}
    for fam in ("FORWARD", "RICCATI", "KALMAN", "TRIAL", "SIM"):
        m = re.search(r"#define LQG_%s_DIMS\(X\)(.*)" % fam, text)
        out[fam] = [tuple(int(v) for v in t.split(",")) for t in re.findall(r"X\(([^)]*)\)", m.group(1))]
    return out


def jobs():
    js = [("abi.o", "lqg_abi.hip", [])]
    for fam, tuples in dims_lists().items():
        for t in tuples:
            name = f"{fam.lower()}_{'_'.join(map(str, t))}.o"
            js.append((name, "lqg_inst.hip", [f"-DLQG_INST_{fam}={','.join(map(str, t))}"]))
    return js


STAMP = LIB + ".stamp"


def source_hash():
    """Content hash of every source that goes into the library (+ compile flags): mtimes do not survive the
    snapshot to the GPU box, the hash does."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(HEADERS + ["lqg_abi.hip", "lqg_inst.hip"]):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def up_to_date():
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == source_hash()


def _newest_src():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + ["lqg_abi.hip", "lqg_inst.hip"])


def _compile(job):
    name, src, defs = job
    out = os.path.join(OBJ, name)
    cmd = [HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr


def build(force=False, workers=None, verbose=True):
    if not force and up_to_date():
        if verbose:
            print(f"[lqg_amd.build] {LIB} is up to date", flush=True)
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    newest = _newest_src()
    todo = [j for j in jobs() if force or not os.path.exists(os.path.join(OBJ, j[0]))
            or os.path.getmtime(os.path.join(OBJ, j[0])) < newest]
    workers = workers or min(8, os.cpu_count() or 1)
    if todo:
        if verbose:
            print(f"[lqg_amd.build] compiling {len(todo)} translation units for {ARCH} with {workers} workers",
                  flush=True)
        # biggest kernels first so the tail of the parallel build is short
        todo.sort(key=lambda j: -sum(int(v) ** 3 for v in re.findall(r"\d+", " ".join(j[2]))))
        with cf.ThreadPoolExecutor(workers) as ex:
            for name, rc, err in ex.map(_compile, todo):
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {name}:\n{err}")
                if verbose:
                    print(f"[lqg_amd.build]   {name}", flush=True)
    objs = [os.path.join(OBJ, j[0]) for j in jobs()]
    if todo or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print(f"[lqg_amd.build] linked {LIB}", flush=True)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    build(force=a.force, workers=a.j)
