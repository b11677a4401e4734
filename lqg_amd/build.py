"""Build liblqg_hip.so for gfx950 with hipcc: one translation unit per kernel instantiation of
csrc/lqg_dims.def, compiled in parallel, linked into lqg_amd/csrc/liblqg_hip.so (in-tree, so that it
travels to the GPU box).  Incremental: an object is rebuilt only when a source it includes is newer.

    python -m lqg_amd.build [-j N] [--force]
"""
import argparse
import concurrent.futures as cf
import contextlib
import fcntl
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "liblqg_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc otherwise packs pairs of scalar f32 FMAs into v_pk_fma_f32, which on gfx950 issues at
# half the instruction rate (no flop gain) and costs ~450 v_mov per step of operand shuffling plus ~300 extra
# live registers in k_forward (measured: 42.5 ms -> 12.3 ms per 2^18 solves, profiles/README.md).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize"]
HEADERS = ["lqg_small.hpp", "lqg_rng.hpp", "lqg_kernels.hpp", "lqg_launch.hpp", "lqg_trial_chunk.hpp", "lqg_dims.def", "../../include/lqg_hip.h"]
COOP_HEADERS = ["lqg_small.hpp", "lqg_rng.hpp", "lqg_kernels.hpp", "lqg_launch.hpp", "lqg_trial_chunk.hpp", "lqg_coop.hpp", "lqg_coop_launch.hpp",
                "../../include/lqg_hip.h"]
SCAN_HEADERS = COOP_HEADERS + ["lqg_scan.hpp"]
ADJ_HEADERS = ["lqg_small.hpp", "lqg_adjoint.hpp", "lqg_adjoint_launch.hpp", "../../include/lqg_hip.h"]
# headers each source depends on (an adjoint-kernel edit must not recompile the forward kernels and vice versa)
DEPS = {"lqg_inst.hip": HEADERS, "lqg_adjoint_inst.hip": ADJ_HEADERS, "lqg_coop_inst.hip": COOP_HEADERS,
        "lqg_scan_inst.hip": SCAN_HEADERS, "lqg_setup.hip": ["../../include/lqg_hip.h"],
        "lqg_coop_adjoint.hip": ["lqg_coop_launch.hpp", "../../include/lqg_hip.h"],
        "lqg_abi.hip": sorted(set(HEADERS + ADJ_HEADERS + ["lqg_coop_launch.hpp"]))}
FAMILIES = ("FORWARD", "RICCATI", "KALMAN", "TRIAL", "SIM", "ADJOINT")
ADJOINT_MAX_JOINT = 12          # on-demand libraries get the gradient sweep up to x + b = 12 (33 s per dtype at 12; minutes beyond)


def dims_lists():
    """Parse the X-macro lists of lqg_dims.def -> {family: [tuple, ...]}."""
    text = open(os.path.join(CSRC, "lqg_dims.def")).read().replace("\\\n", " ")
    out = {}
    for fam in FAMILIES:
        m = re.search(r"#define LQG_%s_DIMS\(X\)(.*)" % fam, text)
        out[fam] = [tuple(int(v) for v in t.split(",")) for t in re.findall(r"X\(([^)]*)\)", m.group(1))]
    return out


def jobs(lists=None, extra_defs=()):
    extra_defs = list(extra_defs)
    js = [("abi.o", "lqg_abi.hip", extra_defs), ("coop.o", "lqg_coop_inst.hip", extra_defs),
          ("scan.o", "lqg_scan_inst.hip", extra_defs), ("setup.o", "lqg_setup.hip", extra_defs),
          ("coopadj.o", "lqg_coop_adjoint.hip", extra_defs)]
    for fam, tuples in (lists or dims_lists()).items():
        for t in tuples:
            for dt in ("F32", "F64"):
                base = f"{fam.lower()}_{'_'.join(map(str, t))}_{dt.lower()}"
                defs = [f"-DLQG_INST_{fam}={','.join(map(str, t))}", f"-DLQG_INST_{dt}"] + extra_defs
                if fam == "ADJOINT":
                    js.append((base + ".o", "lqg_adjoint_inst.hip", defs))
                    continue
                if fam in ("FORWARD", "RICCATI"):   # fixed-dims cooperative kernels of the same shapes (lqg_coop.hpp)
                    cf_ = "COOPF" if fam == "FORWARD" else "COOPR"
                    if fam == "RICCATI" or t[0] + t[1] <= 11:      # WAVES mode: at most two elements per lane (m*m <= 128)
                        js.append((f"{cf_.lower()}_{'_'.join(map(str, t))}_{dt.lower()}.o", "lqg_coop_inst.hip",
                                   [f"-DLQG_INST_{cf_}={','.join(map(str, t))}", f"-DLQG_INST_{dt}"] + extra_defs))
                if fam == "ADJOINT":
                    pass
                elif fam == "FORWARD":        # the 8 k_forward variants of a (dims, dtype) are split over four units
                    js += [(f"{base}_v{v}.o", "lqg_inst.hip", defs + [f"-DLQG_INST_VARIANT={v}"]) for v in range(4)]
                else:
                    js.append((base + ".o", "lqg_inst.hip", defs))
    return js


STAMP = LIB + ".stamp"


def source_hash():
    """Content hash of every source that goes into the library (+ compile flags): mtimes do not survive the
    snapshot to the GPU box, the hash does."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(set(HEADERS + ADJ_HEADERS + SCAN_HEADERS + ["lqg_abi.hip", "lqg_inst.hip", "lqg_adjoint_inst.hip",
                                                                "lqg_coop_inst.hip", "lqg_scan_inst.hip", "lqg_setup.hip",
                                                                "lqg_coop_adjoint.hip"])):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def up_to_date():
    return os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == source_hash()


def _newest_src(src):
    """Newest mtime among the headers and the one source file an object is compiled from."""
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in DEPS.get(src, HEADERS) + [src])


def _compile(job):
    name, src, defs = job
    out = os.path.join(OBJ, name)
    cmd = [HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr


def build_variant(out, extra_flags=(), only=None, objdir=None, workers=None, verbose=True):
    """Developer tool: build a VARIANT library `out` with extra compile flags, recompiling only the
    translation units whose object name contains one of `only` (others are reused from the main build).
    Used for A/B kernel experiments on the GPU box (select with LQG_HIP_LIB=<out>)."""
    objdir = objdir or (out + ".obj")
    os.makedirs(objdir, exist_ok=True)
    sel = [j for j in jobs() if only is None or any(o in j[0] for o in only)]

    def comp(job):
        name, src, defs = job
        o = os.path.join(objdir, name)
        r = subprocess.run([HIPCC] + FLAGS + list(extra_flags) + defs + ["-c", os.path.join(CSRC, src), "-o", o],
                           capture_output=True, text=True)
        return name, r.returncode, r.stderr

    with cf.ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1)) as ex:
        for name, rc, err in ex.map(comp, sel):
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {name}:\n{err}")
            if verbose:
                print(f"[lqg_amd.build]   variant {name}", flush=True)
    chosen = {j[0] for j in sel}
    objs = [os.path.join(objdir if j[0] in chosen else OBJ, j[0]) for j in jobs()]
    r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs, capture_output=True,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return out


def cache_dir(preferred, name):
    """`preferred` (in-tree, so that the libraries travel with the snapshot) when it can be written, else a per-user
    cache directory (read-only installs)."""
    try:
        os.makedirs(preferred, exist_ok=True)
        if os.access(preferred, os.W_OK):
            return preferred
    except OSError:
        pass
    alt = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "lqg_amd", name)
    os.makedirs(alt, exist_ok=True)
    return alt


@contextlib.contextmanager
def locked(target):
    """Exclusive inter-process lock for producing `target` (several ranks of one node may ask for the same library at
    the same time): the first one in compiles, the others block here and then find the finished, stamped file."""
    with open(target + ".lock", "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def atomic_write(path, text):
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def stamped(so, want):
    """True when `so` exists with a stamp equal to `want` (the stamp is written after the library is in place)."""
    stamp = so + ".stamp"
    try:
        return os.path.exists(so) and open(stamp).read().strip() == want
    except OSError:
        return False


DIMS_DIR = os.environ.get("LQG_DIMS_DIR") or os.path.join(CSRC, "dims")


def dims_tag(x, b, u, y, d):
    return f"{x}_{b}_{u}_{y}_{d}"


def build_dims_library(x, b, u, y, d, workers=None, verbose=True):
    """Auxiliary library with the full C ABI for ONE model shape that lqg_dims.def does not list
    (<dims dir>/liblqg_hip_<x>_<b>_<u>_<y>_<d>.so): same sources, single-shape instantiation lists.
    Called on demand by lqg_amd._abi.library_for; cached by a hash of the sources.  Safe under concurrent callers
    (ranks of one node): inter-process lock, link to a temporary name, atomic rename, stamp last."""
    tag = dims_tag(x, b, u, y, d)
    ddir = cache_dir(DIMS_DIR, "dims")
    so = os.path.join(ddir, f"liblqg_hip_{tag}.so")
    want = source_hash()
    if stamped(so, want):
        return so
    with locked(so):
        if stamped(so, want):                      # another process built it while this one waited for the lock
            return so
        lists = {"FORWARD": [(x, b, u, y, d)], "RICCATI": [(b, u)], "KALMAN": [(b, y)], "TRIAL": [(x + b, d)],
                 "SIM": [(x, b, u, y)], "ADJOINT": [(x, b, u, y, d)] if x + b <= ADJOINT_MAX_JOINT else []}
        deff = os.path.join(ddir, f"dims_{tag}.def")
        text = f"// GENERATED by lqg_amd/build.py: instantiation lists of the auxiliary library for shape {tag}\n"
        for fam, tuples in lists.items():
            text += f"#define LQG_{fam}_DIMS(X) " + " ".join("X(" + ", ".join(map(str, t)) + ")" for t in tuples) + "\n"
        atomic_write(deff, text)
        objdir = os.path.join(ddir, f"obj_{tag}")
        os.makedirs(objdir, exist_ok=True)
        js = jobs(lists, extra_defs=[f'-DLQG_DIMS_DEF="{deff}"'])
        if verbose:
            print(f"[lqg_amd.build] compiling auxiliary library for shape (x,b,u,y,d)=({x},{b},{u},{y},{d}): "
                  f"{len(js)} translation units", flush=True)

        def comp(job):
            name, src, defs = job
            r = subprocess.run([HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, name)],
                               capture_output=True, text=True)
            return name, r.returncode, r.stderr

        with cf.ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1)) as ex:
            for name, rc, err in ex.map(comp, js):
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {name}:\n{err[-3000:]}")
        tmp = f"{so}.tmp.{os.getpid()}"
        r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] +
                           [os.path.join(objdir, j[0]) for j in js], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-3000:])
        os.replace(tmp, so)
        atomic_write(so + ".stamp", want)
    return so


def refresh_dims_libraries(verbose=True):
    """Rebuild every auxiliary per-shape library found in the dims directory whose stamp is not the current source hash
    (they travel to the GPU box with the tree; a stale one would be recompiled THERE, on GPU time, at first use)."""
    done = []
    if not os.path.isdir(DIMS_DIR):
        return done
    want = source_hash()
    for f in sorted(os.listdir(DIMS_DIR)):
        m = re.fullmatch(r"liblqg_hip_(\d+)_(\d+)_(\d+)_(\d+)_(\d+)\.so", f)
        if m and not stamped(os.path.join(DIMS_DIR, f), want):
            done.append(build_dims_library(*(int(v) for v in m.groups()), verbose=verbose))
    return done


def build(force=False, workers=None, verbose=True):
    if not force and up_to_date():
        if verbose:
            print(f"[lqg_amd.build] {LIB} is up to date", flush=True)
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = [j for j in jobs() if force or not os.path.exists(os.path.join(OBJ, j[0]))
            or os.path.getmtime(os.path.join(OBJ, j[0])) < _newest_src(j[1])]
    workers = workers or min(8, os.cpu_count() or 1)
    if todo:
        if verbose:
            print(f"[lqg_amd.build] compiling {len(todo)} translation units for {ARCH} with {workers} workers",
                  flush=True)
        # biggest kernels first so the tail of the parallel build is short
        todo.sort(key=lambda j: -sum(int(v) ** 3 for v in re.findall(r"\d+", j[2][0] if j[2] else "")) * (4 if "FORWARD" in " ".join(j[2]) else 1))
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
    ap.add_argument("--variant", default=None, help="output path of a variant library (developer A/B builds)")
    ap.add_argument("--only", default=None, help="comma-separated object-name substrings to recompile")
    ap.add_argument("--flags", default="", help="extra hipcc flags for the variant")
    a = ap.parse_args()
    if a.variant:
        build_variant(a.variant, a.flags.split(), a.only.split(",") if a.only else None, workers=a.j)
    else:
        build(force=a.force, workers=a.j)
