"""ctypes mirror of include/lqg_hip.h and the loader for liblqg_hip.so.

The HIP library is the ONLY compute backend of this package: if it cannot be loaded the import of any
compute entry point raises (no CPU fallback, by design — see DESIGN.md "Boundary").
"""
import ctypes as C
import os

from lqg_amd import options as _options

ABI_VERSION = 3
F32, F64, F32_SYS64 = 0, 1, 2      # F32_SYS64: include/lqg_hip.h (fp32 problem, fp64 spec arrays and system sweeps)
OP_LOG_LIKELIHOOD, OP_CONDITIONAL_MOMENTS = 0, 1
FAM_FORWARD, FAM_RICCATI, FAM_KALMAN, FAM_TRIAL, FAM_SIMULATE, FAM_ADJOINT = range(6)
# which of (x, b, u, y, d) a kernel family is instantiated on (include/lqg_hip.h: lqg_kernel_supported)
_FAMILY_KEYS = {FAM_FORWARD: "xbuyd", FAM_RICCATI: "bu", FAM_KALMAN: "by", FAM_SIMULATE: "xbuy", FAM_ADJOINT: "xbuyd"}

_HERE = os.path.dirname(os.path.abspath(__file__))
# LQG_HIP_LIB selects a variant build of the same library (developer A/B kernel experiments); never a fallback
LIB_PATH = os.environ.get("LQG_HIP_LIB") or os.path.join(_HERE, "csrc", "liblqg_hip.so")


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("st", C.c_int64), ("sr", C.c_int64), ("sc", C.c_int64)]


class Traj(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("sn", C.c_int64), ("st", C.c_int64), ("sd", C.c_int64)]


SPEC_FIELDS = ("Q", "q", "Qf", "qf", "P", "R", "r", "A", "B", "V", "F", "W")


class Spec(C.Structure):
    _fields_ = [(f, View) for f in SPEC_FIELDS]


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x", "b", "u", "y", "d", "nva", "nwa", "nvd", "nwd")]


class Tuning(C.Structure):
    """include/lqg_hip.h: lqg_tuning (all zero = the library's default rules; filled by lqg_amd.options.fill_tuning)."""
    _fields_ = [(n, C.c_int32) for n in ("coop", "trial_chunks", "trial_chunk_waves", "trial_chunk_max_waves", "trial_chunk_tpl",
                                          "coop_trial_chunks", "coop_trial_rows", "coop_sparse", "scan_lane", "scan_rt_waves",
                                          "coop_adjoint", "scan_order", "coop_trial_tpb", "coop_trial_wide", "trial_lds", "hilo")]


class Problem(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("T", C.c_int32), ("n_sys", C.c_int64), ("n_trials", C.c_int64),
                ("dims", Dims), ("actor", Spec), ("dynamics", Spec), ("Sigma0", View), ("eps", C.c_double),
                ("phase_events", C.c_void_p * 4), ("tuning", Tuning)]


NULL_VIEW = View(None, 0, 0, 0, 0)
NULL_TRAJ = Traj(None, 0, 0, 0, 0)


class LqgHipError(RuntimeError):
    pass


_lib = None


def declare(lib, prefix="lqg_", with_stream=True):
    """Attach argtypes/restype for the entry points of include/lqg_hip.h.  `prefix` / `with_stream` let a
    test harness bind host-side twins of the same signatures (without workspace / stream arguments)."""
    P = C.POINTER(Problem)
    tail = [C.c_void_p] if with_stream else []
    ws = [C.c_void_p, C.c_size_t] if with_stream else []
    sig = {
        "riccati_backward": [P, View, View, View] + tail,
        "kalman_forward": [P, View] + tail,
        "conditional_moments": [P, Traj, Traj, View] + ws + tail,
        "log_likelihood": [P, Traj, C.c_void_p, C.c_int64, C.c_int64] + ws + tail,
        "solve_materialised": [P, Traj, View, View, View, View, Traj, View, C.c_void_p, C.c_int64, C.c_int64] + ws + tail,
        "simulate": [P, View, View, View, Traj, Traj, View, View, Traj, Traj, Traj, Traj] + tail,
        "simulate_rng": [P, View, View, View, C.c_uint64, View, View, Traj, Traj, Traj, Traj] + tail,
    }
    for name, args in sig.items():
        if not hasattr(lib, prefix + name):
            continue
        fn = getattr(lib, prefix + name)
        fn.argtypes, fn.restype = args, C.c_int
    return lib


def _bind(path):
    lib = C.CDLL(path)
    lib.lqg_abi_version.restype = C.c_int
    if lib.lqg_abi_version() != ABI_VERSION:
        raise LqgHipError(f"ABI mismatch: library {lib.lqg_abi_version()} vs binding {ABI_VERSION}")
    lib.lqg_last_error.restype = C.c_char_p
    lib.lqg_target_arch.restype = C.c_char_p
    lib.lqg_dims_supported.argtypes, lib.lqg_dims_supported.restype = [C.c_int32, C.POINTER(Dims)], C.c_int
    lib.lqg_kernel_supported.argtypes, lib.lqg_kernel_supported.restype = [C.c_int32, C.POINTER(Dims)], C.c_int
    lib.lqg_coop_supported.argtypes, lib.lqg_coop_supported.restype = [C.POINTER(Dims)], C.c_int
    lib.lqg_strategy.argtypes, lib.lqg_strategy.restype = [C.POINTER(Problem)], C.c_int
    if hasattr(lib, "lqg_log_likelihood_scan"):
        lib.lqg_scan_supported.argtypes, lib.lqg_scan_supported.restype = [C.POINTER(Problem)], C.c_int
        lib.lqg_scan_workspace_bytes.argtypes, lib.lqg_scan_workspace_bytes.restype = [C.POINTER(Problem)], C.c_size_t
        lib.lqg_log_likelihood_scan.argtypes = [C.POINTER(Problem), Traj, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                C.c_size_t, C.c_void_p]
        lib.lqg_log_likelihood_scan.restype = C.c_int
        lib.lqg_log_likelihood_scan_with.argtypes = lib.lqg_log_likelihood_scan.argtypes + [C.c_void_p]
        lib.lqg_log_likelihood_scan_with.restype = C.c_int
        lib.lqg_conditional_moments_scan.argtypes = [C.POINTER(Problem), Traj, Traj, View, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.lqg_conditional_moments_scan.restype = C.c_int
    if hasattr(lib, "lqg_precondition_flags"):
        lib.lqg_precondition_flags.argtypes = [C.POINTER(Problem), C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
        lib.lqg_precondition_flags.restype = C.c_int
    if hasattr(lib, "lqg_fd_candidates"):
        lib.lqg_fd_candidates.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_double, C.c_void_p]
        lib.lqg_fd_candidates.restype = C.c_int
        lib.lqg_fd_combine.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_double, C.c_void_p]
        lib.lqg_fd_combine.restype = C.c_int
    if hasattr(lib, "lqg_point_mass_setup"):
        lib.lqg_point_mass_setup.argtypes = [C.c_int64] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 4
        lib.lqg_point_mass_setup.restype = C.c_int
    lib.lqg_workspace_bytes.argtypes, lib.lqg_workspace_bytes.restype = [C.POINTER(Problem), C.c_int32], C.c_size_t
    lib.lqg_sum_trials.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lqg_sum_trials.restype = C.c_int
    lib.lqg_sum_trials_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.lqg_sum_trials_workspace_bytes.restype = C.c_size_t
    lib.lqg_gaussian_logprob.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, Traj, Traj, View,
                                         C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.lqg_gaussian_logprob.restype = C.c_int
    if hasattr(lib, "lqg_log_likelihood_grad"):
        lib.lqg_grad_supported.argtypes, lib.lqg_grad_supported.restype = [C.c_int32, C.POINTER(Dims)], C.c_int
        lib.lqg_grad_elements.argtypes, lib.lqg_grad_elements.restype = [C.POINTER(Dims)], C.c_int64
        lib.lqg_grad_slabs.argtypes, lib.lqg_grad_slabs.restype = [C.POINTER(Problem)], C.c_int32
        lib.lqg_grad_lanes_per_system.argtypes, lib.lqg_grad_lanes_per_system.restype = [C.POINTER(Problem)], C.c_int32
        lib.lqg_grad_workspace_bytes.argtypes = [C.POINTER(Problem), C.c_int64]
        lib.lqg_grad_workspace_bytes.restype = C.c_size_t
        lib.lqg_log_likelihood_grad.argtypes = [C.POINTER(Problem), Traj, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                                C.c_int32, C.c_void_p]
        lib.lqg_log_likelihood_grad.restype = C.c_int
    declare(lib)
    return lib


def load():
    """Load liblqg_hip.so (built in-tree by __graft_entry__.build() / lqg_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LqgHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lqg_amd/csrc`. lqg_amd has no CPU fallback.")
    _lib = _bind(LIB_PATH)
    return _lib


_dims_libs = {}
# The kernels keep one system's matrices in registers: beyond these bounds an on-demand build would take hours of
# hipcc time and spill everything, so such shapes are rejected (a wave-per-system kernel is the fix).
MAX_STATE, MAX_JOINT, MAX_IO = 10, 20, 4


def shape_in_range(x, b, u, y, d):
    return 1 <= x <= MAX_STATE and 1 <= b <= MAX_STATE and x + b <= MAX_JOINT and 1 <= u <= MAX_IO \
        and 1 <= y <= MAX_IO and 1 <= d <= x


def _dims_struct(dims):
    return Dims(dims["x"], dims["b"], dims["u"], dims["y"], dims["d"], dims.get("nva", 1), dims.get("nwa", 1),
                dims.get("nvd", 1), dims.get("nwd", 1))


# Below this many systems per call a shape without lane kernels runs on the cooperative run-time-dims kernels of the
# main library (no compile); above it an auxiliary lane-kernel library is compiled once (8-30 s) and cached.
JIT_MIN_SYSTEMS = 4096
STRATEGY_LANE, STRATEGY_COOP = 0, 1


def library_for(dims, family=FAM_FORWARD, n_sys=None):
    """The library that holds the kernels of `family` for this model shape.  lqr.backward only depends on (b, u),
    kf.forward on (b, y), simulate on (x, b, u, y) — they are resolved per FAMILY, so e.g. the Riccati sweep of a
    SubjectiveActor's actor spec (b=6, u=2) is served by the main library whatever x, y, d are.  Order: lane kernels of
    liblqg_hip.so; auxiliary libraries already loaded or cached on disk; the COOPERATIVE kernels of liblqg_hip.so
    (run-time dims: any x, b with u, y, d <= 4 — no compile) for small batches, shapes beyond the lane kernels' range
    (x + b > 20) or when no compiler is there; finally an auxiliary lane-kernel library compiled on demand for exactly
    this shape (lqg_amd.build.build_dims_library; needs hipcc).  Never a CPU fallback."""
    lib = load()
    dm = _dims_struct(dims)
    if lib.lqg_kernel_supported(family, C.byref(dm)):
        return lib
    for aux in _dims_libs.values():
        if aux.lqg_kernel_supported(family, C.byref(dm)):
            return aux
    from lqg_amd import build
    key0 = (dims["x"], dims["b"], dims["u"], dims["y"], dims["d"] if 1 <= dims["d"] <= dims["x"] else min(dims["x"], 2))
    dc = dict(dims)                                 # fields the family ignores must not disqualify the shape
    if family == FAM_RICCATI:
        dc.update(y=1, d=1)
    elif family == FAM_KALMAN:
        dc.update(u=1, d=1)
    coop_ok = family in (FAM_FORWARD, FAM_RICCATI, FAM_KALMAN) and lib.lqg_coop_supported(C.byref(_dims_struct(dc)))
    if family == FAM_SIMULATE:
        coop_ok = True                              # k_coop_simulate: any (x, b, u, y)
    if family == FAM_ADJOINT:                       # the cooperative reverse-mode sweep (fp64; x, b <= 64; u, y, d <= 4)
        coop_ok = bool(lib.lqg_grad_supported(F64, C.byref(dm)))
    can_jit = shape_in_range(*key0) and os.path.exists(build.HIPCC) and _options.flag("JIT")
    if family == FAM_ADJOINT and dims["x"] + dims["b"] > build.ADJOINT_MAX_JOINT:
        can_jit = False                             # (on-demand libraries carry the gradient sweep up to x + b = 12)
    big = n_sys is not None and n_sys >= JIT_MIN_SYSTEMS
    names = "xbuyd"
    want = {k: dims[k] for k in _FAMILY_KEYS[family]}
    ddir = build.cache_dir(build.DIMS_DIR, "dims")
    for f in sorted(os.listdir(ddir)) if os.path.isdir(ddir) else ():
        if not (f.startswith("liblqg_hip_") and f.endswith(".so")):
            continue
        try:
            tup = tuple(int(v) for v in f[len("liblqg_hip_"):-3].split("_"))
        except ValueError:
            continue
        if len(tup) != 5 or any(tup[names.index(k)] != v for k, v in want.items()):
            continue
        if tup not in _dims_libs and build.stamped(os.path.join(ddir, f), build.source_hash()):
            _dims_libs[tup] = _bind(os.path.join(ddir, f))
            return _dims_libs[tup]
    if coop_ok and not (can_jit and big):
        return lib                                  # cooperative kernels: the main library decides per call (lqg_strategy)
    key = (dims["x"], dims["b"], dims["u"], dims["y"], dims["d"])
    if family != FAM_FORWARD and not (1 <= key[4] <= key[0]):
        key = key[:4] + (min(key[0], 2),)           # d is irrelevant to this family: the tracking models' width
    if key not in _dims_libs:
        if not shape_in_range(*key):
            raise LqgHipError(f"model shape (x,b,u,y,d)={key} is outside the dims the register-resident kernels serve "
                              f"(x, b <= {MAX_STATE}, x+b <= {MAX_JOINT}, u, y <= {MAX_IO}, d <= x); there is no CPU path")
        if not os.path.exists(build.HIPCC):
            raise LqgHipError(f"no kernels for model shape (x,b,u,y,d)={key} in {LIB_PATH} and no hipcc to compile "
                              "them: add the shape to lqg_amd/csrc/lqg_dims.def and rebuild")
        _dims_libs[key] = _bind(build.build_dims_library(*key))
    return _dims_libs[key]


def shape_available(x, b, u, y, d):
    """True when kernels for the shape exist (lane or cooperative) or can be compiled on demand."""
    lib = load()
    dm = Dims(x, b, u, y, d, 1, 1, 1, 1)
    if lib.lqg_dims_supported(F32, C.byref(dm)) or lib.lqg_coop_supported(C.byref(dm)):
        return True
    from lqg_amd import build
    return shape_in_range(x, b, u, y, d) and os.path.exists(build.HIPCC)


def check(rc, what):
    if rc == 0:
        return
    lib = load()
    msg = lib.lqg_last_error()
    msg = msg.decode() if msg else ""
    kind = "invalid argument" if rc < 0 else "hipError_t"
    raise LqgHipError(f"{what} failed: {kind} {rc}: {msg}")


# ---- view construction from (ptr, shape, element strides) ----------------------------------------------------

def mat_view(ptr, shape, strides, batched, has_time=True, vector=False):
    """Strided view of a spec field.

    shape/strides exclude nothing: [B?][T?][rows][cols?].  `batched` says whether a leading system axis
    is present; `has_time` whether a time axis is present (Qf, qf, Sigma0 have none)."""
    shape, strides = list(shape), list(strides)
    sb = st = 0
    if batched:
        sb = strides.pop(0) if shape[0] > 1 else (strides.pop(0) and 0)
        shape.pop(0)
    if has_time:
        st = strides.pop(0) if shape[0] > 1 else (strides.pop(0) and 0)
        shape.pop(0)
    if vector:
        return View(ptr, sb, st, strides[0], 0)
    return View(ptr, sb, st, strides[0], strides[1])


def traj_view(ptr, shape, strides, batched):
    """Strided view of a trajectory array [B?][N][T][d]."""
    strides = list(strides)
    sb = 0
    if batched:
        sb = strides[0] if shape[0] > 1 else 0
        strides = strides[1:]
    return Traj(ptr, sb, strides[0], strides[1], strides[2])
