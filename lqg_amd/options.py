"""Every switch that selects between equivalent routes of the path, in ONE place.

The C library (include/lqg_hip.h) reads no environment variable and keeps no option state: what selects between its
equivalent kernels travels with each problem in `lqg_problem.tuning` (ABI 3), filled here.  The Python host logic
(lqg_amd/plan.py and friends) asks `get(name)` at the moment it decides — never at import time, never cached — so a
process behaves the same whenever a switch was set.

Two ways to set a switch, both for developer A/B measurements and for the tests that pin every route against the default:
  * `lqg_amd.options.set(name, value)` / `with lqg_amd.options.override(name=value, ...):`  (programmatic, wins)
  * the environment variable `LQG_<NAME>`  (read per call)

name              default   meaning
----------------  --------  ----------------------------------------------------------------------------------------
SCAN              ""        time-parallel system sweeps: "" default rule (few systems, long horizon), "0" never, "1" wherever defined
SCAN_MAX_SYSTEMS  0         systems up to which the default rule takes the scans (0: plan.scan_max_systems(m))
SCAN_MIN_STEPS    0         horizon from which it does (0: plan.scan_min_steps(m))
SCAN_MAX_COND     1e7       cond((V V')[:d, :d]) above which the default rule keeps the sequential sweeps
SCAN_LANE         1         one-launch scans of 1x1 .. 3x3 windows (0: one launch per level)          -> tuning.scan_lane
SCAN_RT_WAVES     0         waves per window of k_scan_level_rt (0 = 16; 8)                           -> tuning.scan_rt_waves
SCAN_ORDER        ""        scans over windows of 25 .. 64: "" rule, "1" work-efficient levels always (Brent-Kung around a scan of the block totals), "2" plain Brent-Kung, "0" Hillis-Steele -> tuning.scan_order
COOP              ""        "" default rule, "1" cooperative kernels wherever supported, "0" never    -> tuning.coop
COOP_SPARSE       1         run-time sparsity lists of the cooperative sweeps                         -> tuning.coop_sparse
COOP_TRIAL_ROWS   1         row-parallel per-trial sweep of large joint dimensions                    -> tuning.coop_trial_rows
COOP_TRIAL_TPB    0         most trials per workgroup of that sweep (0: the rule of coop_trial(); a power of two <= 128)  -> tuning.coop_trial_tpb
COOP_TRIAL_WIDE   ""        that sweep on 1024-thread workgroups: "" rule (at 128 trials per workgroup), "1" always, "0" never -> tuning.coop_trial_wide
COOP_TRIAL_CHUNKS ""        its cut along time: "" rule, "0" / "1" one pass, k chunks                  -> tuning.coop_trial_chunks
COOP_ADJOINT      0         1: the cooperative reverse-mode sweep also for shapes with adjoint lane kernels -> tuning.coop_adjoint
ADJOINT_SP        1         reverse-mode sweep on the structure-specialised adjoint libraries (matrix adjoints once per system, trial sums; csrc/lqg_adjoint_sp.hpp); 0: the round-1 lane kernels (one (system, trial) pair per lane)
TRIAL_CHUNKS      ""        lane per-trial sweep cut along time: "" rule, "0" / "1" one pass, k chunks -> tuning.trial_chunks
TRIAL_CHUNK_WAVES / TRIAL_CHUNK_MAX_WAVES / TRIAL_CHUNK_TPL   0   parameters of that rule             -> tuning.trial_chunk_*
TRIAL_LDS         ""        lane per-trial sweep with the operator stream staged in LDS (k_trial_lds): "" rule (>= 768 trials per system, >= 256 systems), "1" always, "0" never -> tuning.trial_lds
FUSE_TRIALS_MAX   2048      (system, trial) pairs up to which a small multi-trial evaluation runs as fused pairs (0: never)
MIXED             1         fp32 problems on the operator stream run their system sweeps in fp64 (LQG_F32_SYS64)
MIXED_MIN_TRIALS  3         trials per system from which they do
HILO              1         the mixed mode keeps the rounding residual of the operator's Fj - I block (hi + lo operators for systems whose block reaches 2.0); 0: operators rounded once, no residual stream in the workspace -> tuning.hilo
F32_WIDE          1         ill-conditioned fp32 problems run over an fp64 image of specs and data (plan.F32_MAX_COND)
F32_MAX_COND      1e7       cond((V V')[:d, :d]) above which they do
X4_LAYOUT         0         1: merged fp32 components whose lane reads four floats per row are laid [T+1][B][trial][component] (one 16-byte load per step, k_forward_sp<X4>); measured, see DESIGN.md §7
TV_JIT_MIN_WORK   1048576   systems x steps from which a time-varying / affine model whose pattern library does not exist yet gets one compiled (~40 s of hipcc, once; cached on disk); below, the dense generic kernels serve it
NO_SPECIALIZE     0         1: generic dense kernels instead of the pattern libraries
NO_DECOUPLE       0         1: solve the joint problem even when it splits into independent components
NO_MERGE          0         1: do not solve identical components once
GRAPH             1         0: inference loops launch eagerly instead of replaying one hipGraph per evaluation
GRAPH_AFFINE      1         0: captured constructors are not collapsed into their measured affine map
SETUP_KERNEL      1         0: PointMassBoundedActor is discretised by torch.linalg instead of lqg_point_mass_setup
JIT               1         0: never compile an auxiliary lane-kernel library for an unlisted shape on demand

Build / location variables (read once where the files are located, not behaviour switches): LQG_HIP_LIB, LQG_PAT_DIR,
LQG_DIMS_DIR, LQG_SP_FLAGS, HIPCC.
"""
import contextlib
import os

DEFAULTS = {
    "SCAN": "", "SCAN_MAX_SYSTEMS": 0, "SCAN_MIN_STEPS": 0, "SCAN_MAX_COND": 1e7, "SCAN_LANE": 1, "SCAN_RT_WAVES": 0, "SCAN_ORDER": "",
    "COOP": "", "COOP_SPARSE": 1, "COOP_TRIAL_ROWS": 1, "COOP_TRIAL_TPB": 0, "COOP_TRIAL_WIDE": "", "COOP_TRIAL_CHUNKS": "", "COOP_ADJOINT": 0, "ADJOINT_SP": 1, "TRIAL_CHUNKS": "", "TRIAL_LDS": "", "TRIAL_CHUNK_WAVES": 0,
    "TRIAL_CHUNK_MAX_WAVES": 0, "TRIAL_CHUNK_TPL": 0, "FUSE_TRIALS_MAX": 2048, "MIXED": 1, "MIXED_MIN_TRIALS": 3, "HILO": 1,
    "F32_WIDE": 1, "F32_MAX_COND": 1e7, "X4_LAYOUT": 0, "TV_JIT_MIN_WORK": 1 << 20, "NO_SPECIALIZE": 0, "NO_DECOUPLE": 0, "NO_MERGE": 0, "GRAPH": 1, "GRAPH_AFFINE": 1,
    "SETUP_KERNEL": 1, "JIT": 1,
}
_overrides = {}


def get(name):
    """Current value of a switch, typed like its default ("" defaults stay strings: "" means "the default rule")."""
    default = DEFAULTS[name]
    if name in _overrides:
        raw = _overrides[name]
    else:
        raw = os.environ.get("LQG_" + name)
        if raw is None:
            return default
    if isinstance(default, str):
        return str(raw)
    if isinstance(default, float):
        return float(raw)
    return int(raw)


def flag(name):
    return get(name) != 0


def set(name, value):          # noqa: A001  (the module is the namespace: lqg_amd.options.set)
    """Programmatic override (wins over the environment); value None removes it."""
    if name not in DEFAULTS:
        raise KeyError(f"unknown option {name!r}; known: {sorted(DEFAULTS)}")
    if value is None:
        _overrides.pop(name, None)
    else:
        _overrides[name] = value


@contextlib.contextmanager
def override(**kw):
    """`with options.override(SCAN="0", TRIAL_CHUNKS="0"): ...` — restored on exit."""
    missing = object()
    old = {k: _overrides.get(k, missing) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            if v is missing:
                _overrides.pop(k, None)
            else:
                _overrides[k] = v


def _tri(name):
    """"" -> 0 (default rule), "0" -> -1 (off), "1" -> 1 (on / forced)."""
    v = get(name)
    if v == "":
        return 0
    return 1 if int(v) > 0 else -1


def _chunks(name):
    """"" -> 0 (default rule), "0" / "1" -> -1 (one pass), "k" -> k chunks."""
    v = get(name)
    if v == "":
        return 0
    k = int(v)
    return k if k > 1 else -1


def fill_tuning(t):
    """Write the switches the C library acts on into a `lqg_tuning` (lqg_amd/_abi.py: Tuning)."""
    t.coop = _tri("COOP")
    t.trial_chunks = _chunks("TRIAL_CHUNKS")
    t.trial_chunk_waves = get("TRIAL_CHUNK_WAVES")
    t.trial_chunk_max_waves = get("TRIAL_CHUNK_MAX_WAVES")
    t.trial_chunk_tpl = get("TRIAL_CHUNK_TPL")
    t.coop_trial_chunks = _chunks("COOP_TRIAL_CHUNKS")
    t.coop_trial_rows = 0 if flag("COOP_TRIAL_ROWS") else -1
    t.coop_sparse = 0 if flag("COOP_SPARSE") else -1
    t.scan_lane = 0 if flag("SCAN_LANE") else -1
    t.scan_rt_waves = get("SCAN_RT_WAVES")
    t.coop_adjoint = 1 if flag("COOP_ADJOINT") else 0
    so = get("SCAN_ORDER")
    t.scan_order = 0 if so == "" else (-1 if int(so) == 0 else int(so))
    t.coop_trial_tpb = get("COOP_TRIAL_TPB")
    t.coop_trial_wide = _tri("COOP_TRIAL_WIDE")
    v = get("TRIAL_LDS")
    t.trial_lds = 0 if v == "" else (-1 if int(v) == 0 else int(v))
    t.hilo = 0 if flag("HILO") else -1
    return t
