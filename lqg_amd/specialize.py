"""Structure specialisation of the hot path: derive the sparsity pattern of a model family from its spec tensors,
generate + compile (hipcc, gfx950) a library whose kernels carry the pattern as compile-time masks
(csrc/lqg_kernels_sp.hpp), cache it in-tree under csrc/pat/, and hand it to the launch glue.

Why: every model of the reference's zoo has A = I + a few couplings, F = selection rows, diagonal V and W, and
hoisted products that cancel exactly (Fd Bd - Fa Ba == 0).  Dense kernels spend most of their multiply-adds and most
of their registers on those zeros.  XLA cannot exploit this in the reference either (dense dots); a generated
kernel can.  The generic dense library remains the path for everything that is not eligible (several trials per
system, time-varying specs, affine terms) and whenever no compiler is available for a new pattern.

The masks are derived from DATA: an entry is structurally non-zero if it is non-zero for ANY system (candidate) and
time step of the spec.  For the model zoo the mask of a class is taken from a probe instance with random positive
parameters (so that a parameter that happens to be zero in one call can never narrow a cached mask).
"""
import ctypes as C
import hashlib
import os
import shutil
import subprocess

import numpy as np
import torch

from lqg_amd import _abi
from lqg_amd import build as _build

PAT_DIR = os.environ.get("LQG_PAT_DIR") or os.path.join(_build.CSRC, "pat")
_FIELDS = ("Aa", "Ba", "Fa", "VVa", "WWa", "Q", "Rr", "Ad", "AdmI", "Bd", "Fd", "N1", "WWd", "FAa", "FAd", "DB", "N2", "N3",
           "Va", "Wa", "Vd", "Wd")            # (the last four: the stored noise factors themselves, first n columns)
_libs = {}
_class_patterns = {}


def _any_nz(t):
    """[..., r, c] tensor -> bool [r, c]: non-zero for any leading index."""
    t = t.detach()
    nz = t != 0
    while nz.dim() > 2:
        nz = nz.any(dim=0)
    return nz.cpu().numpy()


def _square(mask, n):
    """[n, nv] mask of a stored noise factor -> [n, n]: its first n columns (absent columns False; the kernels load columns
    beyond n in full)."""
    out = np.zeros((n, n), dtype=bool)
    c = min(n, mask.shape[1])
    out[:, :c] = mask[:, :c]
    return out


def _first(t):
    """spec field [(B,) T, r, c]: one time slice when the field is a time-invariant (stride-0) view, else the whole
    stack — the masks are unions over every leading index, time included."""
    return t.select(-3, 0) if (t.stride(-3) == 0 or t.shape[-3] == 1) else t


def pattern_of(system, d, grad_full=False):
    """Masks of every per-system constant the specialised kernels load or hoist (dict name -> bool ndarray).

    grad_full=True (differentiable evaluation): every entry of a field that REQUIRES GRAD counts as structurally
    non-zero — an entry that is zero at the current parameter values may still carry a non-zero derivative
    (A = eye(4) as a leaf: d ll / d A[0, 2] != 0), so the structure must not be read off the values."""
    a, dy = system.actor, system.dynamics

    def val(spec, f):
        t = _first(getattr(spec, f)).detach().double()
        if grad_full and getattr(spec, f).requires_grad:
            t = torch.ones_like(t)
        return t

    Aa, Ba, Fa, Va, Wa = (val(a, f) for f in ("A", "B", "F", "V", "W"))
    Q, Rr = val(a, "Q"), val(a, "R")
    if grad_full and a.Qf.requires_grad:
        Q = torch.ones_like(Q)
    Ad, Bd, Fd, Vd, Wd = (val(dy, f) for f in ("A", "B", "F", "V", "W"))
    T_ = lambda m: m.transpose(-1, -2)
    N1 = Vd @ T_(Vd)
    masks = dict(Aa=Aa, Ba=Ba, Fa=Fa, VVa=Va @ T_(Va), WWa=Wa @ T_(Wa), Q=0.5 * (Q + T_(Q)), Rr=0.5 * (Rr + T_(Rr)),
                 Ad=Ad, AdmI=Ad - torch.eye(Ad.shape[-1], dtype=Ad.dtype, device=Ad.device), Bd=Bd, Fd=Fd, N1=N1, WWd=Wd @ T_(Wd), FAa=Fa @ Aa, FAd=Fd @ Ad, DB=Fd @ Bd - Fa @ Ba,
                 N2=Fd @ N1, N3=Fd @ N1 @ T_(Fd) + Wd @ T_(Wd))
    out = {k: _any_nz(v) for k, v in masks.items()}
    if grad_full:
        # The masks of the hoisted PRODUCTS must not be read off values here: Fd Bd - Fa Ba cancels numerically for a shared
        # spec (and for matrices of ones), A - I has a zero diagonal for A = ones — yet d ll / d Bd, d ll / d A_ii are not
        # zero.  Boolean algebra of the factors' masks instead (conservative, as pattern_of_time_varying).
        bm = lambda p, q: (p.astype(np.int64) @ q.astype(np.int64)) > 0
        o = out
        o["FAa"], o["FAd"] = bm(o["Fa"], o["Aa"]), bm(o["Fd"], o["Ad"])
        o["DB"] = bm(o["Fd"], o["Bd"]) | bm(o["Fa"], o["Ba"])
        o["N2"] = bm(o["Fd"], o["N1"])
        o["N3"] = bm(bm(o["Fd"], o["N1"]), o["Fd"].T) | o["WWd"]
        if dy.A.requires_grad:
            o["AdmI"] = o["Ad"] | np.eye(o["Ad"].shape[0], dtype=bool)
    for k in ("VVa", "WWa", "Q", "Rr", "N1", "WWd", "N3"):      # symmetric quantities: keep the mask symmetric
        out[k] = out[k] | out[k].T
    for k, v in (("Va", Va), ("Wa", Wa), ("Vd", Vd), ("Wd", Wd)):
        out[k] = _square(_any_nz(v), v.shape[-2])
    dims = dict(x=Ad.shape[-1], b=Aa.shape[-1], u=Ba.shape[-1], y=Fa.shape[-2], d=int(d))
    return dims, out


def pattern_of_time_varying(system, d):
    """Masks for specs that vary in TIME (and over the systems) under one sparsity pattern: the raw fields' masks are the
    union over every system and step (a reduction, no temporary of the fields' size — 2^17 systems x 500 steps are 65 M
    matrices per field); the masks of the hoisted products follow from those by the boolean algebra of the products, i.e.
    they are CONSERVATIVE (a numerical cancellation such as Fd Bd - Fa Ba = 0 of the tracking models is not assumed: it does
    not survive independent variation of the entries in time)."""
    a, dy = system.actor, system.dynamics

    def nz(t):
        lead = tuple(range(t.dim() - 2))
        return (torch.count_nonzero(t.detach(), dim=lead) != 0).cpu().numpy() if lead else (t.detach() != 0).cpu().numpy()

    def diag_differs_from_one(t):
        dg = torch.diagonal(t.detach(), dim1=-2, dim2=-1)
        lead = tuple(range(dg.dim() - 1))
        return (torch.count_nonzero(dg - 1, dim=lead) != 0).cpu().numpy() if lead else (dg != 1).cpu().numpy()

    bm = lambda p, q: (p.astype(np.int64) @ q.astype(np.int64)) > 0           # boolean matrix product
    Aa, Ba, Fa, Va, Wa = (nz(getattr(a, f)) for f in ("A", "B", "F", "V", "W"))
    Q, Rr = nz(a.Q) | nz(a.Qf), nz(a.R)
    Ad, Bd, Fd, Vd, Wd = (nz(getattr(dy, f)) for f in ("A", "B", "F", "V", "W"))
    N1 = bm(Vd, Vd.T)
    AdmI = Ad.copy()
    np.fill_diagonal(AdmI, diag_differs_from_one(dy.A))
    masks = dict(Aa=Aa, Ba=Ba, Fa=Fa, VVa=bm(Va, Va.T), WWa=bm(Wa, Wa.T), Q=Q | Q.T, Rr=Rr | Rr.T, Ad=Ad, AdmI=AdmI, Bd=Bd,
                 Fd=Fd, N1=N1, WWd=bm(Wd, Wd.T), FAa=bm(Fa, Aa), FAd=bm(Fd, Ad), DB=bm(Fd, Bd) | bm(Fa, Ba), N2=bm(Fd, N1),
                 N3=bm(bm(Fd, N1), Fd.T) | bm(Wd, Wd.T))
    for k in ("VVa", "WWa", "N1", "WWd", "N3"):
        masks[k] = masks[k] | masks[k].T
    for k, v in (("Va", Va), ("Wa", Wa), ("Vd", Vd), ("Wd", Wd)):
        masks[k] = _square(v, v.shape[0])
    dims = dict(x=Ad.shape[-1], b=Aa.shape[-1], u=Ba.shape[-1], y=Fa.shape[-2], d=int(d))
    return dims, masks


def _varies_in_time(system):
    """True when a spec stack of the system has more than one time slice that is not a stride-0 view."""
    for spec in (system.actor, system.dynamics):
        for f in ("A", "B", "F", "V", "W", "Q", "R"):
            t = getattr(spec, f)
            if t.dim() >= 3 and t.shape[-3] > 1 and t.stride(-3) != 0:
                return True
    return False


def pattern_key(dims, masks):
    h = hashlib.sha1(repr(sorted(dims.items())).encode())
    for k in _FIELDS:
        h.update(k.encode())
        h.update(np.packbits(masks[k].astype(np.uint8)).tobytes())
    return h.hexdigest()[:16]


def density(masks):
    nz = sum(int(masks[k].sum()) for k in _FIELDS)
    return nz / sum(masks[k].size for k in _FIELDS)


def generate_source(key, dims, masks):
    def lit(m):
        return "{{" + ", ".join("true" if v else "false" for v in m.reshape(-1)) + "}}"
    lines = ["// GENERATED by lqg_amd/specialize.py — structure-specialised hot path, pattern " + key,
             '#include "lqg_sp_entry.hpp"', "", "namespace {", "struct Pat {"]
    for k in _FIELDS:
        r, c = masks[k].shape
        lines.append(f"  static constexpr lqg::Mask<{r}, {c}> {k}{lit(masks[k])};")
    lines += ["};", "}  // namespace", "",
              'extern "C" int lqg_log_likelihood_sp(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,',
              "                                     void* workspace, size_t workspace_bytes, void* stream) {",
              f"  return lqg::host::log_likelihood_sp<Pat, {dims['x']}, {dims['b']}, {dims['u']}, {dims['y']}, {dims['d']}>(",
              "      p, x, ll, ll_sb, ll_sn, workspace, workspace_bytes, stream);", "}",
              'extern "C" int lqg_trial_sweep_sp(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,',
              "                                  const void* ops, void* stream) {",
              f"  return lqg::host::trial_sweep_entry<Pat, {dims['x']}, {dims['b']}, {dims['u']}, {dims['y']}, {dims['d']}>(",
              "      p, x, ll, ll_sb, ll_sn, ops, stream);", "}",
              'extern "C" int lqg_solve_materialised_sp(const lqg_problem* p, lqg_traj x, lqg_view L, lqg_view l, lqg_view H, lqg_view K,',
              "                                         lqg_traj mu, lqg_view Sigma, void* ll, int64_t ll_sb, void* workspace,",
              "                                         size_t workspace_bytes, void* stream) {",
              f"  return lqg::host::solve_materialised_sp<Pat, {dims['x']}, {dims['b']}, {dims['u']}, {dims['y']}, {dims['d']}>(",
              "      p, x, L, l, H, K, mu, Sigma, ll, ll_sb, workspace, workspace_bytes, stream);", "}",
              f'extern "C" const char* lqg_sp_pattern_key(void) {{ return "{key}"; }}', ""]
    return "\n".join(lines)


# Extra flags of the pattern libraries.  The specialised kernels already skip structurally-zero terms through their masks
# ("0 * x never issued"); the moment recursion's state Sigma is carried as a dense array whose structural zeros (the
# observed block of the tracking models is diagonal, the target row of F2 is empty, ...) are written as literal 0 each
# step — with these three flags the compiler may fold `0 * x` and `x + (-0)` through the 8-step unrolled chunk, which is
# the same rule applied to the zeros the masks do not type: headline forward kernel 3.32 -> 3.18 ms (+4 % solves/s),
# bitwise identical results on finite data (profiles/r03_exp_flags.txt).  They do NOT enable reassociation or
# reciprocal rewriting; NaN / inf produced by a singular block still propagate through every non-zero coefficient
# (tests/test_gpu_parity.py::test_specialised_path_on_hand_built_system_and_nan_semantics).
SP_EXTRA_FLAGS = ["-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros"]


def _headers_hash():
    h = hashlib.sha1(" ".join(SP_EXTRA_FLAGS).encode())
    for f in ("lqg_small.hpp", "lqg_rng.hpp", "lqg_sparse.hpp", "lqg_kernels.hpp", "lqg_kernels_sp.hpp", "lqg_launch.hpp", "lqg_trial_chunk.hpp",
              "lqg_sp_entry.hpp", "../../include/lqg_hip.h"):
        h.update(open(os.path.join(_build.CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def compile_pattern(key, dims, masks, verbose=False):
    """Generate + compile <pattern dir>/pat_<key>.so (returns its path, or None when no hipcc is available).
    Safe under concurrent callers (ranks of one node asking for the same pattern): inter-process lock, compile to a
    temporary name, atomic rename, stamp last; a read-only tree falls back to a per-user cache directory."""
    pdir = _build.cache_dir(PAT_DIR, "pat")
    so = os.path.join(pdir, f"pat_{key}.so")
    hh = _headers_hash()
    if _build.stamped(so, hh):
        return so
    hipcc = _build.HIPCC if os.path.exists(_build.HIPCC) else shutil.which("hipcc")
    if not hipcc:
        return None
    with _build.locked(so):
        if _build.stamped(so, hh):                 # built by another process while this one waited for the lock
            return so
        src = os.path.join(pdir, f"pat_{key}.hip")
        _build.atomic_write(src, generate_source(key, dims, masks))
        flags = [fl for fl in _build.FLAGS if not fl.startswith("-std=")] + ["-std=c++20"] + SP_EXTRA_FLAGS
        flags += ["-I", _build.CSRC] + os.environ.get("LQG_SP_FLAGS", "").split()   # LQG_SP_FLAGS: developer A/B builds
        tmp = f"{so}.tmp.{os.getpid()}"
        cmd = [hipcc] + flags + ["-shared", src, "-o", tmp]
        if verbose:
            print(f"[lqg_amd.specialize] compiling pattern {key} dims={dims} density={density(masks):.2f}", flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        os.replace(tmp, so)
        _build.atomic_write(so + ".stamp", hh)
    return so


def pattern_on_disk(key):
    """True when the stamped library of this pattern exists already (no compile would be needed)."""
    if key in _libs:
        return _libs[key] is not None
    so = os.path.join(_build.cache_dir(PAT_DIR, "pat"), f"pat_{key}.so")
    return _build.stamped(so, _headers_hash())


def load_pattern(key, dims, masks, verbose=False):
    """ctypes handle of the specialised library for this pattern (compiled on first use), or None.

    Specialisation is an optimisation layer: any failure to produce the library (no compiler, read-only tree, compile
    error) is reported once and answered with None — the caller then runs the generic dense HIP kernels."""
    if key in _libs:
        return _libs[key]
    lib = None
    try:
        so = compile_pattern(key, dims, masks, verbose=verbose)
        if so is not None:
            lib = C.CDLL(so)
            lib.lqg_log_likelihood_sp.argtypes = [C.POINTER(_abi.Problem), _abi.Traj, C.c_void_p, C.c_int64, C.c_int64,
                                                  C.c_void_p, C.c_size_t, C.c_void_p]
            lib.lqg_log_likelihood_sp.restype = C.c_int
            lib.lqg_trial_sweep_sp.argtypes = [C.POINTER(_abi.Problem), _abi.Traj, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_void_p, C.c_void_p]
            lib.lqg_trial_sweep_sp.restype = C.c_int
            if hasattr(lib, "lqg_solve_materialised_sp"):
                lib.lqg_solve_materialised_sp.argtypes = ([C.POINTER(_abi.Problem), _abi.Traj] + [_abi.View] * 4 + [_abi.Traj, _abi.View]
                                                          + [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p])
                lib.lqg_solve_materialised_sp.restype = C.c_int
    except (OSError, RuntimeError) as e:
        import warnings
        warnings.warn(f"lqg_amd: no specialised library for pattern {key} ({e!s:.200}); using the generic dense kernels")
        lib = None
    _libs[key] = lib
    return lib


# ---------------------------------------------------------------- adjoint libraries (round 5: csrc/lqg_adjoint_sp.hpp)
_adj_libs = {}
ADJ_EXTRA_FLAGS = []           # (no value-changing flags: the bars are compared with the restatement entry by entry)


LIVE_FIELDS = ("Aa", "Ba", "Fa", "Va", "Wa", "Q", "R", "Ad", "Bd", "Fd", "Vd", "Wd")
ALL_LIVE = {k: True for k in LIVE_FIELDS}


def adjoint_key(key, live):
    """Key of an adjoint library: the pattern's key + which spec fields any parameter moves (bars of the others are not formed)."""
    bits = "".join("1" if live[k] else "0" for k in LIVE_FIELDS)
    return key if all(live[k] for k in LIVE_FIELDS) else f"{key}_{int(bits, 2):03x}"


def generate_adjoint_sources(key, dims, masks, live=None):
    """{file name: text} of one adjoint library: the pattern as a NAMED struct in a header, four translation units that each
    instantiate the sweep for one (dtype, explicit-Sigma0) combination — compiled in parallel: the m = 8 point-mass kernels
    took 10+ minutes as one unit — and the entry points."""
    live = live or ALL_LIVE

    def lit(m):
        return "{{" + ", ".join("true" if v else "false" for v in m.reshape(-1)) + "}}"
    d = dims
    pat = "Pat_" + key
    tpl = f"lqg::{pat}, {d['x']}, {d['b']}, {d['u']}, {d['y']}, {d['d']}"
    hdr = ["// GENERATED by lqg_amd/specialize.py — sparsity pattern " + key, "#pragma once", '#include "lqg_sparse.hpp"',
           "namespace lqg {", f"struct {pat} {{"]
    for k in _FIELDS:
        r, c = masks[k].shape
        hdr.append(f"  static constexpr Mask<{r}, {c}> {k}{lit(masks[k])};")
    for k in LIVE_FIELDS:
        hdr.append(f"  static constexpr bool live_{k} = {'true' if live[k] else 'false'};")
    hdr += ["};", "}  // namespace lqg", ""]
    args = ("const lqg_problem*, lqg_traj, const void*, long, long, void*, long, long, void*, long, void*, size_t, int, hipStream_t")
    files = {f"padj_{key}_pat.hpp": "\n".join(hdr)}
    combos = [(r, dp) for r in ("float", "double") for dp in ("false", "true")]
    for r, dp in combos:
        files[f"padj_{key}_{r[0]}{'1' if dp == 'true' else '0'}.hip"] = "\n".join([
            "// GENERATED by lqg_amd/specialize.py — structure-specialised reverse-mode sweep, pattern " + key + f" ({r}, Sigma0 given: {dp})",
            f'#include "padj_{key}_pat.hpp"', '#include "lqg_adjoint_sp_entry.hpp"',
            f"template int lqg::host::run_asp<{r}, {tpl}, {dp}>({args});", ""])
    main = ["// GENERATED by lqg_amd/specialize.py — structure-specialised reverse-mode sweep, pattern " + key + " (entry points)",
            f'#include "padj_{key}_pat.hpp"', '#include "lqg_adjoint_sp_entry.hpp"']
    for r, dp in combos:
        main.append(f"extern template int lqg::host::run_asp<{r}, {tpl}, {dp}>({args});")
    main += ['extern "C" int lqg_log_likelihood_grad_sp(const lqg_problem* p, lqg_traj x, const void* g, int64_t g_sb, int64_t g_sn,',
             "                                          void* ll, int64_t ll_sb, int64_t ll_sn, void* grad, int64_t ld, void* workspace,",
             "                                          size_t workspace_bytes, int32_t phases, void* stream) {",
             f"  return lqg::host::log_likelihood_grad_sp<{tpl}>(p, x, g, g_sb, g_sn, ll, ll_sb, ll_sn, grad, ld, workspace,",
             "                                                   workspace_bytes, phases, stream);", "}",
             'extern "C" size_t lqg_grad_workspace_bytes_sp(const lqg_problem* p) {',
             f"  return lqg::host::grad_workspace_bytes_sp<{tpl}>(p);", "}",
             f'extern "C" const char* lqg_sp_pattern_key(void) {{ return "{key}"; }}', ""]
    files[f"padj_{key}.hip"] = "\n".join(main)
    return files


def _adj_headers_hash():
    h = hashlib.sha1(" ".join(ADJ_EXTRA_FLAGS).encode())
    for f in ("lqg_small.hpp", "lqg_sparse.hpp", "lqg_kernels.hpp", "lqg_kernels_sp.hpp", "lqg_launch.hpp", "lqg_trial_chunk.hpp",
              "lqg_adjoint.hpp", "lqg_adjoint_sp.hpp", "lqg_adjoint_trial_sp.hpp", "lqg_adjoint_sp_entry.hpp", "../../include/lqg_hip.h"):
        h.update(open(os.path.join(_build.CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def compile_adjoint_pattern(key, dims, masks, verbose=False, live=None):
    """Generate + compile <pattern dir>/padj_<key>.so — the reverse-mode twin of pat_<key>.so (same masks, same locking
    discipline as compile_pattern).  live: which spec fields a parameter moves (LIVE_FIELDS; None = all)."""
    live = live or ALL_LIVE
    key = adjoint_key(key, live)
    pdir = _build.cache_dir(PAT_DIR, "pat")
    so = os.path.join(pdir, f"padj_{key}.so")
    hh = _adj_headers_hash()
    if _build.stamped(so, hh):
        return so
    hipcc = _build.HIPCC if os.path.exists(_build.HIPCC) else shutil.which("hipcc")
    if not hipcc:
        return None
    with _build.locked(so):
        if _build.stamped(so, hh):
            return so
        import concurrent.futures as cf
        files = generate_adjoint_sources(key, dims, masks, live)
        for name, text in files.items():
            _build.atomic_write(os.path.join(pdir, name), text)
        flags = [fl for fl in _build.FLAGS if not fl.startswith("-std=")] + ["-std=c++20"] + ADJ_EXTRA_FLAGS
        flags += ["-I", _build.CSRC, "-I", pdir] + os.environ.get("LQG_ADJ_FLAGS", "").split()
        if verbose:
            print(f"[lqg_amd.specialize] compiling adjoint pattern {key} dims={dims} density={density(masks):.2f}", flush=True)
        units = [n for n in files if n.endswith(".hip")]
        objs = [os.path.join(pdir, n[:-4] + f".{os.getpid()}.o") for n in units]

        def cc(pair):
            return subprocess.run([hipcc] + flags + ["-c", os.path.join(pdir, pair[0]), "-o", pair[1]], capture_output=True, text=True)
        try:
            with cf.ThreadPoolExecutor(len(units)) as ex:
                for r, n in zip(ex.map(cc, zip(units, objs)), units):
                    if r.returncode != 0:
                        raise RuntimeError(f"hipcc failed on {os.path.join(pdir, n)}:\n{r.stderr[-6000:]}")
            tmp = f"{so}.tmp.{os.getpid()}"
            r = subprocess.run([hipcc, "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", tmp] + objs, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link of {so} failed:\n{r.stderr[-4000:]}")
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
        os.replace(tmp, so)
        _build.atomic_write(so + ".stamp", hh)
    return so


def load_adjoint_pattern(key, dims, masks, verbose=False, live=None):
    """ctypes handle of the adjoint library of this pattern (compiled on first use), or None (the caller then runs the
    round-1 lane kernels of the main library)."""
    live = live or ALL_LIVE
    base_key, key = key, adjoint_key(key, live)
    if key in _adj_libs:
        return _adj_libs[key]
    lib = None
    try:
        so = compile_adjoint_pattern(base_key, dims, masks, verbose=verbose, live=live)
        if so is not None:
            lib = C.CDLL(so)
            lib.lqg_log_likelihood_grad_sp.argtypes = [C.POINTER(_abi.Problem), _abi.Traj, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                       C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                                       C.c_int32, C.c_void_p]
            lib.lqg_log_likelihood_grad_sp.restype = C.c_int
            lib.lqg_grad_workspace_bytes_sp.argtypes = [C.POINTER(_abi.Problem)]
            lib.lqg_grad_workspace_bytes_sp.restype = C.c_size_t
    except (OSError, RuntimeError) as e:
        import warnings
        warnings.warn(f"lqg_amd: no adjoint library for pattern {key} ({e!s:.300}); using the generic adjoint kernels")
        lib = None
    _adj_libs[key] = lib
    return lib


# ---------------------------------------------------------------- model-zoo patterns (probe instances)
def class_pattern(cls, d, **structure_kw):
    """Pattern of a model-zoo class from a CPU probe instance with random positive parameters."""
    from lqg_amd.infer.models import get_model_params
    ck = (cls.__name__, d, tuple(sorted(structure_kw.items())))
    if ck not in _class_patterns:
        g = torch.Generator()
        g.manual_seed(12345)
        params = {k: float(0.3 + torch.rand((), generator=g)) for k in get_model_params(cls)}
        probe = cls(T=2, device="cpu", dtype=torch.float64, **params, **structure_kw)
        dims, masks = pattern_of(probe, d)
        _class_patterns[ck] = (dims, masks, pattern_key(dims, masks))
    return _class_patterns[ck]


def spec_versions(system):
    """In-place modification counters of every spec tensor: part of the cache keys of everything derived from the VALUES
    of a spec (patterns, decoupling), so that `spec.A[..., 0, 2] = 0.3` on a materialised tensor voids them."""
    return tuple(getattr(sp, f)._version for sp in (system.actor, system.dynamics) for f in sp._fields)


def zoo_structure(system):
    """The structure keywords of an EXACT model-zoo class instance (what, besides the class, fixes its sparsity), else None."""
    import lqg_amd
    zoo = (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor,
           lqg_amd.SubjectiveActor, lqg_amd.PointMassBoundedActor)
    return getattr(system, "_zoo_structure", None) if type(system) in zoo else None


_zoo_component_patterns = {}


def zoo_component_pattern(hint):
    """Pattern of component i of a decoupling zoo class, from a CPU probe instance with random positive parameters (as
    class_pattern): hint = (class, structure items, d of the joint data, i) set by decouple.plan."""
    cls, zs, d, i = hint
    ck = (cls, zs, d)
    if ck not in _zoo_component_patterns:
        from lqg_amd import decouple
        from lqg_amd.infer.models import get_model_params
        g = torch.Generator()
        g.manual_seed(12345)
        params = {k: float(0.3 + torch.rand((), generator=g)) for k in get_model_params(cls)}
        probe = cls(T=2, device="cpu", dtype=torch.float64, **params, **dict(zs))
        parts = decouple.plan(probe, d) or []
        out = []
        for sub, cols, _ in parts:
            sub.__dict__.pop("_lqg_zoo_component", None)            # (the probe's own components are read from their values)
            dims, masks = pattern_of(sub, len(cols))
            out.append((dims, masks, pattern_key(dims, masks)))
        _zoo_component_patterns[ck] = out
    pats = _zoo_component_patterns[ck]
    return pats[i] if i < len(pats) else None


def system_pattern(system, d, grad_full=False):
    """(dims, masks, key) for a System: class-level for the model zoo (cached), data-derived otherwise (cached on
    the instance, keyed by the tensors' in-place version counters)."""
    hint = getattr(system, "_lqg_zoo_component", None)
    if hint is not None and not grad_full:
        pat = zoo_component_pattern(hint)
        if pat is not None and pat[0]["d"] == d:
            return pat
    cache = system.__dict__.setdefault("_lqg_patterns", {})
    d_in = d
    d = (d_in, grad_full, spec_versions(system))
    if d not in cache:
        import lqg_amd
        cls = type(system)
        zoo = (lqg_amd.BoundedActor, lqg_amd.OptimalActor, lqg_amd.RelativeObservationBoundedActor,
               lqg_amd.SubjectiveActor, lqg_amd.PointMassBoundedActor)
        zoo_dim = getattr(system, "_zoo_structure", None) if cls in zoo else None   # exact classes only
        if zoo_dim is not None:
            cache[d] = class_pattern(cls, d_in, **zoo_dim)
        elif not grad_full and _varies_in_time(system):
            # stacks that move in time: union masks by reduction (pattern_of would take fp64 images of every stack and multiply
            # them out — 2^17 systems x 500 steps: 250 GB of temporaries, round 6)
            dims, masks = pattern_of_time_varying(system, d_in)
            cache[d] = (dims, masks, pattern_key(dims, masks))
        else:
            dims, masks = pattern_of(system, d_in, grad_full=grad_full)
            cache[d] = (dims, masks, pattern_key(dims, masks))
    return cache[d]


_zoo_live = {}


def _live_between(sa, sb):
    """Which spec fields differ between two systems built by the same constructor from different parameters."""
    pairs = dict(Aa=("actor", "A"), Ba=("actor", "B"), Fa=("actor", "F"), Va=("actor", "V"), Wa=("actor", "W"), R=("actor", "R"),
                 Ad=("dynamics", "A"), Bd=("dynamics", "B"), Fd=("dynamics", "F"), Vd=("dynamics", "V"), Wd=("dynamics", "W"))
    diff = lambda which, f: not torch.equal(getattr(getattr(sa, which), f), getattr(getattr(sb, which), f))
    live = {k: diff(*v) for k, v in pairs.items()}
    live["Q"] = diff("actor", "Q") or diff("actor", "Qf")
    return live


def zoo_live(cls, zs, d, component=None):
    """LIVE_FIELDS of a model-zoo class (or of component `component` of a decoupling one): the spec fields that ANY constructor
    parameter moves, from two CPU probe instances with different random positive parameters."""
    ck = (cls, zs, d)
    if ck not in _zoo_live:
        from lqg_amd import decouple
        from lqg_amd.infer.models import get_model_params
        probes = []
        for seed in (12345, 54321):
            g = torch.Generator()
            g.manual_seed(seed)
            params = {k: float(0.3 + torch.rand((), generator=g)) for k in get_model_params(cls)}
            probes.append(cls(T=2, device="cpu", dtype=torch.float64, **params, **dict(zs)))
        whole = _live_between(*probes)
        parts = [decouple.plan(p, d) or [] for p in probes]
        comps = [_live_between(pa[0], pb[0]) for pa, pb in zip(*parts)] if len(parts[0]) == len(parts[1]) else []
        _zoo_live[ck] = (whole, comps)
    whole, comps = _zoo_live[ck]
    if component is None:
        return whole
    return comps[component] if component < len(comps) else dict(ALL_LIVE)


def adjoint_pattern(system, d):
    """(dims, masks, key, live) for the reverse-mode sweep of a System: the constructor's structure for the model zoo and its
    decoupled components (a structural zero of a zoo constructor is a constant: its adjoint is never needed; `live` = the fields
    its parameters move), else the pattern with every field that requires grad counted as FULL (pattern_of(grad_full=True)) and
    live = the fields that require grad."""
    a, dy = system.actor, system.dynamics
    rg = lambda t: bool(getattr(t, "requires_grad", False))
    live = dict(Aa=rg(a.A), Ba=rg(a.B), Fa=rg(a.F), Va=rg(a.V), Wa=rg(a.W), Q=rg(a.Q) or rg(a.Qf), R=rg(a.R),
                Ad=rg(dy.A), Bd=rg(dy.B), Fd=rg(dy.F), Vd=rg(dy.V), Wd=rg(dy.W))
    # zoo classes: the class-level set (ONE library per class, prebuilt) united with what this instance differentiates — a
    # constructor argument outside get_model_params (dt, process_noise) that requires grad makes its fields live as well
    union = lambda cl: {k: bool(cl[k] or live[k]) for k in LIVE_FIELDS}
    hint = getattr(system, "_lqg_zoo_component", None) or getattr(system, "_lqg_zoo_component_grad", None)
    if hint is not None:
        pat = zoo_component_pattern(hint)
        if pat is not None and pat[0]["d"] == d:
            return pat + (union(zoo_live(hint[0], hint[1], hint[2], component=hint[3])),)
    zs = zoo_structure(system)
    pat = system_pattern(system, d, grad_full=True)
    if zs is not None:
        return pat + (union(zoo_live(type(system), tuple(sorted(zs.items())), d)),)
    return pat + (live,)


def prebuild_zoo_adjoint(verbose=True, workers=None):
    """AOT: the adjoint libraries of the model zoo (1-D classes and the decoupled components of the dim = 2 ones), so that
    torch.autograd through a zoo constructor never compiles on the GPU box."""
    import concurrent.futures as cf
    import lqg_amd
    pats = {}
    for cls, d, kw in [(lqg_amd.BoundedActor, 2, dict(dim=1)), (lqg_amd.OptimalActor, 2, dict(dim=1)),
                       (lqg_amd.RelativeObservationBoundedActor, 2, dict(dim=1)), (lqg_amd.SubjectiveActor, 2, dict(dim=1)),
                       (lqg_amd.PointMassBoundedActor, 2, {}), (lqg_amd.PointMassBoundedActor, 4, {})]:
        dims, masks, key = class_pattern(cls, d, **kw)
        live = zoo_live(cls, tuple(sorted(kw.items())), d)
        pats[adjoint_key(key, live)] = (key, dims, masks, live)
    for cls, d, kw in [(lqg_amd.BoundedActor, 4, dict(dim=2)), (lqg_amd.SubjectiveActor, 4, dict(dim=2))]:
        i = 0
        while True:
            zs = tuple(sorted(kw.items()))
            pat = zoo_component_pattern((cls, zs, d, i))
            if pat is None:
                break
            live = zoo_live(cls, zs, d, component=i)
            pats[adjoint_key(pat[2], live)] = (pat[2], pat[0], pat[1], live)
            i += 1
    with cf.ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda v: compile_adjoint_pattern(v[0], v[1], v[2], verbose=verbose, live=v[3]), pats.values()))
    return sorted(pats)


def prebuild_zoo(verbose=True, workers=None):
    """AOT: compile the specialised libraries of the model zoo (called by __graft_entry__.build())."""
    import concurrent.futures as cf
    import lqg_amd
    todo = [(lqg_amd.BoundedActor, 2, dict(dim=1)), (lqg_amd.BoundedActor, 4, dict(dim=2)),
            (lqg_amd.OptimalActor, 2, dict(dim=1)), (lqg_amd.RelativeObservationBoundedActor, 2, dict(dim=1)),
            (lqg_amd.SubjectiveActor, 2, dict(dim=1)), (lqg_amd.SubjectiveActor, 4, dict(dim=2)),
            (lqg_amd.PointMassBoundedActor, 2, {}), (lqg_amd.PointMassBoundedActor, 4, {})]
    pats = {}
    for cls, d, kw in todo:
        dims, masks, key = class_pattern(cls, d, **kw)
        pats[key] = (dims, masks)
    with cf.ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda kv: compile_pattern(kv[0], kv[1][0], kv[1][1], verbose=verbose), pats.items()))
    return sorted(pats)
