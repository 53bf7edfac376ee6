"""ctypes binding of libfbsmi (include/fbsmi.h) -- the only way fbs_amd reaches the GPU kernels.

There is no CPU fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is
raised.  ``build()`` compiles the HIP sources in-tree for gfx950 with hipcc.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "csrc", n) for n in ("fbsmi_prims.hip", "fbsmi_lg.hip", "fbsmi_sde.hip", "fbsmi_nn.hip", "fbsmi_em.hip")]
_DEPS = _SRC + [os.path.join(_HERE, "csrc", "fbsmi_device.h"), os.path.join(_HERE, "csrc", "fbsmi_host.h"),
                os.path.join(_HERE, "..", "include", "fbsmi.h"), os.path.join(_HERE, "..", "include", "fbsmi_math.h"),
                os.path.join(_HERE, "..", "include", "fbsmi_nn.h")]
LIB_PATH = os.path.join(_HERE, "lib", "libfbsmi.so")
# include/fbsmi_dist.h: the multi-GPU exchange steps, a library of its own (links librccl + libfbsmi)
DIST_LIB_PATH = os.path.join(_HERE, "lib", "libfbsmi_dist.so")
_DIST_SRC = os.path.join(_HERE, "csrc", "fbsmi_dist.hip")
_DIST_DEPS = [_DIST_SRC, os.path.join(_HERE, "..", "include", "fbsmi_dist.h"), os.path.join(_HERE, "..", "include", "fbsmi.h")]

# -ffp-contract=off is part of the numeric specification (include/fbsmi_math.h)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-Wno-unused-value", "-Wno-pass-failed"]
_HEADERS = _DEPS[len(_SRC):]
_OBJ_DIR = os.path.join(_HERE, "lib", "obj")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libfbsmi")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in _DEPS)


def _compile_and_link(force: bool) -> None:
    """One object per source file (kept under lib/obj/, recompiled only when its source or a header is newer), compiled
    side by side, then linked: a change to one kernel file costs one compilation, not five."""
    os.makedirs(_OBJ_DIR, exist_ok=True)
    hip = _hipcc()
    th = max(os.path.getmtime(h) for h in _HEADERS)
    jobs, objs = [], []
    for src in _SRC:
        obj = os.path.join(_OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), th):
            tmp = f"{obj}.{os.getpid()}.tmp"
            jobs.append((subprocess.Popen([hip] + HIPCC_FLAGS + ["-c", "-o", tmp, src]), tmp, obj, src))
    for proc, tmp, obj, src in jobs:
        if proc.wait() != 0:
            for q, t2, _, _ in jobs:
                if q.poll() is None:
                    q.wait()
                if os.path.exists(t2):
                    os.remove(t2)
            raise RuntimeError(f"hipcc failed on {src}")
        os.replace(tmp, obj)
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    subprocess.check_call([hip, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, LIB_PATH)


def _rocm_lib_dir() -> str:
    for cand in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if cand and os.path.exists(os.path.join(cand, "lib", "librccl.so")):
            return os.path.join(cand, "lib")
    raise RuntimeError("librccl.so not found: cannot build libfbsmi_dist")


def _dist_stale() -> bool:
    if not os.path.exists(DIST_LIB_PATH):
        return True
    t = os.path.getmtime(DIST_LIB_PATH)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in _DIST_DEPS)


def build_dist(force: bool = False) -> str:
    """Compile libfbsmi_dist.so (after libfbsmi.so, which it links) if it is missing or older than its sources."""
    build(force)
    if not (force or _dist_stale()):
        return DIST_LIB_PATH
    if not all(os.path.exists(p) for p in _DIST_DEPS):
        if os.path.exists(DIST_LIB_PATH) and not force:
            return DIST_LIB_PATH
        raise RuntimeError("libfbsmi_dist sources missing and no prebuilt library at " + DIST_LIB_PATH)
    import fcntl
    with open(DIST_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _dist_stale():
                tmp = f"{DIST_LIB_PATH}.{os.getpid()}.tmp"
                subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp,
                                       _DIST_SRC, "-L" + os.path.dirname(LIB_PATH), "-lfbsmi", "-L" + _rocm_lib_dir(), "-lrccl",
                                       "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + _rocm_lib_dir()])
                os.replace(tmp, DIST_LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return DIST_LIB_PATH


def build(force: bool = False) -> str:
    """Compile libfbsmi.so for gfx950 if it is missing or older than its sources.  Safe when several processes call it at
    once (the ranks of a multi-GPU launch): one builds under a file lock, into temporary files that are renamed into place,
    the others wait and find it fresh."""
    have_src = all(os.path.exists(p) for p in _DEPS)
    if not (force or _stale()):
        return LIB_PATH
    if not have_src:
        if os.path.exists(LIB_PATH) and not force:
            return LIB_PATH
        raise RuntimeError("libfbsmi sources missing and no prebuilt library at " + LIB_PATH)
    import fcntl
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():
                _compile_and_link(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class LGModelStruct(C.Structure):
    _fields_ = [("du", C.c_int32), ("dv", C.c_int32), ("T", C.c_int32), ("dt", C.c_float),
                ("G", C.c_void_p), ("g", C.c_void_p), ("sd", C.c_void_p), ("lognorm", C.c_void_p),
                ("F", C.c_void_p), ("sqQ", C.c_void_p)]


class EMMaskStruct(C.Structure):
    _fields_ = [("du", C.c_int32), ("dv", C.c_int32), ("u_off", C.c_void_p), ("v_off", C.c_void_p),
                ("role", C.c_void_p)]


# name -> (restype, argtypes); every int-returning entry is status-checked by call()
_vp, _i32, _i64, _u32, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float
SIGNATURES = {
    "fbsmi_abi_version": (C.c_int, []),
    "fbsmi_last_error": (C.c_char_p, []),
    "fbsmi_key_split": (None, [_u32, _u32, C.c_int, C.POINTER(C.c_uint32)]),
    "fbsmi_random_bits": (C.c_int, [_u32, _u32, _i64, _vp, _vp]),
    "fbsmi_uniform": (C.c_int, [_u32, _u32, _i64, _vp, _vp]),
    "fbsmi_normal": (C.c_int, [_u32, _u32, _i64, _vp, _vp]),
    "fbsmi_random_range": (C.c_int, [C.c_int, _u32, _u32, _i64, _i64, _i64, _vp, _vp]),
    "fbsmi_randint": (C.c_int, [_u32, _u32, _i64, _i32, _i32, _vp, _vp]),
    "fbsmi_math_map": (C.c_int, [C.c_int, _vp, _vp, _i64, _vp, _vp]),
    "fbsmi_workspace_bytes": (C.c_size_t, [_i64]),
    "fbsmi_cumsum": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "fbsmi_sum": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "fbsmi_logsumexp": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "fbsmi_normalise": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _vp, _vp]),
    "fbsmi_normalise_ess": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "fbsmi_searchsorted": (C.c_int, [_vp, _i32, _vp, _i64, _vp, _vp]),
    "fbsmi_resample": (C.c_int, [C.c_int, _vp, _u32, _u32, _i32, _vp, _vp, _vp]),
    "fbsmi_cond_resample": (C.c_int, [C.c_int, _u32, _u32, _vp, _i32, _i32, C.c_int, _i32, _vp, _vp, _vp]),
    "fbsmi_categorical": (C.c_int, [_u32, _u32, _vp, _i32, _vp, _vp, _vp]),
    "fbsmi_force_move": (C.c_int, [_u32, _u32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "fbsmi_gather_rows": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "fbsmi_set_row": (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    "fbsmi_backtrace": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "fbsmi_linear_path": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "fbsmi_affine_em_path": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, C.c_int, _vp, _vp]),
    "fbsmi_em_update": (C.c_int, [_vp, _vp, _f, _f, _u32, _u32, _i64, _i64, _i64, _vp, _vp]),
    "fbsmi_lg_transition_sampler": (C.c_int, [C.POINTER(LGModelStruct), _i32, _f, _f, _vp, _vp, _u32, _u32, _i64, _vp, _vp]),
    "fbsmi_lg_transition_sampler_rows": (C.c_int, [C.POINTER(LGModelStruct), _i32, _f, _f, _vp, _vp, _u32, _u32, _i64, _i64, _i64, _vp, _vp]),
    "fbsmi_lg_likelihood_logpdf": (C.c_int, [C.POINTER(LGModelStruct), _i32, _f, _f, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fbsmi_lg_transition_logpdf": (C.c_int, [C.POINTER(LGModelStruct), _i32, _f, _f, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fbsmi_lg_sweep_create": (C.c_int, [C.POINTER(LGModelStruct), _i32, C.c_int, C.c_int, C.c_int, _i32, C.POINTER(_vp)]),
    "fbsmi_lg_sweep_destroy": (None, [_vp]),
    "fbsmi_lg_gibbs_sweep": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "fbsmi_lg_gibbs_chain": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_int, _vp]),
    "fbsmi_lg_sweep_set_group": (C.c_int, [_vp, _i32, _i32]),
    "fbsmi_lg_gibbs_chain_groups": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, C.c_int, _vp]),
    "fbsmi_lg_sweep_view": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(_i64), _vp]),
    "fbsmi_lg_filter_create": (C.c_int, [C.POINTER(LGModelStruct), _i32, C.c_int, C.c_int, C.c_int, _i32, C.POINTER(_vp)]),
    "fbsmi_lg_filter_destroy": (None, [_vp]),
    "fbsmi_lg_filter_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "fbsmi_lg_sweep_profile": (C.c_int, [_vp, C.c_int]),
    "fbsmi_lg_sweep_kernel_us": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "fbsmi_em_concat": (C.c_int, [C.POINTER(EMMaskStruct), _vp, _vp, _vp, _i64, C.c_int, _vp, _vp]),
    "fbsmi_em_finish": (C.c_int, [C.POINTER(EMMaskStruct), _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f, _f, _f, _f, _vp, _vp,
                                  _u32, _u32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "fbsmi_em_transition_logpdf": (C.c_int, [C.POINTER(EMMaskStruct), _vp, _vp, C.c_int, C.c_int, _f, _f, _f, _f, _vp,
                                             _i64, _vp, _vp]),
    # include/fbsmi_nn.h
    "fbsmi_nn_linear_attention": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _vp]),
    "fbsmi_nn_qkv_linear_attention": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "fbsmi_nn_conv3x3": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _i32, _vp]),
    "fbsmi_nn_conv3x3_supported": (C.c_int, [_i32, _i32, _i32, _i32]),   # a 0 / 1 answer, not a status: use lib() directly
    "fbsmi_nn_proj64": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _f, _vp, _vp, _i64, _vp]),
    "fbsmi_nn_channel_layernorm": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _vp, _f, _vp, _vp, _vp]),
    "fbsmi_nn_groupnorm_silu": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fbsmi_nn_bias_add": (C.c_int, [_vp, C.c_int, _i64, _i32, _vp, _vp]),
    "fbsmi_nn_pixel_shuffle": (C.c_int, [_vp, _vp, C.c_int, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libfbsmi.so (building it first when the sources are newer)."""
    global _lib
    if _lib is None:
        # FBSMI_LIB: a diagnostic build of the same sources (tools/build_variants.sh) instead of the in-tree library
        path = os.environ.get("FBSMI_LIB") or build()
        try:
            L = C.CDLL(path)
        except OSError as e:  # no silent fallback
            raise RuntimeError(f"cannot load the HIP extension {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.fbsmi_abi_version() != 1:
            raise RuntimeError("libfbsmi ABI version mismatch")
        _lib = L
    return _lib


def call(name: str, *args):
    """Call an int-status entry point; raise RuntimeError with fbsmi_last_error() on failure."""
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        msg = L.fbsmi_last_error().decode("utf-8", "replace")
        if rc == -3:
            raise NotImplementedError(msg)
        raise RuntimeError(f"{name} failed ({rc}): {msg}")
    return rc


# ---- include/fbsmi_dist.h ----
DIST_SIGNATURES = {
    "fbsmi_dist_abi_version": (C.c_int, []),
    "fbsmi_dist_last_error": (C.c_char_p, []),
    "fbsmi_dist_unique_id": (C.c_int, [_vp]),
    "fbsmi_dist_create": (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.POINTER(_vp)]),
    "fbsmi_dist_destroy": (C.c_int, [_vp]),
    "fbsmi_dist_shard": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "fbsmi_dist_logsumexp": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "fbsmi_dist_resample_exchange": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_int, _vp]),
    "fbsmi_dist_window_export": (C.c_int, [_vp, _i64, _vp]),
    "fbsmi_dist_window_open": (C.c_int, [_vp, _vp]),
    "fbsmi_dist_window_publish": (C.c_int, [_vp, _vp, _i64, _vp]),
    "fbsmi_dist_window_read_row": (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
}

_dist = None


def dist_lib() -> C.CDLL:
    """Load libfbsmi_dist.so (and with it librccl); only multi-GPU callers come here."""
    global _dist
    if _dist is None:
        lib()                                                   # libfbsmi first: the dist library links it
        path = build_dist()
        try:
            L = C.CDLL(path)
        except OSError as e:
            raise RuntimeError(f"cannot load the HIP extension {path}: {e}") from e
        for name, (res, args) in DIST_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.fbsmi_dist_abi_version() != 1:
            raise RuntimeError("libfbsmi_dist ABI version mismatch")
        _dist = L
    return _dist


def dist_call(name: str, *args):
    L = dist_lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.fbsmi_dist_last_error().decode('utf-8', 'replace')}")
    return rc
