"""CPU oracle for the fbs sampler hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings of ``oracle/fbs_oracle.c`` (a plain-C restatement of the reference's
``fbs.samplers`` / ``fbs.sdes`` hot path on JAX's published PRNG / cumsum / searchsorted
semantics) plus thin numpy conveniences.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; ``fbs_amd`` never does.

Pinning status (see the header of fbs_oracle.c): ``split``/``random_bits`` are pinned bit-for-bit
by the reference fixture ``experiments/keys.npy``; every floating-point primitive is PARITY
UNPINNED against JAX (JAX is not installable here) and is pinned instead by closed-form
statistical known-answer tests restated from the reference's ``tests/``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfbs_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "fbs_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fbsmi_math.h")
    so_omp = os.path.join(_HERE, "libfbs_oracle_omp.so")
    stale = (not os.path.exists(_SO)) or (not os.path.exists(so_omp)) or any(
        os.path.exists(p) and os.path.getmtime(p) > min(os.path.getmtime(_SO), os.path.getmtime(so_omp))
        for p in (src, hdr))
    if force or stale:
        if not os.path.exists(src):
            raise RuntimeError("oracle source missing and no prebuilt libfbs_oracle.so")
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:   # several ranks may get here at once: one make at a time
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return _SO


_lib = None
_lib_omp = None
_SO_OMP = os.path.join(_HERE, "libfbs_oracle_omp.so")


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def lib_omp():
    """The OpenMP build of the same source (independent particle loops on several host threads;
    bit-identical results).  Only bench.py's cpu_baseline leg needs it."""
    global _lib_omp
    if _lib_omp is None:
        build()
        _lib_omp = C.CDLL(_SO_OMP)
        _declare(_lib_omp)
    return _lib_omp


_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class LGStruct(C.Structure):
    _fields_ = [("du", C.c_int32), ("dv", C.c_int32), ("T", C.c_int32), ("dt", C.c_float),
                ("G", C.c_void_p), ("g", C.c_void_p), ("sd", C.c_void_p), ("lognorm", C.c_void_p),
                ("F", C.c_void_p), ("sqQ", C.c_void_p)]


def _declare(L):
    vp = C.c_void_p
    L.orc_threefry2x32.argtypes = [_u32p, C.c_uint32, C.c_uint32, _u32p]
    L.orc_random_bits.argtypes = [_u32p, C.c_int64, _u32p]
    L.orc_split.argtypes = [_u32p, C.c_int, _u32p]
    L.orc_uniform.argtypes = [_u32p, C.c_int64, _f32p]
    L.orc_normal.argtypes = [_u32p, C.c_int64, _f32p]
    L.orc_randint.argtypes = [_u32p, C.c_int64, C.c_int32, C.c_int32, _i32p]
    L.orc_cumsum.argtypes = [_f32p, C.c_int64, _f32p]
    L.orc_sum.argtypes = [_f32p, C.c_int64]
    L.orc_sum.restype = C.c_float
    L.orc_max.argtypes = [_f32p, C.c_int64]
    L.orc_max.restype = C.c_float
    L.orc_searchsorted.argtypes = [_f32p, C.c_int32, C.c_float]
    L.orc_searchsorted.restype = C.c_int32
    L.orc_choice.argtypes = [_u32p, _f32p, C.c_int32, C.c_int64, _i32p]
    L.orc_logsumexp.argtypes = [_f32p, C.c_int64]
    L.orc_logsumexp.restype = C.c_float
    L.orc_normalise.argtypes = [_f32p, C.c_int64, C.c_int]
    L.orc_ess.argtypes = [_f32p, C.c_int64]
    L.orc_ess.restype = C.c_float
    for name in ("orc_exp", "orc_log", "orc_erfinv", "orc_log1p", "orc_sqrt"):
        getattr(L, name).argtypes = [_f32p, C.c_int64, _f32p]
    L.orc_div.argtypes = [_f32p, _f32p, C.c_int64, _f32p]
    L.orc_exp_monotone_violations.argtypes = [C.c_float, C.c_float]
    L.orc_exp_monotone_violations.restype = C.c_int64
    for name in ("orc_systematic", "orc_stratified", "orc_multinomial", "orc_killing"):
        getattr(L, name).argtypes = [_f32p, _u32p, C.c_int32, _i32p]
    for name in ("orc_cond_multinomial", "orc_cond_killing", "orc_cond_systematic"):
        getattr(L, name).argtypes = [_u32p, _f32p, C.c_int32, C.c_int32, C.c_int, C.c_int32, _i32p]
    L.orc_cond_systematic.restype = C.c_int
    L.orc_categorical.argtypes = [_u32p, _f32p, C.c_int32]
    L.orc_categorical.restype = C.c_int32
    L.orc_force_move.argtypes = [_u32p, _f32p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    L.orc_force_move.restype = C.c_int32
    lg = C.POINTER(LGStruct)
    L.orc_lg_transition_sampler.argtypes = [lg, C.c_int, _f32p, _f32p, _u32p, C.c_int32, _f32p]
    L.orc_lg_likelihood_logpdf.argtypes = [lg, C.c_int, _f32p, _f32p, _f32p, C.c_int32, _f32p]
    L.orc_lg_transition_logpdf.argtypes = [lg, C.c_int, _f32p, _f32p, _f32p, C.c_int32, _f32p]
    L.orc_lg_fwd_sampler.argtypes = [lg, _u32p, _f32p, C.c_int32, _f32p]
    L.orc_csmc_forward_pass_lg.argtypes = [lg, _u32p, _f32p, _i32p, _f32p, _f32p, _f32p, C.c_int32, C.c_int,
                                           vp, vp, vp, _f32p, _f32p]
    L.orc_backward_scanning_pass.argtypes = [_u32p, _i32p, _f32p, _f32p, C.c_int32, C.c_int32, C.c_int32,
                                             _f32p, _i32p]
    L.orc_backward_sampling_pass_lg.argtypes = [lg, _u32p, _f32p, _f32p, _f32p, C.c_int32, _f32p, _i32p]
    L.orc_gibbs_kernel_lg.argtypes = [lg, _u32p, _f32p, _f32p, _i32p, C.c_int32, C.c_int, C.c_int, _f32p, _f32p,
                                      _i32p, _u8p, vp, vp]
    L.orc_gibbs_chain_lg.argtypes = [lg, _u32p, _f32p, _f32p, _i32p, C.c_int32, C.c_int, C.c_int, C.c_int32, vp]
    L.orc_bootstrap_filter_lg.argtypes = [lg, _u32p, _f32p, _f32p, C.c_int32, C.c_int, vp, vp]
    L.orc_bootstrap_filter_lg.restype = C.c_float
    L.orc_pmcmc_filter_step_lg.argtypes = [lg, _u32p, _f32p, _f32p, C.c_int32, C.c_int, _f32p]
    L.orc_pmcmc_filter_step_lg.restype = C.c_float
    L.orc_backward_smoother_lg.argtypes = [lg, _u32p, _f32p, _f32p, C.c_int32, _f32p]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_bench_gibbs_lg.argtypes = [lg, C.c_uint32, _f32p, _f32p, C.c_int32, C.c_int32, _f32p]


# ------------------------------------------------------------------------------------------------
# jax.random-shaped helpers
# ------------------------------------------------------------------------------------------------
def _key(key) -> np.ndarray:
    k = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2))
    return k


def _f32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _i32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def threefry2x32(key, c0: int, c1: int) -> np.ndarray:
    out = np.zeros(2, np.uint32)
    lib().orc_threefry2x32(_key(key), c0, c1, out)
    return out


def random_bits(key, n: int) -> np.ndarray:
    out = np.zeros(int(n), np.uint32)
    lib().orc_random_bits(_key(key), int(n), out)
    return out


def split(key, num: int = 2) -> np.ndarray:
    out = np.zeros((int(num), 2), np.uint32)
    lib().orc_split(_key(key), int(num), out.reshape(-1))
    return out


def uniform(key, shape=()) -> np.ndarray:
    n = int(np.prod(shape, dtype=np.int64))
    out = np.zeros(n, np.float32)
    lib().orc_uniform(_key(key), n, out)
    return out.reshape(shape)


def normal(key, shape=()) -> np.ndarray:
    n = int(np.prod(shape, dtype=np.int64))
    out = np.zeros(n, np.float32)
    lib().orc_normal(_key(key), n, out)
    return out.reshape(shape)


def randint(key, shape, minval: int, maxval: int) -> np.ndarray:
    n = int(np.prod(shape, dtype=np.int64))
    out = np.zeros(n, np.int32)
    lib().orc_randint(_key(key), n, int(minval), int(maxval), out)
    return out.reshape(shape)


def cumsum(x) -> np.ndarray:
    x = _f32(x)
    out = np.zeros_like(x)
    lib().orc_cumsum(x, x.size, out)
    return out


def tree_sum(x) -> np.float32:
    x = _f32(x)
    return np.float32(lib().orc_sum(x, x.size))


def searchsorted(a, q) -> np.ndarray:
    a = _f32(a)
    q = np.atleast_1d(_f32(q))
    return np.array([lib().orc_searchsorted(a, a.size, float(v)) for v in q], dtype=np.int32)


def choice(key, w, shape=()) -> np.ndarray:
    w = _f32(w)
    m = int(np.prod(shape, dtype=np.int64))
    out = np.zeros(m, np.int32)
    lib().orc_choice(_key(key), w, w.size, m, out)
    return out.reshape(shape)


def logsumexp(x) -> np.float32:
    x = _f32(x)
    return np.float32(lib().orc_logsumexp(x, x.size))


def normalise(lw, log_space: bool = False) -> np.ndarray:
    out = _f32(lw).copy()
    lib().orc_normalise(out, out.size, int(log_space))
    return out


def ess(lw) -> np.float32:
    """1 / sum w^2 of the normalised weights (orc_ess): a diagnostic of this build, not of the reference."""
    x = _f32(lw)
    return np.float32(lib().orc_ess(x, x.size))


def _map1(name, x):
    x = _f32(x)
    out = np.zeros_like(x)
    getattr(lib(), name)(x.reshape(-1), x.size, out.reshape(-1))
    return out


def exp(x):
    return _map1("orc_exp", x)


def log(x):
    return _map1("orc_log", x)


def log1p(x):
    return _map1("orc_log1p", x)


def erfinv(x):
    return _map1("orc_erfinv", x)


def sqrt(x):
    return _map1("orc_sqrt", x)


def bits_to_normal(bits) -> np.ndarray:
    """fbsmi_bits_to_normal (include/fbsmi_math.h) on an array of random words: jax.random.normal's map
    u = max(lo, unit * (hi - lo) + lo), sqrt(2) * erf_inv(u), one float32 rounding per operation."""
    b = np.ascontiguousarray(bits, np.uint32)
    unit = ((b >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo = np.float32(-0.99999994)
    u = (unit * np.float32(2.0)).astype(np.float32) + lo
    u = np.where(u < lo, lo, u).astype(np.float32)
    return (np.float32(1.41421354) * erfinv(u)).astype(np.float32)


def div(x, y):
    x, y = _f32(x), _f32(y)
    out = np.zeros_like(x)
    lib().orc_div(x.reshape(-1), y.reshape(-1), x.size, out.reshape(-1))
    return out


# ------------------------------------------------------------------------------------------------
# resamplers, same argument orders as the reference
# ------------------------------------------------------------------------------------------------
def _uncond(name, weights, key):
    w = _f32(weights)
    idx = np.zeros(w.size, np.int32)
    getattr(lib(), name)(w, _key(key), w.size, idx)
    return idx


def systematic(weights, key):  # fbs/samplers/resampling.py:54
    return _uncond("orc_systematic", weights, key)


def stratified(weights, key):  # fbs/samplers/resampling.py:58
    return _uncond("orc_stratified", weights, key)


def multinomial(weights, key):  # fbs/samplers/resampling.py:62
    return _uncond("orc_multinomial", weights, key)


def killing(weights, key):  # fbs/samplers/resampling.py:71
    return _uncond("orc_killing", weights, key)


def _cond(name, key, weights, i, j, conditional):
    w = _f32(weights)
    idx = np.zeros(w.size, np.int32)
    rc = getattr(lib(), name)(_key(key), w, int(i), int(j), int(bool(conditional)), w.size, idx)
    if name == "orc_cond_systematic" and rc != 0:
        raise NotImplementedError("Not implemented, not used.")  # csmc/resamplings.py:129
    return idx


def cond_multinomial(key, weights, i=0, j=0, conditional=True):  # csmc/resamplings.py:10
    return _cond("orc_cond_multinomial", key, weights, i, j, conditional)


def cond_killing(key, weights, i=0, j=0, conditional=True):  # csmc/resamplings.py:40
    return _cond("orc_cond_killing", key, weights, i, j, conditional)


def cond_systematic(key, weights, i=0, j=0, conditional=True):  # csmc/resamplings.py:91
    return _cond("orc_cond_systematic", key, weights, i, j, conditional)


def force_move(key, weights, k):  # fbs/samplers/gibbs.py:171
    w = _f32(weights)
    alpha = C.c_float(0.0)
    i = lib().orc_force_move(_key(key), w, int(k), w.size, C.byref(alpha))
    return int(i), np.float32(alpha.value)


# ------------------------------------------------------------------------------------------------
# Linear-Gaussian model.  The tables are INPUTS to the oracle: float32
# arrays produced by the caller.  `lg_tables_f64` below is the oracle's own float64 derivation of
# them straight from the reference's closures, used by tests to check the product's table builder.
# ------------------------------------------------------------------------------------------------
class LGModel:
    """Holds float32 per-step tables and the ctypes struct that points at them."""

    def __init__(self, du, dv, dt, G, g, sd, lognorm, F, sqQ):
        self.du, self.dv = int(du), int(dv)
        self.D = self.du + self.dv
        self.G = _f32(G)
        self.T = int(self.G.shape[0])
        self.g, self.sd, self.lognorm = _f32(g), _f32(sd), _f32(lognorm)
        self.F, self.sqQ = _f32(F), _f32(sqQ)
        self.dt = np.float32(dt)
        assert self.G.shape == (self.T, self.D, self.D) and self.g.shape == (self.T, self.D)
        for a in (self.sd, self.lognorm, self.F, self.sqQ):
            assert a.shape == (self.T,)
        self.struct = LGStruct(self.du, self.dv, self.T, float(self.dt), self.G.ctypes.data, self.g.ctypes.data,
                               self.sd.ctypes.data, self.lognorm.ctypes.data, self.F.ctypes.data,
                               self.sqQ.ctypes.data)

    @property
    def ref(self):
        return C.byref(self.struct)


def sde_const(a, b):
    """StationaryConstLinearSDE, fbs/sdes/linear.py:13-45: (drift coeff a(t), dispersion b(t), F, Q)."""
    a, b = float(a), float(b)
    return dict(a=lambda t: a, b=lambda t: b,
                FQ=lambda t, s: (np.exp(a * (t - s)), b ** 2 / (2 * a) * (np.exp(2 * a * (t - s)) - 1)))


def sde_lin(beta_min, beta_max, t0, T):
    """StationaryLinLinearSDE, fbs/sdes/linear.py:48-92."""
    bmin, bmax, t0, T = map(float, (beta_min, beta_max, t0, T))

    def beta(t):
        return (bmax - bmin) / (T - t0) * t + (bmin * T - bmax * t0) / (T - t0)

    def beta_int(t, s):
        return 0.5 * (t - s) * ((bmax - bmin) / (T - t0) * (t + s) + 2 * (bmin * T - bmax * t0) / (T - t0))

    return dict(a=lambda t: -0.5 * beta(t), b=lambda t: np.sqrt(beta(t)),
                FQ=lambda t, s: (np.exp(-0.5 * beta_int(t, s)), 1 - np.exp(-beta_int(t, s))))


def lg_tables_f64(m0, cov0, sde, ts, du, dt=None):
    """Float64 derivation of the affine reverse-drift tables from the reference's closures:
    forward_m_cov / score / reverse_drift of experiments/toy/gp_gibbs.py:73-107."""
    m0 = np.asarray(m0, np.float64)
    cov0 = np.asarray(cov0, np.float64)
    ts = np.asarray(ts, np.float64)
    D = m0.size
    T = ts.size - 1
    Tend = ts[-1]
    dt = (Tend - ts[0]) / T if dt is None else float(dt)
    G = np.zeros((T, D, D))
    g = np.zeros((T, D))
    sd = np.zeros(T)
    F = np.zeros(T)
    sqQ = np.zeros(T)
    for k in range(T):
        tau = ts[k]                      # reverse time t_prev
        t_fwd = Tend - tau               # forward time
        Ft, Qt = sde["FQ"](t_fwd, ts[0])
        mt = Ft * m0
        covt = Ft ** 2 * cov0 + Qt * np.eye(D)
        P = np.linalg.inv(covt)
        a_t, b_t = sde["a"](t_fwd), sde["b"](t_fwd)
        G[k] = -a_t * np.eye(D) - b_t ** 2 * P
        g[k] = b_t ** 2 * (P @ mt)
        sd[k] = np.sqrt(dt) * b_t
        Fk, Qk = sde["FQ"](ts[k + 1], ts[k])
        F[k], sqQ[k] = Fk, np.sqrt(Qk)
    lognorm = np.log(2 * np.pi * sd ** 2)
    return dict(du=du, dv=D - du, dt=dt, G=G, g=g, sd=sd, lognorm=lognorm, F=F, sqQ=sqQ)


def make_lg(m0, cov0, sde, ts, du, dt=None) -> LGModel:
    return LGModel(**lg_tables_f64(m0, cov0, sde, ts, du, dt))


def lg_transition_sampler(m: LGModel, k, us_prev, v_prev, key):
    us_prev = _f32(us_prev).reshape(-1, m.du)
    out = np.zeros_like(us_prev)
    lib().orc_lg_transition_sampler(m.ref, int(k), us_prev, _f32(v_prev), _key(key), us_prev.shape[0], out)
    return out


def lg_likelihood_logpdf(m: LGModel, k, v, us_prev, v_prev):
    us_prev = _f32(us_prev).reshape(-1, m.du)
    out = np.zeros(us_prev.shape[0], np.float32)
    lib().orc_lg_likelihood_logpdf(m.ref, int(k), _f32(v), us_prev, _f32(v_prev), us_prev.shape[0], out)
    return out


def lg_transition_logpdf(m: LGModel, k, u, us_prev, v_prev):
    us_prev = _f32(us_prev).reshape(-1, m.du)
    out = np.zeros(us_prev.shape[0], np.float32)
    lib().orc_lg_transition_logpdf(m.ref, int(k), _f32(u), us_prev, _f32(v_prev), us_prev.shape[0], out)
    return out


def lg_fwd_sampler(m: LGModel, key, xy0):
    xy0 = _f32(xy0).reshape(-1)
    path = np.zeros((m.T + 1, xy0.size), np.float32)
    lib().orc_lg_fwd_sampler(m.ref, _key(key), xy0, xy0.size, path)
    return path


def csmc_forward_pass_lg(m: LGModel, key, us_star, bs_star, vs, us0, lw0, resampler="killing", store=True):
    """csmc.forward_pass (csmc.py:80-164). Returns dict(As, log_wss, uss, us_last, lw_last)."""
    us0 = _f32(us0).reshape(-1, m.du)
    n = us0.shape[0]
    As = np.zeros((m.T, n), np.int32) if store else None
    lws = np.zeros((m.T + 1, n), np.float32) if store else None
    uss = np.zeros((m.T + 1, n, m.du), np.float32) if store else None
    us_last = np.zeros((n, m.du), np.float32)
    lw_last = np.zeros(n, np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data
    lib().orc_csmc_forward_pass_lg(m.ref, _key(key), _f32(us_star).reshape(m.T + 1, m.du), _i32(bs_star),
                                   _f32(vs).reshape(m.T + 1, m.dv), us0, _f32(lw0), n,
                                   {"killing": 0, "multinomial": 1}[resampler], ptr(As), ptr(lws), ptr(uss),
                                   us_last, lw_last)
    return dict(As=As, log_wss=lws, uss=uss, us_last=us_last, lw_last=lw_last)


def backward_scanning_pass(key, As, uss, log_w_T):
    As, uss = _i32(As), _f32(uss)
    T, n = As.shape
    du = uss.shape[2]
    xs = np.zeros((T + 1, du), np.float32)
    bs = np.zeros(T + 1, np.int32)
    lib().orc_backward_scanning_pass(_key(key), As, uss, _f32(log_w_T), T, n, du, xs, bs)
    return xs, bs


def backward_sampling_pass_lg(m: LGModel, key, vs, uss, log_wss):
    uss = _f32(uss)
    n = uss.shape[1]
    xs = np.zeros((m.T + 1, m.du), np.float32)
    bs = np.zeros(m.T + 1, np.int32)
    lib().orc_backward_sampling_pass_lg(m.ref, _key(key), _f32(vs).reshape(m.T + 1, m.dv), uss, _f32(log_wss), n,
                                        xs, bs)
    return xs, bs


def gibbs_kernel_lg(m: LGModel, key, x0, y0, bs_star, nparticles, explicit_backward=True, explicit_final=False,
                    debug=False):
    """gibbs_kernel (gibbs.py:68-168) -> (x0, us_star, bs_star, acc[, us_T, lw_T])."""
    n = nparticles + 1 if explicit_final else nparticles
    x0n = np.zeros(m.du, np.float32)
    usn = np.zeros((m.T + 1, m.du), np.float32)
    bsn = np.zeros(m.T + 1, np.int32)
    acc = np.zeros(m.T + 1, np.uint8)
    usT = np.zeros((n, m.du), np.float32) if debug else None
    lwT = np.zeros(n, np.float32) if debug else None
    ptr = lambda a: None if a is None else a.ctypes.data
    lib().orc_gibbs_kernel_lg(m.ref, _key(key), _f32(x0).reshape(m.du), _f32(y0).reshape(m.dv), _i32(bs_star),
                              int(nparticles), int(explicit_backward), int(explicit_final), x0n, usn, bsn, acc,
                              ptr(usT), ptr(lwT))
    if debug:
        return x0n, usn, bsn, acc.astype(bool), usT, lwT
    return x0n, usn, bsn, acc.astype(bool)


def gibbs_chain_lg(m: LGModel, key, x0, y0, bs_star, nparticles, nsweeps, explicit_backward=True,
                   explicit_final=False, keep=True):
    """nsweeps sweeps with the key chain of tests/test_gibbs.py:115-118. Returns (key, x0, bs_star, x0s)."""
    key = _key(key).copy()
    x0 = _f32(x0).reshape(m.du).copy()
    bs = _i32(bs_star).copy()
    x0s = np.zeros((nsweeps, m.du), np.float32) if keep else None
    lib().orc_gibbs_chain_lg(m.ref, key, x0, _f32(y0).reshape(m.dv), bs, int(nparticles), int(explicit_backward),
                             int(explicit_final), int(nsweeps), None if x0s is None else x0s.ctypes.data)
    return key, x0, bs, x0s


def gibbs_chains_lg(m: LGModel, key, x0s, y0, bs_stars, nparticles, nsweeps, explicit_backward=True,
                    explicit_final=False):
    """C chains driven like experiments/toy/gp_gibbs.py:182-187: per sweep key, subkey = split(key);
    key_chains = split(subkey, C); chain c runs gibbs_kernel(key_chains[c], x0s[c], y0, _, bs_stars[c]).
    Returns (key, x0s (C,du), bs_stars (C,T+1), samples (nsweeps,C,du))."""
    key = _key(key).copy()
    x0s = _f32(x0s).reshape(-1, m.du).copy()
    bss = _i32(bs_stars).reshape(-1, m.T + 1).copy()
    Cn = x0s.shape[0]
    out = np.zeros((nsweeps, Cn, m.du), np.float32)
    for i in range(nsweeps):
        key, subkey = split(key, 2)
        kc = split(subkey, Cn)
        for c in range(Cn):
            x0n, _, bsn, _ = gibbs_kernel_lg(m, kc[c], x0s[c], y0, bss[c], nparticles, explicit_backward,
                                              explicit_final)
            x0s[c], bss[c] = x0n, bsn
        out[i] = x0s
    return key, x0s, bss, out


_RES = {"stratified": 0, "systematic": 1, "multinomial": 2, "killing": 3}


def bootstrap_filter_lg(m: LGModel, key, vs, init_samples, resampling="stratified", return_last=True):
    init = _f32(init_samples).reshape(-1, m.du)
    n = init.shape[0]
    last = np.zeros((n, m.du), np.float32)
    filt = None if return_last else np.zeros((m.T + 1, n, m.du), np.float32)
    nell = lib().orc_bootstrap_filter_lg(m.ref, _key(key), _f32(vs).reshape(m.T + 1, m.dv), init, n,
                                         _RES[resampling], last.ctypes.data,
                                         None if filt is None else filt.ctypes.data)
    return (last if return_last else filt), np.float32(nell)


def pmcmc_filter_step_lg(m: LGModel, key, vs, u0s, resampling="stratified"):
    u0s = _f32(u0s).reshape(-1, m.du)
    uT = np.zeros_like(u0s)
    ell = lib().orc_pmcmc_filter_step_lg(m.ref, _key(key), _f32(vs).reshape(m.T + 1, m.dv), u0s, u0s.shape[0],
                                         _RES[resampling], uT)
    return uT, np.float32(ell)


def backward_smoother_lg(m: LGModel, key, filter_us, vs):
    fu = _f32(filter_us)
    traj = np.zeros((m.T + 1, m.du), np.float32)
    lib().orc_backward_smoother_lg(m.ref, _key(key), fu, _f32(vs).reshape(m.T + 1, m.dv), fu.shape[1], traj)
    return traj


def bench_gibbs_lg(m: LGModel, seed, x0, y0, nparticles, nsweeps, threads: int = 1):
    """nsweeps explicit-backward Gibbs sweeps (the cpu_baseline workload).  threads > 1 uses the OpenMP
    build; returns (x0 after the sweeps, threads actually used)."""
    out = np.zeros(m.du, np.float32)
    L = lib() if threads <= 1 else lib_omp()
    if threads > 1:
        L.orc_set_num_threads(int(threads))
    used = int(L.orc_num_threads()) if threads > 1 else 1
    L.orc_bench_gibbs_lg(m.ref, int(seed), _f32(x0).reshape(m.du), _f32(y0).reshape(m.dv), int(nparticles),
                         int(nsweeps), out)
    return out, used


# ------------------------------------------------------------------------------------------------
# numpy restatements of the remaining hot-path functions (composed from the C primitives above)
# ------------------------------------------------------------------------------------------------
def csmc_kernel_lg(m: LGModel, key, us_star, bs_star, vs, us0, lw0, backward=False):
    """csmc_kernel (csmc.py:14-77) with killing resampling."""
    key_fwd, key_bwd = split(key, 2)                                                # :65
    fp = csmc_forward_pass_lg(m, key_fwd, us_star, bs_star, vs, us0, lw0, "killing")
    if backward:
        return backward_sampling_pass_lg(m, key_bwd, vs, fp["uss"], fp["log_wss"])  # :73
    return backward_scanning_pass(key_bwd, fp["As"], fp["uss"], fp["log_wss"][-1])  # :75


def pcn_proposal(key, delta, x, mean, sampler):
    """smc.py:161-168 in float32."""
    f = np.float32
    beta = 2 / (2 + delta)
    k = split(key, 2)
    r0, r1 = sampler(k[0]), sampler(k[1])
    p = x + f(np.sqrt(delta / 2)) * (r0 - mean)
    return (f(beta) * p + f(1 - beta) * mean + f(np.sqrt(1 - beta)) * (r1 - mean)).astype(f)


def lg_ref_sampler(m0, cov0, FQ_T, du, key, yT, n):
    """ref_sampler of gp_pmcmc.py:130-133 / gp_gibbs.py:138-141: p(u0 | v0) at the terminal time."""
    Ft, Qt = FQ_T
    m0 = np.asarray(m0, np.float64)
    D = m0.size
    m_ref = Ft * m0
    cov_ref = Ft ** 2 * np.asarray(cov0, np.float64) + Qt * np.eye(D)
    gain = cov_ref[:du, du:] @ np.linalg.inv(cov_ref[du:, du:])
    m_ = m_ref[:du] + gain @ (np.asarray(yT, np.float64).reshape(-1) - m_ref[du:])
    cov_ = cov_ref[:du, :du] - gain @ cov_ref[du:, :du]
    chol = np.linalg.cholesky(cov_)
    z = normal(key, (n, du))
    return (m_.astype(np.float32) + z @ chol.astype(np.float32)).astype(np.float32)


def pmcmc_kernel_lg(m: LGModel, key, uT, log_ell, ys, y0, nparticles, ref_sampler, mean_path=None, delta=None,
                    which_u=0, resampling="stratified"):
    """pmcmc_kernel (smc.py:171-258) with the LG closures.  ys, mean_path: (T+1, dv)."""
    f = np.float32
    key_prop, key_u0, key_filter, key_mh = split(key, 4)                           # :231
    fwd_ys = lambda k: lg_fwd_sampler(m, k, _f32(y0).reshape(-1))
    if delta is None:
        prop_ys = fwd_ys(key_prop)
    else:
        prop_ys = pcn_proposal(key_prop, delta, _f32(ys), _f32(mean_path), fwd_ys)  # :237
    vs = prop_ys[::-1].copy()                                                        # :239
    u0s = ref_sampler(key_u0, vs[0], nparticles)                                     # :241
    prop_uTs, prop_log_ell = pmcmc_filter_step_lg(m, key_filter, vs, u0s, resampling)  # :242
    prop_uT = prop_uTs[which_u]
    log_acc = np.minimum(f(0.0), f(prop_log_ell) - f(log_ell))                       # :246
    z = uniform(key_mh, ())                                                          # :248
    acc = bool(log(np.array([z], f))[0] < log_acc)                                   # :249
    if acc:
        return prop_uT, f(prop_log_ell), prop_ys, acc
    return _f32(uT), f(log_ell), _f32(ys), acc


def euler_maruyama_np(key, x0, ts, drift, dispersion, integration_nsteps=1, return_path=False):
    """fbs/sdes/simulators.py:53-106 in float32 numpy (drift / dispersion: python callables)."""
    f = np.float32
    ts = np.asarray(ts, np.float64)
    n = ts.size - 1
    keys = split(key, n)
    x = _f32(x0).copy()
    path = [x.copy()]
    for k in range(n):
        t, t_next = float(ts[k]), float(ts[k + 1])
        ddt = abs(t_next - t) / integration_nsteps
        rnds = normal(keys[k], (integration_nsteps,) + x.shape)
        for j, t_ in enumerate(np.linspace(t, t_next - ddt, integration_nsteps)):
            x = (x + drift(x, float(t_)) * f(ddt) + f(dispersion(float(t_)) * float(np.sqrt(ddt))) * rnds[j]).astype(f)
        path.append(x.copy())
    return np.stack(path) if return_path else x


def doob_bridge_np(key, A, B, S, ddt, x0, xT, T, nsub, replace):
    """doob_bridge_simulator (simulators.py:126-160) for an affine bridge drift A x + B xT, given the
    float32 coefficient tables the product feeds its kernel (so the comparison isolates the kernel)."""
    f = np.float32
    keys = split(key, T)
    x = _f32(x0).reshape(-1).copy()
    tg = _f32(xT).reshape(-1)
    D = x.size
    out = [x.copy()]
    A, B, S, ddt = _f32(A), _f32(B), _f32(S), _f32(ddt)
    for k in range(T):
        h = ddt[k]
        sq = sqrt(np.array([h], f))[0]
        xi = normal(keys[k], (nsub, D))
        for j in range(nsub):
            r = k * nsub + j
            drift = (A[r] * x + B[r] * tg).astype(f)
            x = ((x + drift * h).astype(f) + (f(S[r] * sq) * xi[j]).astype(f)).astype(f)
        out.append(x.copy())
    out = np.stack(out)
    if replace:
        out[-1] = tg
    return out


def discrete_time_simulator_np(key, x0, ts, f, q):
    """fbs/sdes/simulators.py:109-123: X(t_{k+1}) = f(X(t_k), t_{k+1}, t_k) + q(t_{k+1}, t_k) w, float32, one
    rounding per operation; f / q python callables on numpy values."""
    ts = np.asarray(ts, np.float64)
    x = _f32(x0).copy()
    rnds = normal(key, (ts.size - 1,) + x.shape)                                       # :122
    for k in range(ts.size - 1):
        x = (_f32(f(x, float(ts[k + 1]), float(ts[k]))) + (np.float32(q(float(ts[k + 1]), float(ts[k]))) * rnds[k]).astype(np.float32)).astype(np.float32)
    return x


def twisted_smc_np(key, y, ts, init_sampler, transition_logpdf, twisting_logpdf, twisting_prop_sampler,
                   twisting_prop_logpdf, resampling, nparticles):
    """fbs/samplers/smc.py:261-309 with numpy closures; `resampling(weights, key)` one of this module's resamplers.
    -> (samples, normalised log-weights, [ancestor indices per step])."""
    ts = np.asarray(ts, np.float64)
    nsteps = ts.size - 1
    key_init, key_filter = split(key, 2)                                               # :298
    keys = split(key_filter, nsteps)                                                   # :299
    xs = _f32(init_sampler(key_init, nparticles))                                      # :301
    log_ps = _f32(twisting_logpdf(y, xs, ts[0]))                                       # :302
    log_ws = normalise(log_ps, True)                                                   # :303
    inds_all = []
    for k in range(nsteps):                                                            # scan over (keys, ts[1:]) :306-307
        key_resampling, key_prop = split(keys[k], 2)                                   # :281
        t_prev = ts[k + 1]
        inds = resampling(exp(log_ws), key_resampling)                                 # :284
        xs_prev, log_ps_prev = xs[inds], log_ps[inds]                                  # :285-286
        xs = _f32(twisting_prop_sampler(key_prop, xs_prev, t_prev, y))                 # :289
        log_ps = _f32(twisting_logpdf(y, xs, t_prev))                                  # :292
        lw = ((_f32(transition_logpdf(xs, xs_prev, t_prev)) + log_ps).astype(np.float32)
              - _f32(twisting_prop_logpdf(xs, xs_prev, t_prev, y))).astype(np.float32)
        log_ws = normalise((lw - log_ps_prev).astype(np.float32), True)                # :293-295
        inds_all.append(inds)
    return xs, log_ws, inds_all


def gibbs_kernel_lg_marg_y(m: LGModel, key, x0, y0, bs_star, nparticles, bridge):
    """gibbs_kernel (gibbs.py:68-168) with marg_y=True, explicit_backward=True, explicit_final=False: the observation
    path is re-drawn by bridge_sampler (gibbs.py:17-20,130).  `bridge(key, y_first, y_last)` -> (T+1, dv) is the
    Doob-bridge restatement (doob_bridge_np with the caller's coefficient tables)."""
    x0, y0 = _f32(x0).reshape(m.du), _f32(y0).reshape(m.dv)
    bs_star = _i32(bs_star)
    key_fwd, key_csmc, key_bridge = split(key, 3)                                      # :126
    path = lg_fwd_sampler(m, key_fwd, np.concatenate([x0, y0]))                        # :127
    path_x, path_y = path[:, :m.du], path[:, m.du:]
    us = path_x[::-1].copy()                                                           # :129
    vs = bridge(key_bridge, path_y[0], path_y[-1])[::-1].copy()                        # :130
    us0 = np.tile(us[0][None, :], (nparticles, 1)).astype(np.float32)                  # :140-141
    lw0 = np.full(nparticles, -np.log(nparticles), np.float32)                         # :143-144
    k_fwd, k_x0, k_us, k_bs = split(key_csmc, 4)                                       # :147
    fw = csmc_forward_pass_lg(m, k_fwd, us, bs_star, vs, us0, lw0, store=False)        # :148
    idx, _ = force_move(k_x0, exp(fw["lw_last"]), int(bs_star[-1]))                    # :152
    x0n = fw["us_last"][idx]                                                           # :154
    us_next = lg_fwd_sampler(m, k_us, np.concatenate([x0n, y0]))[:, :m.du][::-1].copy()  # :155
    bs_next = randint(k_bs, (m.T + 1,), 0, nparticles)                                 # :156
    return us_next[-1], us_next, bs_next, bs_next != bs_star                           # :167-168
