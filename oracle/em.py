"""numpy restatement of the per-step closures of the image experiments -- TEST INFRASTRUCTURE (see
oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

What it follows, for a network output `net` = nn(concat(u, v), T - t) that the caller supplies:
  experiments/imgs/inpainting.py:102-103  reverse_drift = -sde.drift(uv, T - t) + dispersion^2 * score
  experiments/sb_imgs/supr.py:84-85       reverse_drift = nn_drift(uv, T - t, param_bwd)
  inpainting.py:106-115                   rdu, rdv = unpack(reverse_drift(concat(u, v)))
  inpainting.py:122-128                   transition_sampler = us_prev + rdu dt + sqrt(dt) b normal(key, shape)
  inpainting.py:131-138 / 141-147         transition_logpdf / likelihood_logpdf = sum(norm.logpdf(., loc, scale))
  fbs/samplers/csmc/csmc.py:140,143       ancestor gather, reference pin
  fbs/data/images.py:333-363              unpack / concat as index gathers on the ravelled pixel axis
float32 arithmetic, one rounding per operation in the order written (numpy does not contract);
jax.scipy.stats.norm.logpdf is (log(2 pi scale^2) + (x - loc)^2 / scale^2) / -2; jnp.sum is the build's
pairwise tree (orc_sum).  parity unpinned against JAX.
"""
import numpy as np

from . import log as _spec_log
from . import normal as _normal

F = np.float32


def element_tables(unobs_pix, obs_pix, channels):
    """Pixel index lists (fbs/data/images.py:212-225) -> float offsets into the (w*h*c) image of the
    elements of the unobserved (p, c) and observed (q, c) parts, and the inverse map `role`."""
    c = int(channels)
    u_off = (np.asarray(unobs_pix, np.int64)[:, None] * c + np.arange(c)[None, :]).reshape(-1)
    v_off = (np.asarray(obs_pix, np.int64)[:, None] * c + np.arange(c)[None, :]).reshape(-1)
    role = np.zeros(u_off.size + v_off.size, np.int64)
    role[u_off] = np.arange(u_off.size)
    role[v_off] = ~np.arange(v_off.size)
    return u_off.astype(np.int32), v_off.astype(np.int32), role.astype(np.int32)


def to_bf16_bits(x):
    """float32 -> bfloat16 bit patterns, round to nearest even."""
    u = np.ascontiguousarray(x, F).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(h):
    return (np.asarray(h, np.uint16).astype(np.uint32) << 16).view(F)


def tree_sum_rows(x):
    """orc_sum of every row of x (rows, d): adjacent pairs level by level, an odd last node carried."""
    a = np.ascontiguousarray(x, F)
    if a.shape[1] == 0:
        return np.zeros(a.shape[0], F)
    while a.shape[1] > 1:
        m = a.shape[1]
        b = a[:, 0:m - (m & 1):2] + a[:, 1:m:2]
        a = np.concatenate([b, a[:, m - 1:m]], axis=1) if m & 1 else b
    return a[:, 0].copy()


def concat(us, A, v_prev, role):
    """img[r] = concat(us[A[r]], v_prev) (csmc.py:140 + images.py:355-363) as float32 (rows, D)."""
    us = np.asarray(us, F).reshape(us.shape[0], -1)
    rows = us if A is None else us[np.asarray(A, np.int64)]
    vp = np.asarray(v_prev, F).reshape(-1)
    img = np.empty((rows.shape[0], role.size), F)
    isu = role >= 0
    img[:, isu] = rows[:, role[isu]]
    img[:, ~isu] = vp[~role[~isu]][None, :]
    return img


def _drift(mode, cx, cs, x, s):
    if mode == 1:
        return s.astype(F)
    return (F(cx) * x).astype(F) + (F(cs) * s).astype(F)


def _logpdf_terms(target, base, drift, dt, sd):
    var = F(sd) * F(sd)
    lognorm = _spec_log(np.array([F(6.2831855) * var], F))[0]
    m = base + (drift * F(dt)).astype(F)
    df = (target - m).astype(F)
    q = ((df * df).astype(F) / var).astype(F)
    return ((lognorm + q).astype(F) * F(-0.5)).astype(F)


def finish(us, A, net, mode, cx, cs, dt, sd, v, v_prev, key, n_total, row0, pin_row, pin_value, u_off, v_off,
           want_us=True, want_lw=True):
    """-> (us_new (n, du) or None, lw (n,) or None) for the n rows [row0, row0 + n) of an ensemble of
    n_total rows; net (n, D) float32 (a bfloat16 network output is passed already widened)."""
    net = np.asarray(net, F)
    n = net.shape[0]
    du = u_off.size
    us2 = np.asarray(us, F).reshape(-1, du) if us is not None else None
    us_new = lw = None
    if want_us:
        x = us2 if A is None else us2[np.asarray(A, np.int64)]
        z = _normal(key, (int(n_total), du))[row0:row0 + n]
        d = _drift(mode, cx, cs, x, net[:, u_off])
        us_new = ((x + (d * F(dt)).astype(F)).astype(F) + (F(sd) * z).astype(F)).astype(F)
        if pin_row is not None and pin_row >= 0:
            us_new[pin_row] = np.asarray(pin_value, F).reshape(-1)
    if want_lw:
        vp = np.asarray(v_prev, F).reshape(1, -1)
        d = _drift(mode, cx, cs, np.broadcast_to(vp, (n, vp.shape[1])), net[:, v_off])
        lw = tree_sum_rows(_logpdf_terms(np.asarray(v, F).reshape(1, -1), vp, d, dt, sd))
    return us_new, lw


def transition_logpdf(us, net, mode, cx, cs, dt, sd, u, u_off):
    """inpainting.py:131-138 for every row of us (n, du)."""
    net = np.asarray(net, F)
    x = np.asarray(us, F).reshape(net.shape[0], -1)
    d = _drift(mode, cx, cs, x, net[:, u_off])
    return tree_sum_rows(_logpdf_terms(np.asarray(u, F).reshape(1, -1), x, d, dt, sd))


def forward_pass(key, us_star, bs, vs, us0, lw0, ts, coef_fn, net_fn, dt, u_off, v_off, role):
    """csmc.forward_pass (fbs/samplers/csmc/csmc.py:150-159) with conditional killing resampling over the image
    closures, for a caller-supplied network: net_fn(img (n, D) float32, t_prev) -> (n, D) float32 and
    coef_fn(t_prev) -> (mode, cx, cs, sd).  us0 (n, du) / lw0 (n,) are the initial particles and log-weights.
    -> (As (T, n), final normalised log-weights, final particles)."""
    from . import cond_killing, exp, normalise, split
    T = us_star.shape[0] - 1
    _, key_scan = split(key, 2)
    us = np.array(us0, F)
    us[bs[0]] = us_star[0]
    log_ws = normalise(lw0, True)
    keys = split(key_scan, T)
    As = []
    for k in range(T):
        kr, kt = split(keys[k], 2)
        A = cond_killing(kr, exp(log_ws), int(bs[k]), int(bs[k + 1]), True)
        mode, cx, cs, sd = coef_fn(ts[k])
        net = net_fn(concat(us, A, vs[k], role), ts[k])
        us, lw = finish(us, A, net, mode, F(cx), F(cs), F(dt), F(sd), vs[k + 1], vs[k], kt, us.shape[0], 0, int(bs[k + 1]),
                        us_star[k + 1], u_off, v_off)
        log_ws = normalise(lw, True)
        As.append(A)
    return np.stack(As), log_ws, us
