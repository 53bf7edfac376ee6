"""Device primitives of the sampler hot path on torch (ROCm) tensors, backed by libfbsmi.

``fbs_amd.random``-style PRNG draws follow JAX's threefry semantics (keys are ``uint32[2]`` host
arrays, explicit and stateless like ``jax.random``).  Every function launches HIP kernels on the
current torch stream; none falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_ws_cache: dict = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(f"fbs_amd: {name} must live on the GPU (got {t.device}); there is no CPU path")


def _ws(n: int, device) -> torch.Tensor:
    """Scratch for the tree kernels, cached per (device, stream) and grown on demand."""
    need = int(_lib.lib().fbsmi_workspace_bytes(int(n)))
    key = (torch.device(device).index, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _f32c(x: torch.Tensor, name="tensor") -> torch.Tensor:
    _require_cuda(x, name)
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    return x.contiguous()


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("fbs_amd: no GPU visible; the sampler engine has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


# ------------------------------------------------------------------------------------------------
# PRNG (jax.random semantics)
# ------------------------------------------------------------------------------------------------
def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def _k(key):
    if isinstance(key, torch.Tensor):
        key = key.detach().cpu().numpy()
    k = np.asarray(key, dtype=np.uint32).reshape(2)
    return int(k[0]), int(k[1])


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split(key, num) -> (num, 2) uint32 host array (pure integer work on the host)."""
    k0, k1 = _k(key)
    out = np.zeros((int(num), 2), dtype=np.uint32)
    _lib.lib().fbsmi_key_split(k0, k1, int(num), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def _numel(shape) -> int:
    if isinstance(shape, int):
        shape = (shape,)
    return int(np.prod(shape, dtype=np.int64)), tuple(shape)


def random_bits(key, shape, device=None) -> torch.Tensor:
    n, shape = _numel(shape)
    out = torch.empty(n, dtype=torch.int32, device=device or _default_device())
    k0, k1 = _k(key)
    _lib.call("fbsmi_random_bits", k0, k1, n, out.data_ptr(), _stream())
    return out.reshape(shape)


def _draw_rows(mode, key, shape, rows, device):
    """Rows [offset, offset+count) of the draw of `shape` (leading axis = rows): identical values to
    slicing the full draw -- what one rank of a sharded ensemble generates."""
    n, shape = _numel(shape)
    offset, count = int(rows[0]), int(rows[1])
    rowlen = n // shape[0] if shape and shape[0] else 0
    out = torch.empty(count * rowlen, dtype=torch.float32, device=device or _default_device())
    k0, k1 = _k(key)
    _lib.call("fbsmi_random_range", mode, k0, k1, n, offset * rowlen, count * rowlen, out.data_ptr(), _stream())
    return out.reshape((count,) + tuple(shape[1:]))


def uniform(key, shape=(), device=None, rows=None) -> torch.Tensor:
    if rows is not None:
        return _draw_rows(1, key, shape, rows, device)
    n, shape = _numel(shape)
    out = torch.empty(n, dtype=torch.float32, device=device or _default_device())
    k0, k1 = _k(key)
    _lib.call("fbsmi_uniform", k0, k1, n, out.data_ptr(), _stream())
    return out.reshape(shape)


def normal(key, shape=(), device=None, rows=None) -> torch.Tensor:
    if rows is not None:
        return _draw_rows(2, key, shape, rows, device)
    n, shape = _numel(shape)
    out = torch.empty(n, dtype=torch.float32, device=device or _default_device())
    k0, k1 = _k(key)
    _lib.call("fbsmi_normal", k0, k1, n, out.data_ptr(), _stream())
    return out.reshape(shape)


def randint(key, shape, minval: int, maxval: int, device=None) -> torch.Tensor:
    n, shape = _numel(shape)
    out = torch.empty(n, dtype=torch.int32, device=device or _default_device())
    k0, k1 = _k(key)
    _lib.call("fbsmi_randint", k0, k1, n, int(minval), int(maxval), out.data_ptr(), _stream())
    return out.reshape(shape)


def categorical(key, weights: torch.Tensor) -> torch.Tensor:
    """jax.random.choice(key, n, (), p=weights) -> int32 scalar tensor (stays on the device)."""
    w = _f32c(weights, "weights").reshape(-1)
    out = torch.empty(1, dtype=torch.int32, device=w.device)
    k0, k1 = _k(key)
    _lib.call("fbsmi_categorical", k0, k1, w.data_ptr(), w.numel(), out.data_ptr(), _ws(w.numel(), w.device).data_ptr(),
              _stream())
    return out.reshape(())


def choice(key, a, shape=(), p=None, axis: int = 0):
    """Subset of jax.random.choice used by the reference: with replacement; ``a`` an int or a tensor
    indexed along axis 0; ``p`` optional weights."""
    n_inputs = int(a) if isinstance(a, int) else a.shape[axis]
    n, shp = _numel(shape)
    if p is None:
        dev = a.device if isinstance(a, torch.Tensor) else _default_device()
        ind = randint(key, shp, 0, n_inputs, device=dev)
    else:
        w = _f32c(p, "p").reshape(-1)
        if n == 1 and shp == ():
            ind = categorical(key, w)
        else:
            c = cumsum(w)
            u = uniform(key, (n,), device=w.device)
            r = c[-1] * (1 - u)
            ind = searchsorted(c, r).reshape(shp)
    if isinstance(a, int):
        return ind
    return torch.index_select(a, axis, ind.reshape(-1).long()).reshape(*shp, *a.shape[1:]) if shp else a[ind.long()]


# ------------------------------------------------------------------------------------------------
# numeric-specification probes
# ------------------------------------------------------------------------------------------------
_MATH_OPS = {"exp": 0, "log": 1, "log1p": 2, "erfinv": 3, "sqrt": 4, "div": 5, "bits_to_normal": 6, "bits_to_normal_kernel": 7, "div_kernel": 8}


def math_map(op: str, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
    _require_cuda(x, "x")
    x = x.contiguous()
    out = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
    yp = _f32c(y).data_ptr() if y is not None else None
    _lib.call("fbsmi_math_map", _MATH_OPS[op], x.data_ptr(), yp, x.numel(), out.data_ptr(), _stream())
    return out.reshape(x.shape)


# ------------------------------------------------------------------------------------------------
# tree reductions / scans
# ------------------------------------------------------------------------------------------------
def cumsum(x: torch.Tensor) -> torch.Tensor:
    """jnp.cumsum of a 1-D float32 tensor in lax.associative_scan order."""
    x = _f32c(x, "x").reshape(-1)
    out = torch.empty_like(x)
    if x.numel():
        _lib.call("fbsmi_cumsum", x.data_ptr(), x.numel(), out.data_ptr(), _ws(x.numel(), x.device).data_ptr(), _stream())
    return out


def tree_sum(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x, "x").reshape(-1)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.call("fbsmi_sum", x.data_ptr(), x.numel(), out.data_ptr(), _ws(x.numel(), x.device).data_ptr(), _stream())
    return out.reshape(())


def logsumexp(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x, "x").reshape(-1)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.call("fbsmi_logsumexp", x.data_ptr(), x.numel(), out.data_ptr(), _ws(x.numel(), x.device).data_ptr(), _stream())
    return out.reshape(())


def normalise(log_weights: torch.Tensor, log_space: bool = False, return_lse: bool = False, return_ess: bool = False):
    """fbs/samplers/csmc/csmc.py:273-292.  return_lse / return_ess add the logsumexp (the step's increment of the log
    normalising constant) and the effective sample size 1 / sum w^2 as 0-d device tensors (no host synchronisation)."""
    lw = _f32c(log_weights, "log_weights").reshape(-1)
    out = torch.empty_like(lw)
    lse = torch.empty(1, dtype=torch.float32, device=lw.device)
    if return_ess:
        ess = torch.empty(1, dtype=torch.float32, device=lw.device)
        _lib.call("fbsmi_normalise_ess", lw.data_ptr(), lw.numel(), int(bool(log_space)), out.data_ptr(), lse.data_ptr(),
                  ess.data_ptr(), _ws(lw.numel(), lw.device).data_ptr(), _stream())
        return (out, lse.reshape(()), ess.reshape(())) if return_lse else (out, ess.reshape(()))
    _lib.call("fbsmi_normalise", lw.data_ptr(), lw.numel(), int(bool(log_space)), out.data_ptr(), lse.data_ptr(),
              _ws(lw.numel(), lw.device).data_ptr(), _stream())
    return (out, lse.reshape(())) if return_lse else out


def searchsorted(a: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    a = _f32c(a, "a").reshape(-1)
    qq = _f32c(q, "q")
    out = torch.empty(qq.numel(), dtype=torch.int32, device=a.device)
    _lib.call("fbsmi_searchsorted", a.data_ptr(), a.numel(), qq.data_ptr(), qq.numel(), out.data_ptr(), _stream())
    return out.reshape(qq.shape)


# ------------------------------------------------------------------------------------------------
# data movement
# ------------------------------------------------------------------------------------------------
def take_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """jnp.take(src, idx, axis=0) for a float32 tensor of any trailing shape."""
    _require_cuda(src, "src")
    if src.dtype != torch.float32:
        return torch.index_select(src, 0, idx.long())
    s = src.contiguous()
    i = idx.to(torch.int32).contiguous()
    d = int(np.prod(s.shape[1:], dtype=np.int64)) if s.dim() > 1 else 1
    out = torch.empty((i.numel(),) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
    _lib.call("fbsmi_gather_rows", s.data_ptr(), i.data_ptr(), i.numel(), d, out.data_ptr(), _stream())
    return out


def set_row(dst: torch.Tensor, row: int, value: torch.Tensor) -> torch.Tensor:
    """x.at[row].set(value): functional (returns a new tensor)."""
    out = _f32c(dst, "dst").clone()
    v = _f32c(value.to(out.device) if isinstance(value, torch.Tensor) else torch.as_tensor(value, device=out.device))
    d = int(np.prod(out.shape[1:], dtype=np.int64)) if out.dim() > 1 else 1
    v = v.expand(out.shape[1:]).contiguous() if out.dim() > 1 else v.reshape(1)
    _lib.call("fbsmi_set_row", out.data_ptr(), int(row), v.data_ptr(), d, _stream())
    return out


def backtrace(As: torch.Tensor, B_T: torch.Tensor) -> torch.Tensor:
    """Bs[T] = B_T; Bs[k-1] = As[k-1, Bs[k]] (csmc.py:262-267)."""
    A = As.to(torch.int32).contiguous()
    _require_cuda(A, "As")
    T, n = A.shape
    b = B_T.to(torch.int32).reshape(1).contiguous()
    out = torch.empty(T + 1, dtype=torch.int32, device=A.device)
    _lib.call("fbsmi_backtrace", A.data_ptr(), T, n, b.data_ptr(), out.data_ptr(), _stream())
    return out
