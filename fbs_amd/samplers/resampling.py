"""Unconditional resamplers, signature ``f(weights, key)`` (fbs/samplers/resampling.py:54-101).

Each call is a few HIP kernel launches (canonical-tree CDF + bisection searchsorted); indices come
back as an int32 tensor on the GPU.
"""
from __future__ import annotations

import torch

from .. import _lib, ops

_KINDS = {"stratified": 0, "systematic": 1, "multinomial": 2, "killing": 3}


def _resample(kind: str, weights: torch.Tensor, key) -> torch.Tensor:
    w = ops._f32c(weights, "weights").reshape(-1)
    idx = torch.empty(w.numel(), dtype=torch.int32, device=w.device)
    k0, k1 = ops._k(key)
    _lib.call("fbsmi_resample", _KINDS[kind], w.data_ptr(), k0, k1, w.numel(), idx.data_ptr(),
              ops._ws(w.numel(), w.device).data_ptr(), ops._stream())
    return idx


def systematic(weights, key):  # resampling.py:54-55
    return _resample("systematic", weights, key)


def stratified(weights, key):  # resampling.py:58-59
    return _resample("stratified", weights, key)


def multinomial(weights, key):  # resampling.py:62-68
    return _resample("multinomial", weights, key)


def killing(weights, key):  # resampling.py:71-101
    return _resample("killing", weights, key)
