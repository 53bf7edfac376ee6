"""Conditional resamplers, signature ``f(key, weights, i, j, conditional)``
(fbs/samplers/csmc/resamplings.py:10-125): the ancestor at slot j is forced to be i.
"""
from __future__ import annotations

import torch

from ... import _lib, ops

_KINDS = {"multinomial": 0, "killing": 1, "systematic": 2}


def _as_int(x) -> int:
    if isinstance(x, torch.Tensor):
        return int(x.item())
    return int(x)


def _cond(kind: str, key, weights, i, j, conditional) -> torch.Tensor:
    w = ops._f32c(weights, "weights").reshape(-1)
    idx = torch.empty(w.numel(), dtype=torch.int32, device=w.device)
    k0, k1 = ops._k(key)
    _lib.call("fbsmi_cond_resample", _KINDS[kind], k0, k1, w.data_ptr(), _as_int(i), _as_int(j),
              int(bool(conditional)), w.numel(), idx.data_ptr(), ops._ws(w.numel(), w.device).data_ptr(), ops._stream())
    return idx


def multinomial(key, weights, i=0, j=0, conditional: bool = True):  # resamplings.py:10-37
    return _cond("multinomial", key, weights, i, j, conditional)


def killing(key, weights, i=0, j=0, conditional: bool = True):  # resamplings.py:40-88
    return _cond("killing", key, weights, i, j, conditional)


def systematic(key, weights, i=0, j=0, conditional: bool = True):  # resamplings.py:91-125
    # the conditional variant raises NotImplementedError in the reference (:129); so does the ABI
    return _cond("systematic", key, weights, i, j, conditional)
