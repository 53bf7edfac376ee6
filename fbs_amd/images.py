"""Image restoration masks and the X/Y split (fbs/data/images.py:212-363, 413-421).

``ImageRestore.unpack`` / ``concat`` are what every image closure of the reference calls around the
score network (experiments/imgs/inpainting.py:106-115): unpack gathers the unobserved pixels X (p, c)
and the observed pixels Y (q, c) of an image (w, h, c) by ravelled pixel indices; concat is the inverse
scatter.  Masks are generated with the JAX-compatible PRNG, so a given key yields the reference's mask.
Dataset loading / corruption operators are out of scope (no datasets in this environment).
"""
from __future__ import annotations

import itertools
from typing import NamedTuple, Tuple

import numpy as np
import torch

from . import ops


class InpaintingMask(NamedTuple):
    """fbs/data/images.py:212-219."""
    width: int
    height: int
    shift: int
    unobs_inds_ravelled: torch.Tensor
    obs_inds_ravelled: torch.Tensor


class SRMask(NamedTuple):
    """fbs/data/images.py:222-225."""
    rate: int
    unobs_inds_ravelled: torch.Tensor
    obs_inds_ravelled: torch.Tensor


class ImageRestore:
    """fbs/data/images.py:228-363 without the dataset part."""

    def __init__(self, task: str, image_shape: Tuple[int, int, int], sr_random: bool = True, device=None):
        self.image_shape = tuple(image_shape)
        self.task = task
        w, h, c = image_shape
        s = int(task.split('-')[-1])
        if 'inpaint' in task:
            self.unobs_shape = (s ** 2, c)
        elif 'supr' in task:
            self.unobs_shape = (int(w * h * (s ** 2 - 1) / s ** 2), c)
        else:
            raise ValueError(f'Unknown task {task}.')
        self.sr_random = sr_random
        self.device = torch.device(device) if device is not None else ops._default_device()

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(self.device)

    def _gen_supr_mask(self, key, rate: int, random: bool = True) -> SRMask:  # :258-282
        img_w, img_h = self.image_shape[:2]
        nblocks = int(img_w * img_h / rate ** 2)
        if random:
            shifts = ops.randint(key, (nblocks, 2), 0, rate, device=self.device).cpu().numpy().astype(np.int64)
        else:
            shifts = np.ones((nblocks, 2), dtype=np.int64) * (rate // 2)
        block_inds = np.asarray(list(itertools.product(range(0, img_w, rate), range(0, img_h, rate))), np.int64)
        all_inds = np.asarray(list(itertools.product(range(img_w), range(img_h))), np.int64)
        block_r = np.ravel_multi_index([block_inds[:, 0] + shifts[:, 0], block_inds[:, 1] + shifts[:, 1]],
                                       (img_w, img_h), mode='clip')
        all_r = np.ravel_multi_index([all_inds[:, 0], all_inds[:, 1]], (img_w, img_h), mode='clip')
        unobs = np.setdiff1d(all_r, block_r, assume_unique=True)
        return SRMask(rate, unobs_inds_ravelled=self._dev(unobs), obs_inds_ravelled=self._dev(block_r))

    def _gen_inpaint_mask(self, key, width: int, height: int) -> InpaintingMask:  # :284-302
        img_w, img_h = self.image_shape[:2]
        width, height = min(width, img_w), min(height, img_h)
        rect_inds = np.asarray(list(itertools.product(range(width), range(height))), np.int64)
        all_inds = np.asarray(list(itertools.product(range(img_w), range(img_h))), np.int64)
        max_shift = min(img_w, img_h) - max(width, height)
        shift = int(ops.randint(key, (), 0, max_shift, device=self.device).item())
        rect_r = np.ravel_multi_index([rect_inds[:, 0] + shift, rect_inds[:, 1] + shift], (img_w, img_h), mode='clip')
        all_r = np.ravel_multi_index([all_inds[:, 0], all_inds[:, 1]], (img_w, img_h), mode='clip')
        obs = np.setdiff1d(all_r, rect_r, assume_unique=True)
        return InpaintingMask(width, height, shift, unobs_inds_ravelled=self._dev(rect_r),
                              obs_inds_ravelled=self._dev(obs))

    def gen_mask(self, key):  # :304-312
        s = int(self.task.split('-')[-1])
        if 'inpaint' in self.task:
            return self._gen_inpaint_mask(key, s, s)
        if 'supr' in self.task:
            return self._gen_supr_mask(key, s, random=self.sr_random)
        raise ValueError(f'Unknown task {self.task}.')

    def unpack(self, xy: torch.Tensor, mask) -> Tuple[torch.Tensor, torch.Tensor]:  # :333-353
        """(..., w, h, c) -> unobserved (..., p, c), observed (..., q, c)."""
        img_w, img_h, img_c = self.image_shape
        flat = xy.reshape(*xy.shape[:-3], img_w * img_h, img_c)
        x = torch.index_select(flat, -2, mask.unobs_inds_ravelled)
        y = torch.index_select(flat, -2, mask.obs_inds_ravelled)
        return x, y

    def concat(self, x: torch.Tensor, y: torch.Tensor, mask) -> torch.Tensor:  # :355-363
        """The reverse operation of `unpack`."""
        img_w, img_h, img_c = self.image_shape
        lead = x.shape[:-2]
        img = torch.zeros((*lead, img_w * img_h, img_c), dtype=x.dtype, device=x.device)
        img.index_copy_(-2, mask.unobs_inds_ravelled, x)
        img.index_copy_(-2, mask.obs_inds_ravelled, y.expand(*lead, *y.shape[-2:]) if y.dim() < x.dim() else y)
        return img.reshape(*lead, img_w, img_h, img_c)


def normalise(img: torch.Tensor, method: str = 'clip') -> torch.Tensor:  # :413-421
    if method == 'clip':
        return torch.clamp(img, 0., 1.)
    mins = torch.amin(img, dim=(-2, -3), keepdim=True)
    maxs = torch.amax(img, dim=(-2, -3), keepdim=True)
    return (img - mins) / (maxs - mins)
