"""numpy restatement of the mask / unpack / concat part of fbs/data/images.py:212-363 -- TEST
INFRASTRUCTURE (see oracle/__init__.py)."""
import itertools

import numpy as np

from . import randint


def gen_supr_mask(key, image_shape, rate, random=True):  # images.py:258-282
    img_w, img_h = image_shape[:2]
    nblocks = int(img_w * img_h / rate ** 2)
    shifts = randint(key, (nblocks, 2), 0, rate).astype(np.int64) if random else np.ones((nblocks, 2), np.int64) * (rate // 2)
    block = np.asarray(list(itertools.product(range(0, img_w, rate), range(0, img_h, rate))), np.int64)
    obs = (np.clip(block[:, 0] + shifts[:, 0], 0, img_w - 1)) * img_h + np.clip(block[:, 1] + shifts[:, 1], 0, img_h - 1)
    unobs = np.array(sorted(set(range(img_w * img_h)) - set(obs.tolist())), np.int64)
    return unobs, obs


def gen_inpaint_mask(key, image_shape, width, height):  # images.py:284-302
    img_w, img_h = image_shape[:2]
    width, height = min(width, img_w), min(height, img_h)
    rect = np.asarray(list(itertools.product(range(width), range(height))), np.int64)
    max_shift = min(img_w, img_h) - max(width, height)
    shift = int(randint(key, (), 0, max_shift))
    unobs = np.clip(rect[:, 0] + shift, 0, img_w - 1) * img_h + np.clip(rect[:, 1] + shift, 0, img_h - 1)
    obs = np.array(sorted(set(range(img_w * img_h)) - set(unobs.tolist())), np.int64)
    return shift, unobs, obs


def unpack(xy, image_shape, unobs, obs):  # images.py:333-353
    w, h, c = image_shape
    flat = xy.reshape(*xy.shape[:-3], w * h, c)
    return flat[..., unobs, :], flat[..., obs, :]


def concat(x, y, image_shape, unobs, obs):  # images.py:355-363
    w, h, c = image_shape
    img = np.zeros((*x.shape[:-2], w * h, c), x.dtype)
    img[..., unobs, :] = x
    img[..., obs, :] = y
    return img.reshape(*img.shape[:-2], w, h, c)
