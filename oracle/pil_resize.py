"""Oracle for Pillow's bilinear ``Image.resize`` on 8-bit images (TEST INFRASTRUCTURE, see oracle/__init__.py).

The PP-LCNet classifiers' pre-processor (model/cls/image_processing_pplcnet.py:272-304, 327-455) resizes with
``transformers.image_transforms.resize`` -> ``PIL.Image.resize((w, h), resample=BILINEAR)``.  Pillow is a third-party
dependency of the reference (requirements.txt, unpinned; 12.2.0 in this image); its algorithm (libImaging/Resample.c,
``ImagingResample`` with the triangle filter) is restated here:

* a separable two-pass convolution, horizontal first, with an 8-bit intermediate image;
* per output coordinate: centre = (xx + 0.5) * scale, support = max(scale, 1), taps [int(centre - support + 0.5),
  int(centre + support + 0.5)) clipped to the image, weights (1 - |x| / max(scale, 1)) normalised in double precision,
  then quantised to 22 fractional bits (round half away from zero);
* accumulation in int32 starting from 1 << 21, result = clip(acc >> 22, 0, 255).
A pass whose input and output size are equal is skipped.  Pinned against Pillow itself by tests/golden/pil_resize.npz
(tests/golden/make_golden.py pil_resize) and, where Pillow is importable, live in tests/test_oracle_pplcnet.py."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size: int, out_size: int):
    """-> (xmin int32 [out], xmax int32 [out] (tap counts), coefficients int32 [out, ksize])"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, np.int32)
    xmaxs = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        # normalize_coeffs_8bpc: round half away from zero through C's (int) truncation
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx] = np.trunc(q).astype(np.int32)
        xmins[xx], xmaxs[xx] = xmin, xmax
    return xmins, xmaxs, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    in_size = img.shape[axis]
    xmins, xmaxs, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        n = int(xmaxs[xx])
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[xmins[xx]:xmins[xx] + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [h, w, c] -> uint8 [out_h, out_w, c], Pillow's Image.resize((out_w, out_h), BILINEAR)"""
    assert img.dtype == np.uint8 and img.ndim == 3
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0)
    return img


IMAGENET_MEAN = (0.485, 0.456, 0.406)     # transformers.utils.IMAGENET_DEFAULT_MEAN / _STD (image_processing_pplcnet.py:267-268)
IMAGENET_STD = (0.229, 0.224, 0.225)


def pplcnet_preprocess(img_rgb: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PPLCNetImageProcessor.preprocess (image_processing_pplcnet.py:327-455) for one RGB uint8 image:
    resize -> * 1/255 -> (x - mean) / std -> float32 [3, out_h, out_w]"""
    r = pil_resize_bilinear_u8(img_rgb, out_h, out_w).astype(np.float32) * np.float32(1 / 255)
    r = (r - np.array(IMAGENET_MEAN, np.float32)) / np.array(IMAGENET_STD, np.float32)
    return np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32)
