"""Oracle for the DB detection pre-processors (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates
  PPOcrDetectionPreprocessor.__call__   model/db_pp/processor_ocr_db_pp.py:103-145
  DetResizeForTest.resize_image_type0   model/db_pp/image_operators.py:269-316
  NormalizeImage / ToCHWImage           model/db_pp/image_operators.py:78-118
  OCRDetectionPreprocessor              model/db_net/processor_ocr_dbnet.py:50-102

PARITY UNPINNED for the resize: the reference calls ``cv2.resize`` (opencv-python, unpinned,
requirements.txt:3; not installed here).  ``cv2_resize_linear_u8`` restates OpenCV's 8-bit
INTER_LINEAR path (imgproc/resize.cpp: 11-bit fixed-point coefficients, HResizeLinear in int32,
VResizeLinear ``((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``, and the switch to INTER_AREA for
exact 2x decimation).  Known answers in tests/test_oracle_db_pre.py pin identity, constant images and
the 2x2 area case; the size arithmetic and the normalisation are plain numpy as in the reference.
"""
from __future__ import annotations

import math

import numpy as np


def det_plan_db_pp(h: int, w: int, limit_side_len: int = 960):
    """resize_image_type0 with limit_type='max' (image_operators.py:277-300)."""
    if max(h, w) > limit_side_len:
        ratio = float(limit_side_len) / h if h > w else float(limit_side_len) / w
    else:
        ratio = 1.0
    rh, rw = int(h * ratio), int(w * ratio)
    rh = max(int(round(rh / 32) * 32), 32)
    rw = max(int(round(rw / 32) * 32), 32)
    return rh, rw


def det_plan_db_torch(h: int, w: int, short_side: int = 736):
    """OCRDetectionPreprocessor.resize (processor_ocr_dbnet.py:50-60)."""
    if h < w:
        nh = short_side
        nw = int(math.ceil(nh / h * w / 32) * 32)
    else:
        nw = short_side
        nh = int(math.ceil(nw / w * h / 32) * 32)
    return nh, nw


def _coef(dsize: int, ssize: int, clamp_frac: bool):
    scale = float(ssize) / dsize
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0
        s[hi] = ssize - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    s0 = np.clip(s, 0, ssize - 1)
    s1 = np.clip(s + 1, 0, ssize - 1)
    return s0, s1, a0, a1


def cv2_resize_linear_u8(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """uint8 HxWxC -> uint8 nh x nw x C, OpenCV INTER_LINEAR semantics (see module docstring)."""
    h, w, _ = img.shape
    if (h, w) == (nh, nw):
        return img.copy()
    src = img.astype(np.int64)
    if w == 2 * nw and h == 2 * nh:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _coef(nw, w, True)
    y0, y1, ay0, ay1 = _coef(nh, h, False)
    hor = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]   # [h, nw, C] int
    s0 = hor[y0]
    s1 = hor[y1]
    out = (((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_db_pp(img_rgb: np.ndarray):
    """RGB uint8 HxWx3 -> (float32 CHW image, shape_list [src_h, src_w, ratio_h, ratio_w])."""
    img = img_rgb[:, :, ::-1]                      # processor_ocr_db_pp.py:124
    h, w, _ = img.shape
    nh, nw = det_plan_db_pp(h, w)
    res = cv2_resize_linear_u8(np.ascontiguousarray(img), nw, nh)
    scale = np.float32(1.0 / 255.0)
    mean = np.array([0.485, 0.456, 0.406]).reshape(1, 1, 3).astype("float32")
    std = np.array([0.229, 0.224, 0.225]).reshape(1, 1, 3).astype("float32")
    x = (res.astype("float32") * scale - mean) / std   # image_operators.py:100-101
    return x.transpose(2, 0, 1), np.array([h, w, nh / float(h), nw / float(w)])


def preprocess_db_torch(img_rgb: np.ndarray):
    """RGB uint8 -> float32 CHW, org_shape [h, w] (processor_ocr_dbnet.py:62-102)."""
    img = img_rgb[:, :, ::-1]
    h, w, _ = img.shape
    nh, nw = det_plan_db_torch(h, w)
    res = cv2_resize_linear_u8(np.ascontiguousarray(img), nw, nh)
    x = res - np.array([123.68, 116.78, 103.94], dtype=np.float32)
    x /= 255.
    return x.transpose(2, 0, 1).astype(np.float32), [h, w]
