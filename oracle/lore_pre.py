"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of Lore's image pre-process.

``lore_preprocess`` follows TableLorePreProcessor.process, /root/reference/src/pdftable/model/lore/processer_lore.py:66-109
(upper_left=False branch: centre/scale affine, cv2.warpAffine INTER_LINEAR, (x/255 - mean)/std, HWC->CHW, int64 meta).
``warp_affine_u8`` restates cv2.warpAffine (opencv-python is an unpinned, un-vendored dependency, requirements.txt:3,
and is not installed): OpenCV imgwarp.cpp WarpAffineInvoker -- inverse map in float64, 10-bit fixed-point coordinates
with 1/32-pixel interpolation positions, 15-bit bilinear weights, constant (0) border.  PARITY UNPINNED: pinned only by
the known-answer tests in tests/test_oracle_lore.py (identity, integer shifts, exact 2x down-scale).
"""
from __future__ import annotations

import numpy as np
import torch

from .crnn import _bilinear_tab
from .lore_decode import lore_preprocess_geometry

MEAN = np.array([0.408, 0.447, 0.470], dtype=np.float32)      # processer_lore.py:67-70
STD = np.array([0.289, 0.274, 0.278], dtype=np.float32)


def invert_affine(M: np.ndarray) -> np.ndarray:
    """cv::invertAffineTransform's arithmetic as inlined in cv::warpAffine (float64)."""
    m = np.asarray(M, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m.reshape(2, 3)


def warp_affine_u8(img: np.ndarray, M: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.warpAffine(img, M, (out_w, out_h), flags=INTER_LINEAR) for uint8 HxWx3, BORDER_CONSTANT 0."""
    tab = _bilinear_tab()
    h, w, ch = img.shape
    m = invert_affine(M).reshape(6)
    xs = np.arange(out_w, dtype=np.float64)
    sat = lambda v: np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)   # saturate_cast<int>(double)
    adelta = sat(m[0] * xs * 1024.0)
    bdelta = sat(m[3] * xs * 1024.0)
    src = img.astype(np.int64)
    out = np.zeros((out_h, out_w, ch), np.uint8)
    for y in range(out_h):
        X0 = int(sat(np.float64((m[1] * y + m[2]) * 1024.0))) + 16
        Y0 = int(sat(np.float64((m[4] * y + m[5]) * 1024.0))) + 16
        X = (X0 + adelta) >> 5
        Y = (Y0 + bdelta) >> 5
        sx = np.clip(X >> 5, -32768, 32767)
        sy = np.clip(Y >> 5, -32768, 32767)
        wts = tab[Y & 31, X & 31]
        acc = np.zeros((out_w, ch), np.int64)
        for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            pix = np.zeros((out_w, ch), np.int64)
            pix[ok] = src[yy[ok], xx[ok]]
            acc += pix * wts[:, k:k + 1]
        out[y] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out


def lore_preprocess(img: np.ndarray, inp_h: int = 1024, inp_w: int = 1024, upper_left: bool = False):
    """img uint8 HxWx3 in the channel order the reference's ``process`` receives (BGR for path / PIL inputs)
    -> (pixel_values f32 [1,3,inp_h,inp_w], meta int64 [7])."""
    h, w = img.shape[:2]
    trans, meta = lore_preprocess_geometry(h, w, inp_h, inp_w, upper_left)
    # cv2.resize(img, (width, height)) to its own size is a copy
    warped = warp_affine_u8(img, trans, inp_w, inp_h)
    x = ((warped / 255.0 - MEAN.reshape(1, 1, 3)) / STD.reshape(1, 1, 3)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None], meta
