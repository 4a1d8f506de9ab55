"""Oracle for the CRNN text-line recogniser and its pre/post steps (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates
  CRNN.forward                         model/crnn/modeling_crnn.py:92-113 (layers :40-90)
  BidirectionalLSTM.forward            model/crnn/modeling_crnn.py:26-33 (nn.LSTM equations, PyTorch docs)
  OCRRecognitionPreprocessor           model/ocr_recognition/processor_ocr_recognition.py:44-115 (no chunking for CRNN:
                                       configuration_ocr_document.py:43-48 -- img 32 x 640)
  OCRRecognition.postprocess           model/ocr_recognition/modeling_ocr_recognition.py:168-184 (greedy CTC)
  OcrCommonUtils.order_point/crop_image utils/ocr/ocr_common_utils.py:214-304

Pinning: ``crnn_forward_fp32`` is pinned to the reference ``CRNN`` module by tests/golden/crnn.npz; ``order_point`` by
tests/golden/db_host_numpy.npz.  PARITY UNPINNED for ``cv2.getPerspectiveTransform`` / ``cv2.warpPerspective`` /
``cv2.resize`` (opencv-python not installed; restated from OpenCV imgproc imgwarp.cpp: the inverse map is evaluated in
double, quantised to 1/32 pixel (INTER_BITS = 5), and blended with the 15-bit fixed-point bilinear table).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .db_pre import cv2_resize_linear_u8

BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------
# network
# ------------------------------------------------------------------------------------------------
def _cbr(sd, p_conv, p_bn, x, stride=1, pad=1):
    x = F.conv2d(x, sd[p_conv + ".weight"], sd[p_conv + ".bias"], stride, pad)
    x = F.batch_norm(x, sd[p_bn + ".running_mean"], sd[p_bn + ".running_var"], sd[p_bn + ".weight"], sd[p_bn + ".bias"],
                     False, 0.0, BN_EPS)
    return F.relu(x)


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """x [T, B, I] -> h [T, B, H]; gate order i, f, g, o (torch.nn.LSTM)."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H, dtype=x.dtype)
    c = torch.zeros(B, H, dtype=x.dtype)
    out = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out)


def bilstm(sd, p, x):
    """BidirectionalLSTM: nn.LSTM(bidirectional=True) + Linear on the concatenated hidden states."""
    fw = lstm_direction(x, sd[p + ".rnn.weight_ih_l0"], sd[p + ".rnn.weight_hh_l0"], sd[p + ".rnn.bias_ih_l0"],
                        sd[p + ".rnn.bias_hh_l0"])
    bw = lstm_direction(x, sd[p + ".rnn.weight_ih_l0_reverse"], sd[p + ".rnn.weight_hh_l0_reverse"],
                        sd[p + ".rnn.bias_ih_l0_reverse"], sd[p + ".rnn.bias_hh_l0_reverse"], reverse=True)
    rec = torch.cat([fw, bw], dim=2)
    T, B, Hh = rec.shape
    out = rec.reshape(T * B, Hh) @ sd[p + ".embedding.weight"].t() + sd[p + ".embedding.bias"]
    return out.view(T, B, -1)


def crnn_features_fp32(sd, x):
    """conv stack: x fp32 [B,3,32,W] in [0,1] -> [W/4, B, 512]."""
    g = x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140
    f = _cbr(sd, "conv0.0", "conv0.1", g)
    f = F.max_pool2d(f, (2, 2), (2, 2))
    f = _cbr(sd, "conv1.0", "conv1.1", f)
    f = F.max_pool2d(f, (2, 2), (2, 2))
    f = _cbr(sd, "conv2.0", "conv2.1", f)
    f = _cbr(sd, "conv2.3", "conv2.4", f)
    f = F.max_pool2d(f, (2, 1), (2, 1))
    f = _cbr(sd, "conv3.0", "conv3.1", f)
    f = _cbr(sd, "conv3.3", "conv3.4", f)
    f = F.max_pool2d(f, (2, 1), (2, 1))
    f = _cbr(sd, "conv4.0", "conv4.1", f, stride=(2, 1), pad=0)
    assert f.shape[2] == 1, "the height of conv must be 1"
    return f.squeeze(2).permute(2, 0, 1)


_NATIVE = {}


def bilstm_native(sd, p, x):
    """The same layer through torch's own LSTM operator -- what the reference runs (nn.LSTM inside BidirectionalLSTM,
    crnn/modeling_crnn.py:16-33); used by the timed CPU-baseline leg so that the baseline is not slowed by a Python loop.
    Equal to ``bilstm`` up to fp32 summation order (tests/test_oracle_db_net.py)."""
    key = (id(sd), p)
    m = _NATIVE.get(key)
    if m is None:
        w = sd[p + ".rnn.weight_hh_l0"]
        m = torch.nn.LSTM(sd[p + ".rnn.weight_ih_l0"].shape[1], w.shape[1], bidirectional=True)
        m.load_state_dict({k[len(p) + 5:]: v for k, v in sd.items() if k.startswith(p + ".rnn.")}, strict=True)
        m.eval()
        _NATIVE[key] = m
    with torch.no_grad():
        rec, _ = m(x)
    T, B, Hh = rec.shape
    out = rec.reshape(T * B, Hh) @ sd[p + ".embedding.weight"].t() + sd[p + ".embedding.bias"]
    return out.view(T, B, -1)


def crnn_forward_fp32(sd, x, native_lstm=False):
    """-> logits fp32 [B, W/4, 7644] (no softmax in the model)."""
    f = crnn_features_fp32(sd, x)
    rnn = bilstm_native if native_lstm else bilstm
    r = rnn(sd, "rnn.0", f)
    r = rnn(sd, "rnn.1", r)
    out = r @ sd["cls.weight"].t()
    return out.permute(1, 0, 2)


def ctc_greedy_ids(logits_or_ids):
    """argmax, collapse repeats, drop blank 0 (modeling_ocr_recognition.py:168-184) -> list of id lists."""
    a = np.asarray(logits_or_ids)
    ids = a if a.ndim == 2 else a.argmax(-1)
    out = []
    for row in ids:
        last, keep = 0, []
        for p in row.tolist():
            if p != last and p != 0:
                keep.append(p)
            last = p
        out.append(keep)
    return out


# ------------------------------------------------------------------------------------------------
# crop + resize
# ------------------------------------------------------------------------------------------------
def order_point(coor):
    """ocr_common_utils.py:287-304."""
    arr = np.array(coor).reshape([4, 2])
    sum_ = np.sum(arr, 0)
    centroid = sum_ / arr.shape[0]
    theta = np.arctan2(arr[:, 1] - centroid[1], arr[:, 0] - centroid[0])
    sort_points = arr[np.argsort(theta)]
    sort_points = sort_points.reshape([4, -1])
    if sort_points[0][0] > centroid[0]:
        sort_points = np.concatenate([sort_points[3:], sort_points[:3]])
    return sort_points.reshape([4, 2]).astype("float32")


def crop_geometry(position):
    """The pure-Python part of crop_image (ocr_common_utils.py:224-262): -> (src corners f32 [4,2], dst corners f32
    [4,2], out_w int, out_h int)."""
    def distance(x1, y1, x2, y2):
        return math.sqrt(pow(x1 - x2, 2) + pow(y1 - y2, 2))
    position = np.asarray(position).tolist()
    for i in range(4):
        for j in range(i + 1, 4):
            if position[i][0] > position[j][0]:
                position[i], position[j] = position[j], position[i]
    if position[0][1] > position[1][1]:
        position[0], position[1] = position[1], position[0]
    if position[2][1] > position[3][1]:
        position[2], position[3] = position[3], position[2]
    x1, y1 = position[0]
    x2, y2 = position[2]
    x3, y3 = position[3]
    x4, y4 = position[1]
    corners = np.zeros((4, 2), np.float32)
    corners[0] = [x1, y1]
    corners[1] = [x2, y2]
    corners[2] = [x4, y4]
    corners[3] = [x3, y3]
    img_width = distance((x1 + x4) / 2, (y1 + y4) / 2, (x2 + x3) / 2, (y2 + y3) / 2)
    img_height = distance((x1 + x2) / 2, (y1 + y2) / 2, (x4 + x3) / 2, (y4 + y3) / 2)
    corners_trans = np.zeros((4, 2), np.float32)
    corners_trans[0] = [0, 0]
    corners_trans[1] = [img_width - 1, 0]
    corners_trans[2] = [0, img_height - 1]
    corners_trans[3] = [img_width - 1, img_height - 1]
    return corners, corners_trans, int(img_width), int(img_height)


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform: the 8x8 system of imgwarp.cpp solved in float64 -> 3x3 (c22 = 1)."""
    a = np.zeros((8, 8), np.float64)
    b = np.zeros(8, np.float64)
    for i in range(4):
        a[i, 0] = a[i + 4, 3] = src[i][0]
        a[i, 1] = a[i + 4, 4] = src[i][1]
        a[i, 2] = a[i + 4, 5] = 1
        a[i, 6] = -float(src[i][0]) * float(dst[i][0])
        a[i, 7] = -float(src[i][1]) * float(dst[i][0])
        a[i + 4, 6] = -float(src[i][0]) * float(dst[i][1])
        a[i + 4, 7] = -float(src[i][1]) * float(dst[i][1])
        b[i] = dst[i][0]
        b[i + 4] = dst[i][1]
    x = np.linalg.solve(a, b)
    return np.append(x, 1.0).reshape(3, 3)


def _bilinear_tab():
    """OpenCV's BilinearTab_i (imgwarp.cpp initInterTab2D): 32x32 sub-pixel positions x 4 int16 weights scaled by
    2^15.  For the bilinear kernel every weight (32-ay)(32-ax)/1024 * 32768 is an exact multiple of 32, so the
    table sums to 32768 without the rounding fix-up; the one saturating entry (ay = ax = 0: 32768 -> 32767, +1 on
    another tap) still evaluates to the centre pixel exactly, so it is represented as a unit weight."""
    ay = np.arange(32)[:, None]
    ax = np.arange(32)[None, :]
    return np.stack([(32 - ay) * (32 - ax), (32 - ay) * ax, ay * (32 - ax), ay * ax], axis=-1).astype(np.int64) * 32


_TAB = None


def warp_perspective_u8(img, M, out_w, out_h):
    """cv2.warpPerspective(img, M, (out_w, out_h)) for uint8 HxWx3, INTER_LINEAR, BORDER_CONSTANT(0)."""
    global _TAB
    if _TAB is None:
        _TAB = _bilinear_tab()
    h, w, ch = img.shape
    Mi = np.linalg.inv(np.asarray(M, np.float64))
    out = np.zeros((out_h, out_w, ch), np.uint8)
    if out_w <= 0 or out_h <= 0:
        return out
    xs = np.arange(out_w, dtype=np.float64)
    src = img.astype(np.int64)
    for y in range(out_h):
        X0 = Mi[0, 0] * xs + Mi[0, 1] * y + Mi[0, 2]
        Y0 = Mi[1, 0] * xs + Mi[1, 1] * y + Mi[1, 2]
        W0 = Mi[2, 0] * xs + Mi[2, 1] * y + Mi[2, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            Wn = np.where(W0 != 0, 32.0 / W0, 0.0)
        fX = np.clip(X0 * Wn, -2147483648.0, 2147483647.0)
        fY = np.clip(Y0 * Wn, -2147483648.0, 2147483647.0)
        X = np.rint(fX).astype(np.int64)
        Y = np.rint(fY).astype(np.int64)
        sx, sy = X >> 5, Y >> 5
        ax, ay = (X & 31), (Y & 31)
        wts = _TAB[ay, ax]                                   # [out_w, 4]
        acc = np.zeros((out_w, ch), np.int64)
        for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            pix = np.zeros((out_w, ch), np.int64)
            pix[ok] = src[yy[ok], xx[ok]]
            acc += pix * wts[:, k:k + 1]
        out[y] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out


def crop_image(img, position):
    """ocr_common_utils.py:214-266."""
    src, dst, ow, oh = crop_geometry(position)
    M = get_perspective_transform(src, dst)
    return warp_perspective_u8(img, M, ow, oh)


def keepratio_resize(img, target_height=32, target_width=640):
    """processor_ocr_recognition.py:44-62."""
    cur_ratio = img.shape[1] / float(img.shape[0])
    if cur_ratio > float(target_width) / target_height:
        cur_h, cur_w = target_height, target_width
    else:
        cur_h, cur_w = target_height, int(target_height * cur_ratio)
    mask = np.zeros([target_height, target_width, 3]).astype(np.uint8)
    if cur_w > 0:
        r = cv2_resize_linear_u8(np.ascontiguousarray(img), cur_w, cur_h)
        mask[:r.shape[0], :r.shape[1], :] = r
    return mask


def rec_preprocess(crop, target_height=32, target_width=640):
    """crop uint8 HxWx3 -> fp32 [1,3,32,640] (processor_ocr_recognition.py:94-113, no chunking)."""
    img = keepratio_resize(crop, target_height, target_width)
    data = torch.FloatTensor(img).view(1, target_height, target_width, 3) / 255.
    return data.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------
# PP-OCR flavour: CTCLabelDecode (ocr_rec_pp/rec_postprocess.py:126-195), pinned by tests/golden/ctc_decode.*
# ------------------------------------------------------------------------------------------------
def ctc_label_decode(preds, characters, reverse=False):
    """preds float [B,T,C]; characters = dict lines (+ ' '); -> [(text, conf)] exactly like CTCLabelDecode.__call__."""
    import re
    table = ["blank"] + list(characters)
    preds = np.asarray(preds)
    idx = preds.argmax(axis=2)
    prob = preds.max(axis=2)
    out = []
    for b in range(len(idx)):
        sel = np.ones(len(idx[b]), dtype=bool)
        sel[1:] = idx[b][1:] != idx[b][:-1]
        sel &= idx[b] != 0
        chars = [table[t] for t in idx[b][sel]]
        conf = prob[b][sel]
        if len(conf) == 0:
            conf = [0]
        text = "".join(chars)
        if reverse:
            pred_re, cur = [], ""
            for c in text:
                if not bool(re.search("[a-zA-Z0-9 :*./%+-]", c)):
                    if cur != "":
                        pred_re.append(cur)
                    pred_re.append(c)
                    cur = ""
                else:
                    cur += c
            if cur != "":
                pred_re.append(cur)
            text = "".join(pred_re[::-1])
        out.append((text, np.mean(conf).tolist()))
    return out
