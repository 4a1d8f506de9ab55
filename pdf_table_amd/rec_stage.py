"""Batched text-line recognition stage: quad geometry on the host (vectorised numpy), everything else on the GPU.

Replaces the reference's per-line Python loop (ocr_system_task.py:296-336 / modeling_ocr_pdf.py:269-302):
``order_point`` -> ``crop_image`` (cv2 perspective warp) -> ``OCRRecognitionPreprocessor`` -> ``CRNN`` ->
arg-max -> CTC collapse.  Here the host only computes, for all quads of a page batch at once, the ordered corners, the
crop size and the inverse perspective matrix (an 8x8 solve per quad); cropping, resizing, the network and the
arg-max run in one ``pt_rec_forward`` call, and only ``int32 [lines, 160]`` token ids come back.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import lib as L
from .engine import REC_LINE_DTYPE, HipEngine

__all__ = ["order_points", "crop_geometry", "perspective_inverse", "build_lines", "ctc_collapse", "RecStage",
           "synthetic_vocab"]


def order_points(quads: np.ndarray) -> np.ndarray:
    """OcrCommonUtils.order_point (utils/ocr/ocr_common_utils.py:287-304) for [n, 8] / [n, 4, 2] quads at once:
    sort the corners by angle around the centroid; rotate so that the first one is left of the centroid."""
    a = np.asarray(quads).reshape(-1, 4, 2)
    cen = a.sum(1) / 4                                            # same dtype rules as np.sum(arr, 0) / 4
    theta = np.arctan2(a[:, :, 1] - cen[:, None, 1], a[:, :, 0] - cen[:, None, 0])
    idx = np.argsort(theta, axis=1)                               # default kind, like the reference
    sp = np.take_along_axis(a, idx[:, :, None], axis=1)
    rot = sp[:, 0, 0] > cen[:, 0]
    sp[rot] = np.concatenate([sp[rot][:, 3:], sp[rot][:, :3]], axis=1)
    return sp.astype("float32")


def _cswap(p, i, j, cond):
    pi, pj = p[:, i].copy(), p[:, j].copy()
    p[:, i] = np.where(cond[:, None], pj, pi)
    p[:, j] = np.where(cond[:, None], pi, pj)


def crop_geometry(pts: np.ndarray):
    """The pure-Python part of crop_image (ocr_common_utils.py:224-262), vectorised over quads.
    pts f32 [n,4,2] -> (src f32 [n,4,2], dst f32 [n,4,2], out_w int [n], out_h int [n])."""
    p = np.asarray(pts, dtype=np.float32).reshape(-1, 4, 2).astype(np.float64)   # .tolist() gives python floats
    for i in range(4):                       # the reference's exchange sort on x (fixed compare-exchange sequence)
        for j in range(i + 1, 4):
            _cswap(p, i, j, p[:, i, 0] > p[:, j, 0])
    _cswap(p, 0, 1, p[:, 0, 1] > p[:, 1, 1])
    _cswap(p, 2, 3, p[:, 2, 1] > p[:, 3, 1])
    x1, y1 = p[:, 0, 0], p[:, 0, 1]
    x2, y2 = p[:, 2, 0], p[:, 2, 1]
    x3, y3 = p[:, 3, 0], p[:, 3, 1]
    x4, y4 = p[:, 1, 0], p[:, 1, 1]
    src = np.stack([np.stack([x1, y1], 1), np.stack([x2, y2], 1), np.stack([x4, y4], 1), np.stack([x3, y3], 1)], 1)
    src = src.astype(np.float32)

    def dist(ax, ay, bx, by):
        return np.sqrt((ax - bx) ** 2 + (ay - by) ** 2)
    iw = dist((x1 + x4) / 2, (y1 + y4) / 2, (x2 + x3) / 2, (y2 + y3) / 2)
    ih = dist((x1 + x2) / 2, (y1 + y2) / 2, (x4 + x3) / 2, (y4 + y3) / 2)
    z = np.zeros_like(iw)
    dst = np.stack([np.stack([z, z], 1), np.stack([iw - 1, z], 1), np.stack([z, ih - 1], 1), np.stack([iw - 1, ih - 1], 1)], 1)
    return src, dst.astype(np.float32), iw.astype(np.int64), ih.astype(np.int64)      # int() truncates


def perspective_inverse(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getPerspectiveTransform(src, dst) (8x8 system, float64) and its 3x3 inverse, batched: -> [n, 9] float64.
    Degenerate quads (singular system) yield all-zero matrices (their crops are empty or constant)."""
    n = src.shape[0]
    s = src.astype(np.float64)
    d = dst.astype(np.float64)
    a = np.zeros((n, 8, 8))
    b = np.zeros((n, 8))
    for i in range(4):
        a[:, i, 0] = a[:, i + 4, 3] = s[:, i, 0]
        a[:, i, 1] = a[:, i + 4, 4] = s[:, i, 1]
        a[:, i, 2] = a[:, i + 4, 5] = 1
        a[:, i, 6] = -s[:, i, 0] * d[:, i, 0]
        a[:, i, 7] = -s[:, i, 1] * d[:, i, 0]
        a[:, i + 4, 6] = -s[:, i, 0] * d[:, i, 1]
        a[:, i + 4, 7] = -s[:, i, 1] * d[:, i, 1]
        b[:, i] = d[:, i, 0]
        b[:, i + 4] = d[:, i, 1]
    out = np.zeros((n, 9))
    # batched LAPACK (gesv / getri per matrix, i.e. the same routine a per-quad call runs); singular systems are
    # masked out first so that one degenerate quad does not poison the batch
    with np.errstate(all="ignore"):
        ok = np.abs(np.linalg.det(a)) > 0
    if ok.any():
        x = np.linalg.solve(a[ok], b[ok][:, :, None])[:, :, 0]
        m = np.concatenate([x, np.ones((x.shape[0], 1))], 1).reshape(-1, 3, 3)
        with np.errstate(all="ignore"):
            good = np.abs(np.linalg.det(m)) > 0
        inv = np.zeros_like(m)
        inv[good] = np.linalg.inv(m[good])
        out[ok] = inv.reshape(-1, 9)
    return out


def build_lines(boxes_per_page: Sequence[np.ndarray]) -> np.ndarray:
    """Detection boxes ([k,8] per page, source pixels) -> pt_rec_line records for every line of the batch."""
    counts = [len(b) for b in boxes_per_page]
    tot = sum(counts)
    lines = np.zeros(tot, dtype=REC_LINE_DTYPE)
    if tot == 0:
        return lines
    allb = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 8) for b in boxes_per_page if len(b)], 0)
    pts = order_points(allb)
    src, dst, ow, oh = crop_geometry(pts)
    lines["minv"] = perspective_inverse(src, dst)
    lines["page"] = np.repeat(np.arange(len(counts)), counts)
    lines["crop_w"] = np.clip(ow, 0, 1 << 20)
    lines["crop_h"] = np.clip(oh, 0, 1 << 20)
    return lines


def ctc_collapse(ids: np.ndarray) -> List[List[int]]:
    """OCRRecognition.postprocess after the arg-max (modeling_ocr_recognition.py:172-182): drop repeats, drop 0."""
    ids = np.asarray(ids)
    prev = np.concatenate([np.zeros((ids.shape[0], 1), ids.dtype), ids[:, :-1]], 1)
    keep = (ids != prev) & (ids != 0)
    return [row[k].tolist() for row, k in zip(ids, keep)]


def synthetic_vocab(n: int = L.PT_REC_NCLS - 1) -> List[str]:
    """Stand-in for vocab.txt (not available offline): n distinct CJK code points."""
    return [chr(0x4E00 + i) for i in range(n)]


class RecStage:
    def __init__(self, eng: HipEngine, vocab: Optional[Sequence[str]] = None, recognizer: str = "CRNN"):
        self.eng = eng
        self.recognizer = recognizer
        chunking = recognizer == "ConvNextViT"
        vocab = list(vocab) if vocab is not None else synthetic_vocab(L.PT_CVIT_NCLS - 2 if chunking else L.PT_REC_NCLS - 1)
        # labelMapping starts at 1 for CRNN and at 2 for the chunking ConvNextViT: modeling_ocr_recognition.py:119-132,
        # processor_ocr_recognition.py:131-145 (class 1 has no entry there either)
        first = 2 if chunking else 1
        self.label = {i + first: ch for i, ch in enumerate(vocab)}

    def ids(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray]):
        lines = build_lines(boxes_per_page)
        if self.recognizer == "ConvNextViT":
            ids, _ = self.eng.rec_cvit_forward(pages, lines)
        else:
            ids, _ = self.eng.rec_forward(pages, lines, want_maxlogit=False)
        return ids, lines

    def start(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray]):
        """device half (asynchronous on the current stream): quad geometry on the host, one pt_rec_forward, and the token ids
        on their way to pinned host memory behind it -- finish() waits for THAT copy, not for whatever else has been queued
        on the stream since (a pipelined caller queues the next batch before it collects this one)"""
        ids, lines = self.ids(pages, boxes_per_page)
        host = done = None
        degenerate = ()
        if len(lines):
            host = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
            host.copy_(ids, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            # a crop more than 32 times as high as wide resizes to width 0: cv2.resize raises in the reference and the system path turns that
            # into '' (ocr_system_task.py:275-283); the engine decodes an all-padding line there, so finish() blanks these lines
            bad = (lines["crop_w"].astype(np.int64) * 32 < lines["crop_h"]) | (lines["crop_w"] <= 0) | (lines["crop_h"] <= 0)
            if bad.any():
                degenerate = np.nonzero(bad)[0]
        # the blank-out list travels IN the state: a table keyed by id(host) outlived a failed batch (finish() raising before the pop) and could
        # match a later batch's buffer once CPython reused the id
        return ids, len(lines), [len(b) for b in boxes_per_page], host, done, degenerate

    def finish(self, state) -> List[List[str]]:
        """host half: wait for the ids of that start(), CTC collapse, vocabulary"""
        ids, nl, per_page, host, done, degenerate = state
        toks = []
        if nl:
            done.synchronize()
            toks = ctc_collapse(host.numpy())
        self.eng.check()          # the batch has executed: surface device-side failures of it
        texts = ["".join(self.label.get(t, "") for t in row) for row in toks]
        for i in degenerate:
            texts[int(i)] = ""
        out, o = [], 0
        for k in per_page:
            out.append(texts[o:o + k])
            o += k
        return out

    def __call__(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray]) -> List[List[str]]:
        return self.finish(self.start(pages, boxes_per_page))
