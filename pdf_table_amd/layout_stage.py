"""Batched layout-detection stage (PicoDet): pre-process + network + candidate compaction on the GPU, the per-class hard
NMS on the host over the few candidates that can pass the score threshold.

Replaces the reference's per-page chain ``OCRPicodetPreProcessor.__call__`` -> ONNX run -> ``OCRPicodetPostProcessor.__call__``
(picodet/processor_picodet.py:72-113, 184-298; ocr_layout_task.py:70-157).  The device hands back, per page, only the
anchors whose best class score exceeds a threshold just below ``score_threshold`` (the reference's post-processor drops
everything else in ``probs > self.score_threshold``, :252); their box distributions are decoded here with the
reference's numpy arithmetic (float32 soft-max, float64 boxes), so the result equals the oracle's on the same head values.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Dict, List, Sequence

import numpy as np
import torch
from .streams import shared_stream

from . import lib as L
from .engine import HipEngine

__all__ = ["PicodetConfig", "LayoutStage", "hard_nms", "layout_tables", "LAYOUT_LABELS"]

LAYOUT_LABELS = {   # configuration_picodet.py:66-104
    "ch": ["text", "title", "figure", "figure_caption", "table", "table_caption", "header", "footer", "reference", "equation"],
    "en": ["text", "title", "list", "table", "figure"],
    "table": ["table"],
}


@dataclass
class PicodetConfig:
    """configuration_picodet.py:41-113 (inference fields)"""
    task_type: str = "en"
    model_name: str = "picodet"
    model_path: str = ""
    backbone: str = "LCNet"
    img_height: int = 800
    img_width: int = 608
    score_threshold: float = 0.5
    nms_threshold: float = 0.5
    strides: Sequence[int] = (8, 16, 32, 64)
    nms_top_k: int = 1000
    keep_top_k: int = 100
    labels: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.labels = list(LAYOUT_LABELS.get(self.task_type, LAYOUT_LABELS["ch"]))
        self.id2label = dict(enumerate(self.labels))


def _iou_of(b0, b1, eps=1e-5):
    lt = np.maximum(b0[..., :2], b1[..., :2])
    rb = np.minimum(b0[..., 2:], b1[..., 2:])
    hw = np.clip(rb - lt, 0.0, None)
    inter = hw[..., 0] * hw[..., 1]
    a0 = np.clip(b0[..., 2:] - b0[..., :2], 0.0, None)
    a1 = np.clip(b1[..., 2:] - b1[..., :2], 0.0, None)
    return inter / (a0[..., 0] * a0[..., 1] + a1[..., 0] * a1[..., 1] - inter + eps)


def hard_nms(box_scores: np.ndarray, iou_threshold: float, top_k: int = -1, candidate_size: int = 200) -> np.ndarray:
    """processor_picodet.py:301-330: greedy NMS over the candidate_size best boxes, at most top_k kept"""
    scores, boxes = box_scores[:, -1], box_scores[:, :-1]
    picked = []
    idx = np.argsort(scores)[-candidate_size:]
    while len(idx) > 0:
        cur = idx[-1]
        picked.append(cur)
        if 0 < top_k == len(picked) or len(idx) == 1:
            break
        idx = idx[:-1]
        idx = idx[_iou_of(boxes[idx, :], boxes[cur:cur + 1, :]) <= iou_threshold]
    return box_scores[picked, :]


def _softmax_f32(x: np.ndarray) -> np.ndarray:
    """scipy.special.softmax(x, axis=1) on float32 input: exp(x - max) / sum, in float32"""
    e = np.exp(x - np.max(x, axis=1, keepdims=True))
    return e / np.sum(e, axis=1, keepdims=True)


def layout_tables(layout_result: List[Dict], label: str = "table", score_threshold: float = 0.2) -> List[Dict]:
    """TableProcessUtils.get_layout_by_type (pdf_table/table_common.py:1287-1300): regions of one label, top to bottom"""
    res = [it for it in layout_result if it["label"].lower() == label.lower() and it["score"] >= score_threshold]
    res.sort(key=lambda x: x["bbox"][1])
    return res


class LayoutStage:
    def __init__(self, eng: HipEngine, config: PicodetConfig = None, max_cands: int = 2048, precision: Optional[int] = None):
        """``precision``: an L.PT_PRECISION_* for THIS stage's network, whatever the engine's is (None: the engine's).  The layout net is the
        pipeline's smallest (3 % of a step) and its output is a DECISION the table stage inherits -- the crop of every table at the box's ROUNDED
        coordinates (ocr_system_task.py:184-198): in a 16-bit mode an edge that rounds one pixel differently is a different input to the table
        net.  ``precision=L.PT_PRECISION_BF16X3`` runs just this stage in the pair mode (its blob must then be a bf16 blob with the pair tiles,
        also on an engine that otherwise computes in PT_PRECISION_F16: blobs carry their format per model)."""
        self.eng = eng
        self.precision = precision
        self.config = config or PicodetConfig()
        self.max_cands = max_cands
        self._copy_stream = None
        self._pinned = [None, None]          # two host landing buffers: a forward() may be queued while the previous finish() still reads
        self._turn = 0

    def forward(self, pages: torch.Tensor):
        """device half (asynchronous): network + candidate compaction on the current stream, then the D2H of the
        records on a copy stream behind an event -- finish() waits for that copy only, not for whatever else has
        been queued on the compute stream meanwhile"""
        cfg = self.config
        # the stage's precision is host state read when a call is QUEUED: the launches below carry it, later calls do not; the scope holds the
        # engine's lock, so no other host thread can queue a call in this stage's precision meanwhile
        with self.eng.precision_scope(self.precision):
            counts, cands = self.eng.layout_forward(pages, cfg.img_height, cfg.img_width, len(cfg.labels),
                                                    thr_lo=cfg.score_threshold - 1e-3, max_cands=self.max_cands)
        n = counts.shape[0]
        if self._copy_stream is None:
            self._copy_stream = shared_stream(counts.device, "layout_copy")
        slot = self._turn
        self._turn ^= 1
        if self._pinned[slot] is None or self._pinned[slot][0].shape[0] < n:
            self._pinned[slot] = (torch.empty((n,), dtype=torch.int32).pin_memory(),
                                  torch.empty((n, self.max_cands, L.PT_LAYOUT_CAND_FLOATS), dtype=torch.float32).pin_memory())
        host = self._pinned[slot]
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            host[0][:n].copy_(counts, non_blocking=True)
            host[1][:n].copy_(cands, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        counts.record_stream(self._copy_stream)
        cands.record_stream(self._copy_stream)
        return n, (done, host)

    def decode_outputs(self, scores: Sequence[np.ndarray], dists: Sequence[np.ndarray], org_shape) -> List[Dict]:
        """One page's outputs of a PicoDet ONNX export -- per level class PROBABILITIES [A_l, ncls] and box-distribution logits [A_l, 32], the
        two halves ``OcrLayoutTask.get_onnx_output_dict`` splits (ocr_layout_task.py:159-175) -- -> the same result list: anchors that can pass
        the score threshold become the candidate records decode_page() reads from the engine's own graph"""
        cfg = self.config
        ncls = len(cfg.labels)
        recs = []
        for l, (sc, bd) in enumerate(zip(scores, dists)):
            sc, bd = np.asarray(sc, np.float32).reshape(-1, ncls), np.asarray(bd, np.float32).reshape(-1, 32)
            keep = np.nonzero(sc.max(1) > cfg.score_threshold - 1e-3)[0]
            r = np.zeros((len(keep), 2 + ncls + 32), np.float32)
            r[:, 0] = np.full(len(keep), l, np.int32).view(np.float32)
            r[:, 1] = keep.astype(np.int32).view(np.float32)
            r[:, 2:2 + ncls], r[:, 2 + ncls:] = sc[keep], bd[keep]
            recs.append(r)
        return self.decode_page(np.concatenate(recs, 0) if recs else np.zeros((0, 2 + ncls + 32), np.float32), org_shape, probs=True)

    def decode_page(self, rec: np.ndarray, org_shape, probs: bool = False) -> List[Dict]:
        """candidate records [k, 48] of one page -> the `bboxs` list of OCRPicodetPostProcessor.__call__ (probs: the class columns already
        went through the sigmoid, as in an ONNX export's outputs)"""
        cfg = self.config
        ncls = len(cfg.labels)
        if len(rec) == 0:
            return []
        # the device appends candidates with an atomic counter: put them in the reference's dense order (level, anchor) so that
        # ties in the argsorts below break as they do there and the result does not depend on the order of arrival
        rec = rec[np.lexsort((rec[:, 1].view(np.int32), rec[:, 0].view(np.int32)))]
        level = rec[:, 0].view(np.int32)
        anchor = rec[:, 1].view(np.int32)
        logits = rec[:, 2:2 + ncls]
        scores = np.ascontiguousarray(logits, dtype=np.float32) if probs else \
            torch.sigmoid(torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32))).numpy()   # F.sigmoid of forward_eval
        reg = rec[:, 2 + ncls:2 + ncls + 32].astype(np.float32)
        reg_max = 7
        th, tw = cfg.img_height, cfg.img_width
        boxes_l, scores_l = [], []
        for l, stride in enumerate(cfg.strides):
            m = level == l
            if not m.any():
                continue
            a, sc, bd = anchor[m], scores[m], reg[m]
            fm_w = int(np.arange(tw / stride).shape[0])                      # np.arange(fm_w) of a float count (:213-215)
            ct_row = (a // fm_w + 0.5) * stride
            ct_col = (a % fm_w + 0.5) * stride
            center = np.stack((ct_col, ct_row, ct_col, ct_row), axis=1)
            dist = _softmax_f32(bd.reshape(-1, reg_max + 1)) * np.expand_dims(np.arange(reg_max + 1), 0)
            dist = np.sum(dist, axis=1).reshape(-1, 4) * stride
            order = np.argsort(sc.max(axis=1))[::-1][:cfg.nms_top_k]         # the level's top-K rule (:225-229)
            boxes_l.append(center[order] + [-1, -1, 1, 1] * dist[order])
            scores_l.append(sc[order])
        bboxes = np.concatenate(boxes_l, 0)
        conf = np.concatenate(scores_l, 0)
        # per-class greedy NMS: candidate order by numpy (its tie order), the greedy loop on the host in C++ (pt_hard_nms)
        groups, orders, labels_c = [], [], []
        base = 0
        for c in range(ncls):
            probs = conf[:, c]
            mask = probs > cfg.score_threshold
            if not mask.any():
                continue
            bs = np.concatenate([bboxes[mask], probs[mask].reshape(-1, 1)], axis=1)
            groups.append(bs)
            orders.append(np.argsort(bs[:, -1])[-200:] + base)
            labels_c.append(c)
            base += len(bs)
        picked, labels = [], []
        if groups:
            allb = np.ascontiguousarray(np.concatenate(groups, 0), dtype=np.float64)
            order = np.ascontiguousarray(np.concatenate(orders), dtype=np.int64)
            goff = np.zeros(len(groups) + 1, np.int64)
            goff[1:] = np.cumsum([len(o) for o in orders])
            out = np.empty(len(order), np.int64)
            npk = np.zeros(len(groups), np.int32)
            L.check(L.load().pt_hard_nms(allb.ctypes.data, order.ctypes.data, goff.ctypes.data, len(groups),
                                             float(cfg.nms_threshold), int(cfg.keep_top_k), out.ctypes.data, npk.ctypes.data),
                    "pt_hard_nms")
            for g, c in enumerate(labels_c):
                sel = out[goff[g]:goff[g] + npk[g]]
                picked.append(allb[sel])
                labels.extend([c] * len(sel))
        if not picked:
            return []
        pb = np.concatenate(picked)
        oh, ow = np.float32(org_shape[0]), np.float32(org_shape[1])
        b = pb[:, :4]
        x = b[:, [0, 2, 0, 2]]
        y = b[:, [1, 3, 3, 1]]
        xy = np.stack([x.min(1), y.min(1), x.max(1), y.max(1)], 1)
        xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, ow)        # warp_boxes clips to the ORIGINAL size in the resized frame (:158-159)
        xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, oh)
        pb[:, :4] = xy.astype(np.float32)
        sf = np.array([float(th) / org_shape[0], float(tw) / org_shape[1]], dtype=np.float32)
        pb[:, :4] /= np.concatenate([sf[::-1], sf[::-1]])
        return [{"bbox": pb[i, :4].copy(), "label": cfg.id2label[int(c)], "score": pb[i, 4], "category_id": int(c)}
                for i, c in enumerate(labels)]

    def __call__(self, pages: torch.Tensor) -> List[List[Dict]]:
        n, ticket = self.forward(pages)
        return self.finish(n, ticket, tuple(pages.shape[1:3]))

    def finish(self, n: int, ticket, org) -> List[List[Dict]]:
        """host half: wait for the candidate copy of that forward(), decode + NMS per page.  At most one newer forward() may
        have been issued in between (two landing buffers)."""
        done, host = ticket
        done.synchronize()
        counts = host[0][:n].numpy().copy()
        if (counts > self.max_cands).any():
            raise RuntimeError(f"layout: {int(counts.max())} candidate anchors on a page exceed max_cands={self.max_cands}")
        kmax = int(counts.max()) if len(counts) else 0
        rec = host[1][:n, :max(kmax, 1)].numpy()
        return [self.decode_page(rec[i, :counts[i]], org) for i in range(len(counts))]
