"""Batched DB text-detection stage: device forward + bitmap, host contour candidates, device box scores,
host finalize.  This is the batched replacement of the reference's per-image loop
(ocr_detection_task.py:89-141 + db_pp/processor_ocr_db_pp.py:291-342), which runs batch 1, synchronously.

Host steps are C++ inside libpdftable_hip.so and release the GIL (ctypes), so a thread pool spreads pages
over the host cores while the GPU works on the next micro-batch.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import ctypes as C

import numpy as np
import torch
from .streams import shared_stream

from . import engine as E
from . import lib as L

__all__ = ["DetConfig", "DetStage", "filter_tag_det_res", "sort_boxes_reading_order"]


@dataclass
class DetConfig:
    """Defaults follow the reference configs (SURVEY.md section 5 constants)."""
    flavour: str = "db_pp"          # "db_pp": PPOcr pre/post; "db": torch DB pre/post (ocr_detection_task.py:35-56)
    thresh: float = 0.3             # DbPPConfig.thresh; the CLI passes 0.2 (entity/common_entity.py:269)
    box_thresh: float = 0.6         # db_pp/configuration_db_pp.py:45-60
    unclip_ratio: float = 1.5
    max_candidates: int = 1000
    min_size: float = 3.0
    use_dilation: bool = False

    def resolved(self):
        if self.flavour == "db":    # hard-coded constants of db_net/ocr_detection_utils.py:183-196
            return DetConfig("db", self.thresh, 0.3, 1.5, 1000, 3.0, False)
        return self

    @property
    def pre(self):
        return L.PT_DET_PRE_DB_TORCH if self.flavour == "db" else L.PT_DET_PRE_DB_PP

    @property
    def post(self):
        return L.PT_DET_POST_DB_TORCH if self.flavour == "db" else L.PT_DET_POST_DB_PP


def filter_tag_det_res(dt_boxes: np.ndarray, img_h: int, img_w: int) -> np.ndarray:
    """order TL,TR,BR,BL + clip + drop small (processor_ocr_db_pp.py:344-386), vectorised.  int boxes [n,4,2]."""
    if len(dt_boxes) == 0:
        return np.zeros((0, 4, 2), dtype=np.float32)
    b = dt_boxes.reshape(-1, 4, 2).astype(np.float32)
    n = b.shape[0]
    ar = np.arange(n)[:, None]
    xs = b[ar, np.argsort(b[:, :, 0], axis=1, kind="quicksort")]     # argsort default kind, as the reference
    left, right = xs[:, :2], xs[:, 2:]
    left = left[ar, np.argsort(left[:, :, 1], axis=1, kind="quicksort")]
    right = right[ar, np.argsort(right[:, :, 1], axis=1, kind="quicksort")]
    rect = np.stack([left[:, 0], right[:, 0], right[:, 1], left[:, 1]], axis=1)      # tl, tr, br, bl
    rect[:, :, 0] = np.trunc(np.minimum(np.maximum(rect[:, :, 0], 0), img_w - 1))
    rect[:, :, 1] = np.trunc(np.minimum(np.maximum(rect[:, :, 1], 0), img_h - 1))
    w = np.linalg.norm(rect[:, 0] - rect[:, 1], axis=1).astype(np.int64)
    h = np.linalg.norm(rect[:, 0] - rect[:, 3], axis=1).astype(np.int64)
    return rect[(w > 3) & (h > 3)]


def sort_boxes_reading_order(det_result: np.ndarray) -> np.ndarray:
    """OcrSystemTask.text_detection sort (ocr_system_task.py:159-162); stable like Python's sorted()."""
    d = np.asarray(det_result, dtype=np.float64).reshape(-1, 8)
    if len(d) == 0:
        return d
    key = np.array([0.01 * sum(r[::2]) / 4 + sum(r[1::2]) / 4 for r in d.tolist()])
    return d[np.argsort(key, kind="stable")]


class DetStage:
    def __init__(self, eng: "E.HipEngine", cfg: Optional[DetConfig] = None, workers: Optional[int] = None, net=None):
        """net: optional callable bf16 NHWC4 [n, H, W, 4] -> probability map f32 [n, H, W] on the device, in place of the
        engine's own detector networks (an imported ONNX graph run by onnx_exec.HipGraphExecutor); the pre-processing,
        bitmap, box scores and the host post-process stay the engine's"""
        self.eng = eng
        self.net = net
        self.cfg = (cfg or DetConfig()).resolved()
        self.workers = workers or max(1, min(32, (os.cpu_count() or 8)))     # threads of the batch host calls
        self._pin = {}
        self.side = None

    def close(self):
        pass

    # ---- device half -----------------------------------------------------------------------------------
    def forward(self, pages: torch.Tensor, slot: int = 0, early_copy: bool = False):
        """pages uint8 [n,h,w,3] on the GPU -> (prob f32 [n,nh,nw], bitmap i32 [n,nh,nw/32], event).
        Asynchronous on the current stream; outputs live in per-slot buffers so that the host half of batch
        k (on the side stream) can overlap the device half of batch k+1."""
        n, h, w, _ = pages.shape
        nh, nw = self.eng.det_plan(h, w, self.cfg.pre)
        key = ("out", slot)
        buf = self._pin.get(key)
        if buf is None or buf[0].shape != (n, nh, nw):
            dev = pages.device
            buf = (torch.empty((n, nh, nw), dtype=torch.float32, device=dev),
                   torch.empty((n, nh, nw // 32), dtype=torch.int32, device=dev))
            self._pin[key] = buf
        if self.net is not None:
            prob = self.net(self.eng.det_preprocess(pages, self.cfg.pre))
            if tuple(prob.shape) != (n, nh, nw):
                raise RuntimeError(f"the detector network returned a {tuple(prob.shape)} map for {n} pages of {nh}x{nw}")
            buf[0].copy_(prob)
            buf[1].copy_(self.eng.det_bitmap(buf[0], self.cfg.thresh, self.cfg.use_dilation))
            prob, bitmap = buf
        else:
            prob, bitmap = self.eng.det_forward(pages, self.cfg.pre, self.cfg.thresh, self.cfg.use_dilation,
                                                out_prob=buf[0], out_bitmap=buf[1])
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        if not early_copy:
            return prob, bitmap, ev
        # early_copy (callers that queue OTHER stages' work behind this detection before they call boxes(): predict_stream, the four-stage
        # bench loop): the bit-packed bitmap starts its way to pinned memory NOW, on the side stream behind the detector's last kernel --
        # queued before any later copy.  Issued from boxes() (a step later) the 7 MB copy could land behind the token-id / table-row copies
        # of batches queued since: a DMA engine serves its queue in order and those wait for kernels still to run; the host sat 15-50 ms
        # per batch in that one copy, bimodal from run to run (490 vs 580 pages/s through predict_stream).  Which hardware queue a stream's
        # work uses is the runtime's choice (4 by default for all streams of a process): a stream of its own for this copy, a third
        # high-priority stream for the box scores and one shared copy stream per engine all measured SLOWER (525-560 pages/s) than riding
        # on the box-score stream; a kernel copy into the mapped pinned buffer (pt_copy_bytes) never waits, but 7 MB of PCIe writes from
        # high-priority workgroups slowed the compute stream by 10 %.  A detection-only loop calls boxes(k-1) AFTER forward(k): there the
        # early copy of batch k would make the scores of batch k-1 wait for detection k, and nothing else is queued: early_copy=False.
        if self.side is None:
            self.side = shared_stream(pages.device, "det_side", priority=-1)
        hb = self._pinned(("bm", slot), bitmap.shape, torch.int32)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            hb.copy_(bitmap, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        return prob, bitmap, (ev, done, hb)

    def _pinned(self, key, shape, dtype):
        t = self._pin.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, pin_memory=True)
            self._pin[key] = t
        return t

    # ---- host + scoring half ---------------------------------------------------------------------------
    def boxes(self, prob: torch.Tensor, bitmap: torch.Tensor, src_hw, ev=None) -> List[np.ndarray]:
        """-> per page: boxes in source-page pixels; db_pp: float32 [k,8] after filter_tag_det_res,
        db: int32 [k,8] (reference return types, ocr_detection_task.py:126-141)."""
        cfg = self.cfg
        n, nh, nw = prob.shape
        if self.side is None:
            # HIGH priority: the bitmap copy and the box-score kernel are a few hundred microseconds of work the host WAITS for while the
            # compute stream holds one or two batches of queued kernels; on a normal-priority stream they could land on a hardware queue
            # shared with that backlog and sit behind it (measured: 20 ms vs 60 ms per 64 pages in this call, bimodal from run to run --
            # the difference between 490 and 580 pages/s through predict_stream)
            self.side = shared_stream(prob.device, "det_side", priority=-1)
        import time as _t
        tm = self.__dict__.setdefault("timing", {"copy": 0.0, "cand": 0.0, "score": 0.0, "final": 0.0})
        t0 = _t.perf_counter()
        if isinstance(ev, tuple):           # forward() already sent the bitmap on its way
            ev, done, hb = ev
            done.synchronize()
            if self.side is None:
                self.side = shared_stream(prob.device, "det_side", priority=-1)
        else:
            hb = self._pinned("bm", bitmap.shape, torch.int32)
            with torch.cuda.stream(self.side):
                if ev is not None:
                    self.side.wait_event(ev)
                else:
                    self.side.wait_stream(torch.cuda.current_stream())
                hb.copy_(bitmap, non_blocking=True)
            self.side.synchronize()
        bm = hb.numpy()
        t1 = _t.perf_counter()
        tm["copy"] += t1 - t0
        # contour candidates of all pages: one call, pages on threads inside the library (no Python per page)
        cand, counts = E.db_candidates_batch(bm, cfg.max_candidates, cfg.min_size, self.workers)
        t2 = _t.perf_counter()
        tm["cand"] += t2 - t1
        tot = int(counts.sum())
        cap = cand.shape[1]
        valid = np.arange(cap)[None, :] < counts[:, None]                 # [n, cap]
        scores = np.zeros((n, cap), np.float32)
        if tot:
            allb = self._pinned("boxes", (tot, 9), torch.float32)
            ab = allb.numpy()
            ab[:, 0] = np.repeat(np.arange(n, dtype=np.float32), counts)
            ab[:, 1:] = cand[valid]
            with torch.cuda.stream(self.side):
                if isinstance(ev, torch.cuda.Event):
                    self.side.wait_event(ev)
                scores[valid] = self.eng.det_box_scores(prob, allb.to(prob.device, non_blocking=True)).cpu().numpy()
        t3 = _t.perf_counter()
        tm["score"] += t3 - t2
        # score gate, unclip, second rectangle, rescale (+ filter_tag_det_res for the db_pp flavour), again in one call
        try:
            return self._finalize(cand, scores, counts, nh, nw, src_hw)
        finally:
            tm["final"] += _t.perf_counter() - t3

    def _finalize(self, cand, scores, counts, nh, nw, src_hw):
        cfg = self.cfg
        return E.db_finalize_batch(cand, scores, counts, (nh, nw), src_hw, cfg.box_thresh, cfg.unclip_ratio, cfg.min_size,
                                   cfg.post, filter_tag=cfg.flavour != "db", n_threads=self.workers)

    def __call__(self, pages: torch.Tensor) -> List[np.ndarray]:
        prob, bitmap, ev = self.forward(pages)
        return self.boxes(prob, bitmap, (pages.shape[1], pages.shape[2]), ev)
