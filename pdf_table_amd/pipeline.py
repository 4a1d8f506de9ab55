"""``OcrTablePipeline.predict()`` -- the batched page-inference façade named by BASELINE.json's north star.

The reference has no class of this name (SURVEY.md finding F1); its semantics are those of
``OcrSystemTask.__call__`` (src/pdftable/model/ocr_pdf/ocr_system_task.py:549-734) restricted to the vision path of
an *image* page: [text-line orientation vote -> 180 degree rotation + second detection (:441-491)] -> layout (PicoDet,
:203-215) -> text detection (:629, :148-166, incl. the reading-order sort) -> text recognition (:630, :296-336) ->
table structure (:184-199, Lore) on the layout regions labelled "table" (pdf_table/table_common.py:1287-1301; or on
``predict(..., table_boxes=...)``), each table's quads shifted into page pixels (``convert_table_sep_to_merge``,
table_common.py:1811-1825) -> cells + text -> HTML (``OcrTableToHtmlTask``).

What is different from the reference, by design: pages are processed as a batch (the reference is batch 1 and
synchronous), crops never leave the GPU -- in particular the table crop is NOT written as a JPEG and read back
(ocr_system_task.py:192-198: a lossy round trip through the file system) -- and a failed
page/line follows the reference's containment rules (a failed crop yields ``""``: ocr_system_task.py:275-283) but is
logged, never silent.
"""
from __future__ import annotations

import contextlib
import os
import logging
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from .streams import shared_stream

from .det_stage import DetConfig, DetStage, sort_boxes_reading_order
from .engine import HipEngine
from .ocr_detection_task import OcrDetectionTask, _read_image
from .layout_stage import layout_tables
from .ocr_layout_task import OcrLayoutTask
from .ocr_recognition_task import OcrRecognitionTask
from .rec_stage import order_points
from .trace_ranges import stage_range
from .ocr_table_structure_task import OcrTableStructureTask

__all__ = ["OcrTablePipeline", "PageResult"]

logger = logging.getLogger(__name__)


@dataclass
class PageResult:
    """Subset of OcrSystemModelOutput (ocr_output.py:25-61) the built stages fill."""
    det_result: np.ndarray                       # [n, 8] boxes in reading order (ocr_system_task.py:159-162)
    ocr_result: List[Dict] = field(default_factory=list)   # [{"index", "text", "bbox"}] like modeling_ocr_pdf.py:286-291
    layout_result: Optional[list] = None
    table_structure_result: Optional[list] = None
    text_upright: Optional[bool] = None          # text_line_orientation's vote (ocr_system_task.py:395-439); None = not run
    rotated_180: bool = False                    # the page was voted upside-down and rotated before the stages ran (:471-478)
    text_line_orientation: Optional[list] = None  # per detected line: {"class_ids", "scores", "label_names"}


class OcrTablePipeline:
    def __init__(self, device: int = 0, detect_model: str = "db", recognizer: str = "CRNN", thresh: float = 0.2,
                 synthetic_seed: Optional[int] = None, det_task_path: Optional[str] = None,
                 rec_task_path: Optional[str] = None, layout: bool = False, table_structure: bool = False,
                 table_structure_model: str = "Lore", table_structure_task_type: str = "wtw",
                 tsr_task_path: Optional[str] = None, layout_model: str = "picodet", layout_task_type: str = "en",
                 layout_task_path: Optional[str] = None, text_orientation: bool = False,
                 orientation_task_path: Optional[str] = None, table_html: bool = False, overlap_rec: bool = True,
                 rotate_upside_down: bool = True, aux_layout: bool = False, tsr_on_aux: bool = False, lookahead: int = 1,
                 precision: str = "bf16", layout_precision: Optional[str] = None, **kwargs):
        self.engine = HipEngine(device)
        # arithmetic of every stage on this engine: "bf16" (BASELINE.json's), "fp16" (the reference's own default precision,
        # base_infer_task.py:56-57: the engine's single-pass IEEE-half mode, same speed, 8x finer rounding) or "fp32" (three-pass pair mode)
        _p = str(precision).lower()
        if _p in ("fp16", "f16", "half", "float16"):
            self.engine.set_precision(L.PT_PRECISION_F16)
        elif _p in ("fp32", "bf16x3", "float32"):
            self.engine.set_precision(L.PT_PRECISION_BF16X3)
        elif _p != "bf16":
            raise ValueError(f"precision={precision!r}: expected 'bf16', 'fp16' or 'fp32'")
        kwargs = dict(kwargs, precision=precision)
        # predict_stream() schedule switches: layout / the Lore processor on an auxiliary stream beside the main one.  Off: ONE compute
        # stream measured faster (the weight-stationary cluster LSTM wants the GPU to itself, and concurrent small kernels slowed the
        # large ones: 500 vs 590 pages/s on 64-page batches), at the price of results arriving three batches behind instead of two
        self.aux_layout, self.tsr_on_aux, self.lookahead = aux_layout, tsr_on_aux, lookahead
        # the recogniser of a page batch runs on a second stream beside the layout and table-structure stages of the same
        # batch (every stage owns its arena inside the engine); the results do not depend on it.  The weight-stationary
        # cluster LSTM needs the whole GPU to itself (co-resident workgroups), so an overlapping pipeline runs the
        # streaming LSTM kernel instead (pt_engine_set_lstm_cluster)
        self.overlap_rec = overlap_rec
        # text_orientation=True: a page voted upside-down is rotated by 180 degrees and detected again, like
        # OcrSystemTask.image_pre_process (ocr_system_task.py:471-478); every result of that page is in rotated pixels
        self.rotate_upside_down = rotate_upside_down
        self._rec_stream = None
        dk = dict(kwargs)
        rk = dict(kwargs)
        if synthetic_seed is not None:
            dk["synthetic_seed"] = synthetic_seed
            rk["synthetic_seed"] = synthetic_seed + 1
        if det_task_path:
            dk["task_path"] = det_task_path
        if rec_task_path:
            rk["task_path"] = rec_task_path
        self.text_detector = OcrDetectionTask(model=detect_model, thresh=thresh, engine=self.engine, **dk)
        self.text_recognizer = OcrRecognitionTask(model=recognizer, engine=self.engine, **rk)
        self.layout_task = None
        if layout:
            lk = dict(kwargs)
            if synthetic_seed is not None:
                lk["synthetic_seed"] = synthetic_seed + 4
            if layout_task_path:
                lk["task_path"] = layout_task_path
            if layout_precision:      # the layout net alone in the pair mode ("fp32"): exact table crops under a 16-bit engine (LayoutStage.precision)
                lk["stage_precision"] = layout_precision
            self.layout_task = OcrLayoutTask(model=layout_model, engine=self.engine, task_type=layout_task_type, **lk)
        self.table_html = table_html      # structure + recognised text -> HTML per table (OcrTableToHtmlTask, section 8f-2)
        self.orientation_task = None
        if text_orientation:      # PP-LCNet text-line orientation over every detected line (ocr_system_task.py:116-146, 395-439)
            from .cls_image_pulc_task import ClsImagePulcTask
            ok = dict(kwargs)
            if synthetic_seed is not None:
                ok["synthetic_seed"] = synthetic_seed + 5
            if orientation_task_path:
                ok["task_path"] = orientation_task_path
            self.orientation_task = ClsImagePulcTask(task_type="textline_orientation", engine=self.engine, slot=0, **ok)
        self.table_structure_task = None
        if table_structure:
            tk = dict(kwargs)
            if synthetic_seed is not None:
                tk["synthetic_seed"] = synthetic_seed + 2
            if tsr_task_path:
                tk["task_path"] = tsr_task_path
            self.table_structure_task = OcrTableStructureTask(model=table_structure_model, engine=self.engine,
                                                              task_type=table_structure_task_type, **tk)

    @classmethod
    def from_engine(cls, engine: HipEngine, det_stage: DetStage, rec_stage, layout_stage=None, tsr_stage=None, table_html: bool = False,
                    overlap_rec: bool = False, aux_layout: bool = False, tsr_on_aux: bool = False, lookahead: int = 1) -> "OcrTablePipeline":
        """The façade over an engine whose weights are ALREADY loaded -- e.g. packed once on rank 0, broadcast over RCCL and loaded from
        device memory on every rank (dist_utils.broadcast_blob, ``bench.py --gpus N``) -- and over stage objects the caller built on it.
        ``predict()`` / ``predict_stream()`` are the same code as after the ordinary constructor."""
        import types
        self = cls.__new__(cls)
        self.engine, self.overlap_rec, self.rotate_upside_down, self._rec_stream = engine, overlap_rec, True, None
        self.table_html, self.orientation_task, self.aux_layout, self.tsr_on_aux = table_html, None, aux_layout, tsr_on_aux
        self.lookahead = lookahead
        self.text_detector = types.SimpleNamespace(_stage=det_stage)
        self.text_recognizer = types.SimpleNamespace(_stage=rec_stage)
        self.layout_task = None if layout_stage is None else types.SimpleNamespace(_stage=layout_stage, detect_pages=layout_stage)
        self.table_structure_task = None if tsr_stage is None else types.SimpleNamespace(
            _stage=tsr_stage, recognize_tables=lambda pages, boxes, page_frame=True: tsr_stage(pages, boxes, page_frame=page_frame))
        return self

    def predict(self, pages: Sequence, table_boxes: Optional[Sequence[np.ndarray]] = None, **kwargs) -> List[PageResult]:
        """pages: RGB images (paths / PIL / ndarrays); table_boxes: per page int [k,4] x1,y1,x2,y2 table regions (what the
        layout stage would deliver).  Returns one PageResult per page, plus ``self.metric``."""
        if self.table_structure_task is not None and table_boxes is None and self.layout_task is None:
            raise ValueError("table_structure=True needs layout=True or predict(table_boxes=...)")
        t0 = time.time()
        imgs = [_read_image(p) for p in pages]
        results: List[Optional[PageResult]] = [None] * len(imgs)
        groups: Dict[tuple, List[int]] = {}
        for i, im in enumerate(imgs):
            groups.setdefault(im.shape, []).append(i)
        t_det = t_rec = t_tsr = 0.0
        # opt-in (PT_PREDICT_CHUNK=n or self.predict_chunk = n; default 0 = off): a group of >= 2 n equally sized pages goes through
        # predict_stream() in chunks of n, so that the host halves of one chunk (contours, unclip, NMS, CTC collapse, result shaping) run
        # while the device works on the next, where the loop below waits for every stage of the whole batch in turn.  Same per-page results
        # (no kernel looks across pages: test_predict_in_chunks_equals_predict).  Measured on 64 host pages per call (tools/pipeline_timing.py):
        # serial 315-332 pages/s, n = 32: 351, n = 16: 294-308, n = 8: 179 -- the stream's three steps of latency and the smaller launches
        # eat what the overlap gives, which is why it is off by default and predict_stream() is the throughput API.
        chunk = int(os.environ.get("PT_PREDICT_CHUNK", str(getattr(self, "predict_chunk", 0))))
        streamed = {}
        if chunk > 0 and self.orientation_task is None:
            for shape, idxs in groups.items():
                if len(idxs) < 2 * chunk:
                    continue
                parts = [idxs[j:j + chunk] for j in range(0, len(idxs), chunk)]
                tbs = None if table_boxes is None else [[np.asarray(table_boxes[i]).reshape(-1, 4) for i in part] for part in parts]
                a = time.time()
                for part, res in zip(parts, self.predict_stream(([imgs[i] for i in part] for part in parts), table_boxes=tbs)):
                    for i, r in zip(part, res):
                        results[i] = r
                t_det += time.time() - a      # the stages overlap: the group's wall time is booked here
                streamed[shape] = True
        for shape, idxs in groups.items():
            if shape in streamed:
                continue
            batch = torch.from_numpy(np.stack([imgs[i] for i in idxs])).to(self.engine._tdev)
            a = time.time()
            stage: DetStage = self.text_detector._stage

            def detect(pages_t):
                prob, bitmap, ev = stage.forward(pages_t)
                return [sort_boxes_reading_order(b) for b in stage.boxes(prob, bitmap, shape[:2], ev)]

            with stage_range("text_detection"):
                boxes = detect(batch)
            ori = None
            rotated = [False] * len(idxs)
            if self.orientation_task is not None:
                # per page: classify every detected line, vote (ocr_system_task.py:395-439); a page voted upside-down is
                # rotated by 180 degrees and detected again (:471-478) -- the vote of the FIRST detection is what is kept
                flat, _ = self.orientation_task.lines(batch, boxes)
                ori, o = [], 0
                for b in boxes:
                    res = flat[o:o + len(b)]
                    o += len(b)
                    ori.append((res, self.orientation_task._stage.orientation_vote(res)))
                down = [k for k, (_, up) in enumerate(ori) if not up] if self.rotate_upside_down else []
                if down:
                    sel = torch.tensor(down, device=batch.device)
                    batch[sel] = torch.flip(batch[sel], dims=(1, 2))
                    again = detect(batch[sel].contiguous())
                    for k, bx in zip(down, again):
                        boxes[k], rotated[k] = bx, True
            b_ = time.time()
            rec_stage = self.text_recognizer._stage
            side = None
            if self.overlap_rec and (self.layout_task is not None or self.table_structure_task is not None):
                if self._rec_stream is None:
                    self._rec_stream = shared_stream(self.engine._tdev, "rec")
                    self.engine.set_lstm_cluster(False)      # the recogniser now shares the GPU with the other stages
                side = self._rec_stream
                side.wait_stream(torch.cuda.current_stream(self.engine._tdev))
                batch.record_stream(side)

            def on_side():
                return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

            def recognise():
                with on_side():
                    return rec_stage.start(batch, boxes)            # asynchronous: crops, CRNN, arg-max

            def finish_rec():
                return self._finish_rec(rec_stage, batch, boxes, rec_state, side)

            rec_state = None
            with stage_range("text_recognition"):
                try:
                    rec_state = recognise()
                except Exception as e:                 # noqa: BLE001
                    logger.warning("text recognition could not be queued: %r", e)
                texts = finish_rec() if side is None else None
            c = time.time()
            with stage_range("layout"):
                lay = self.layout_task.detect_pages(batch) if self.layout_task is not None else None
            tsr = None
            if self.table_structure_task is not None:
                if table_boxes is not None:
                    tb = [np.asarray(table_boxes[i]).reshape(-1, 4) for i in idxs]
                    for k in range(len(idxs)):      # boxes were given for the page as handed in: follow its rotation
                        if rotated[k] and len(tb[k]):
                            x1, y1, x2, y2 = tb[k].T
                            tb[k] = np.stack([shape[1] - x2, shape[0] - y2, shape[1] - x1, shape[0] - y1], 1)
                else:
                    tb = self._layout_table_boxes(lay)
                with stage_range("table_structure"):
                    tsr = self.table_structure_task.recognize_tables(batch, tb)
            if side is not None:
                with stage_range("text_recognition"):
                    texts = finish_rec()
            if tsr is not None and self.table_html:
                self._attach_html(tsr, tb, boxes, texts)
            d_ = time.time()
            t_det += b_ - a
            t_rec += c - b_
            t_tsr += d_ - c
            for k, i in enumerate(idxs):
                # bbox = OcrCommonUtils.order_point(det_result[j]) (ocr_system_task.py:311-320), what OcrCell.parse consumes
                pts = order_points(boxes[k]) if len(boxes[k]) else np.zeros((0, 4, 2), np.float32)
                ocr = [{"index": j + 1, "text": t, "bbox": pts[j]} for j, t in enumerate(texts[k])]
                results[i] = PageResult(rotated_180=rotated[k],
                                        det_result=boxes[k], ocr_result=ocr, layout_result=None if lay is None else lay[k],
                                        table_structure_result=None if tsr is None else tsr[k],
                                        text_upright=None if ori is None else ori[k][1],
                                        text_line_orientation=None if ori is None else ori[k][0])
        self.metric = {"use_time": time.time() - t0, "text_detection": {"use_time": t_det},
                       "text_recognition": {"use_time": t_rec, "total": sum(len(r.ocr_result) for r in results)},
                       "table_structure": {"use_time": t_tsr}}
        return results

    # ------------------------------------------------------------------------------------------------------------------
    def _layout_table_boxes(self, lay_pages) -> List[np.ndarray]:
        """layout regions labelled "table", score >= 0.2, top to bottom, at rounded coordinates
        (ocr_system_task.py:184-198, crop_image_by_box utils/ocr/ocr_common_utils.py:279-280)"""
        tb = []
        for lay in lay_pages:
            bx = [[round(float(v)) for v in t["bbox"]] for t in layout_tables(lay, "table", 0.2)]
            bx = [b for b in bx if b[2] > b[0] and b[3] > b[1]]
            tb.append(np.array(bx, dtype=np.int64).reshape(-1, 4))
        return tb

    @staticmethod
    def _drop_empty_crops(tb: Sequence[np.ndarray], shape, kept: Optional[list] = None) -> List[np.ndarray]:
        """table boxes whose crop is empty once clamped to the page are dropped ONE BY ONE, each with a log line -- the reference contains a failed
        crop the same way (ocr_system_task.py:275-283 turns a failing line crop into ''); every other error of the table stage propagates
        (an ``except ValueError`` around the whole batch used to swallow the engine wrappers' own shape errors: ADVICE r04).
        ``kept`` (a list to fill): per page, the indices of the boxes that stay -- _realign_tables() uses them to hand CALLER-supplied
        ``table_boxes`` their results index-aligned, with None where a box was dropped."""
        ph, pw = int(shape[0]), int(shape[1])
        out = []
        for pi, boxes in enumerate(tb):
            b = np.asarray(boxes).reshape(-1, 4)
            idx, n_in = np.arange(len(b)), len(b)
            if len(b):
                ok = (np.minimum(b[:, 2], pw) > np.maximum(b[:, 0], 0)) & (np.minimum(b[:, 3], ph) > np.maximum(b[:, 1], 0))
                for bad in b[~ok]:
                    logger.warning("table region %s of page %d of the batch is empty on a %dx%d page: skipped", bad.tolist(), pi, ph, pw)
                b, idx = b[ok], idx[ok]
            out.append(b)
            if kept is not None:
                kept.append((idx, n_in))
        return out

    @staticmethod
    def _realign_tables(tsr, kept):
        """per-page table results -> lists as long as the caller's ``table_boxes[page]``, result j at the position of box j, None for a dropped box"""
        if tsr is None or kept is None:
            return tsr
        out = []
        for page_res, (idx, n_in) in zip(tsr, kept):
            if len(idx) == n_in:
                out.append(page_res)
                continue
            row = [None] * n_in
            for j, res in zip(idx.tolist(), page_res):
                row[j] = res
            out.append(row)
        return out

    def predict_stream(self, batches, table_boxes=None):
        """``predict()`` over a stream of page batches, software-pipelined: a generator that takes an iterable of batches (each a
        sequence of equally sized RGB pages, or a uint8 tensor [n, h, w, 3] already on the device) and yields one
        ``List[PageResult]`` per batch, in order, with the results ``predict()`` gives for that batch.

        The reference runs a page's stages back to back and waits for each (ocr_system_task.py:549-734); ``predict()`` keeps
        that order per batch.  Here batch k's device work is queued while the host still works on the batches before it:

            step k:  queue layout(k) and detection(k); recognition(k-a) and the table detector + decode (k-a) on the boxes / regions
                     the host produced for batch k-a
                     batch k-2a: cell counts from pinned memory -> queue the Lore processor over its cells
                     host: texts and tables of batch k-2a-1 (CTC collapse, result shaping, HTML)  -> yield
                     host: boxes of batch k-a+1 (contours, box scores, unclip, reading order), its layout decode + NMS

        on ONE compute stream (copies ride on their own streams behind events), with a = ``lookahead`` (constructor, default 1: results
        arrive three batches behind the input; a = 2 lets the enqueue thread run two steps ahead of the GPU at five batches of latency);
        the generator drains at the end.  ``tsr_on_aux=True`` runs the processor on an auxiliary stream in the collect step
        instead (a = 1, two batches of latency, measured slower: concurrent small kernels beside the cluster LSTM and the large
        convolutions); ``aux_layout=True`` does the same for the layout network.
        ``table_boxes``: optional iterable aligned with ``batches`` (per batch: per-page int [k, 4] regions).  The text-line
        orientation vote needs a second, dependent detection pass per page and is not pipelined: use ``predict()`` for it."""
        if self.orientation_task is not None:
            raise ValueError("predict_stream() does not run the text-line orientation vote; use predict()")
        if self.table_structure_task is not None and table_boxes is None and self.layout_task is None:
            raise ValueError("table_structure=True needs layout=True or predict_stream(table_boxes=...)")
        dev = self.engine._tdev
        main = torch.cuda.current_stream(dev)
        if self.overlap_rec and self._rec_stream is None:
            self._rec_stream = shared_stream(dev, "rec")
            self.engine.set_lstm_cluster(False)      # the recogniser shares the GPU with the other stages (see __init__)
        need_aux = bool(getattr(self, "aux_layout", False) or getattr(self, "tsr_on_aux", False))
        if need_aux and getattr(self, "_aux_stream", None) is None:      # every extra stream competes for the few hardware queues
            self._aux_stream = shared_stream(dev, "aux")
        # overlap_rec=False: the recogniser stays on the main stream behind detection (weight-stationary cluster LSTM, the GPU
        # to itself) -- with nothing in this schedule waiting on the newest work, that measures faster than sharing the CUs
        rec_s, aux = (self._rec_stream if self.overlap_rec else main), getattr(self, "_aux_stream", None)
        det: DetStage = self.text_detector._stage
        rec_stage = self.text_recognizer._stage
        lay_stage = self.layout_task._stage if self.layout_task is not None else None
        tsr_stage = self.table_structure_task._stage if self.table_structure_task is not None else None
        tb_iter = iter(table_boxes) if table_boxes is not None else None
        t_start = time.time()
        total_lines = 0
        host = {"queue_first": 0.0, "queue_second": 0.0, "process_tables": 0.0, "collect": 0.0, "host_halves": 0.0}      # host seconds per phase (self.metric)

        def queue_first(batch, k):
            """layout(k) + detection(k)"""
            up = torch.cuda.Event()
            if torch.is_tensor(batch) and batch.device.type == "cpu":
                # host batch: H2D on a copy stream, queued NOW -- the enqueue thread runs ahead of the GPU, so the transfer overlaps the
                # compute of the batches before it (asynchronous when the batch is pinned; a pageable batch is staged by the runtime)
                if getattr(self, "_copy_stream", None) is None:
                    self._copy_stream = shared_stream(dev, "h2d")
                with torch.cuda.stream(self._copy_stream):
                    pages_t = batch.to(dev, non_blocking=batch.is_pinned())
                up.record(self._copy_stream)
                main.wait_event(up)
                pages_t.record_stream(main)
            elif torch.is_tensor(batch):
                pages_t = batch.to(dev)
                up.record(main)
            else:
                imgs = [_read_image(p) for p in batch]
                if len({im.shape for im in imgs}) != 1:
                    raise ValueError("predict_stream() needs equally sized pages inside a batch (predict() groups by size)")
                pages_t = torch.from_numpy(np.stack(imgs)).to(dev)
                up.record(main)
            st = {"pages": pages_t, "shape": tuple(pages_t.shape[1:3]), "n": pages_t.shape[0],
                  "tb": [np.asarray(b).reshape(-1, 4) for b in next(tb_iter)] if tb_iter is not None else None}
            st["uploaded"] = up
            if lay_stage is not None:
                if getattr(self, "aux_layout", False):
                    with torch.cuda.stream(aux):
                        aux.wait_event(up)
                        st["lay"] = lay_stage.forward(pages_t)
                    pages_t.record_stream(aux)
                else:       # one compute stream: per-kernel durations are those of the kernel alone (what bench.py's roofline divides by)
                    with stage_range("layout"):
                        st["lay"] = lay_stage.forward(pages_t)
            with stage_range("text_detection"):
                st["det"] = det.forward(pages_t, slot=k & 1, early_copy=True)
            return st

        def host_halves(st):
            """host halves of detection / layout of the batch queued one step ago"""
            prob, bitmap, ev = st["det"]
            t0 = time.perf_counter()
            st["boxes"] = [sort_boxes_reading_order(b) for b in det.boxes(prob, bitmap, st["shape"], ev)]
            t1 = time.perf_counter()
            st["layout"] = lay_stage.finish(st["lay"][0], st["lay"][1], st["shape"]) if lay_stage is not None else None
            host["halves.boxes"] = host.get("halves.boxes", 0.0) + t1 - t0
            host["halves.layout"] = host.get("halves.layout", 0.0) + time.perf_counter() - t1

        def queue_second(st):
            """recognition + table detector / decode on the boxes and regions host_halves() produced one step ago"""
            t0 = time.perf_counter()
            with torch.cuda.stream(rec_s), stage_range("text_recognition"):
                rec_s.wait_event(st["uploaded"])
                try:
                    st["rec"] = rec_stage.start(st["pages"], st["boxes"])
                except Exception as e:                 # noqa: BLE001
                    logger.warning("text recognition could not be queued: %r", e)
                    st["rec"] = None
            st["pages"].record_stream(rec_s)
            t1 = time.perf_counter()
            host["second.rec_start"] = host.get("second.rec_start", 0.0) + t1 - t0
            if staged_tsr:
                st["kept"] = [] if st["tb"] is not None else None      # caller-supplied boxes: results go back index-aligned (collect)
                tb = self._drop_empty_crops(st["tb"] if st["tb"] is not None else self._layout_table_boxes(st["layout"]), st["shape"], st["kept"])
                st["tb"] = tb
                tables, metas = tsr_stage.tables(st["shape"], tb)
                offs = np.stack([tables["x0"], tables["y0"]], 1).astype(np.float32) if len(tables) else None
                t2 = time.perf_counter()
                with stage_range("table_structure"):
                    pending = tsr_stage.start(st["pages"], tables) if len(tables) else None
                host["second.tsr_tables"] = host.get("second.tsr_tables", 0.0) + t2 - t1
                host["second.tsr_start"] = host.get("second.tsr_start", 0.0) + time.perf_counter() - t2
                ev = None
                if pending is not None:      # the processor of these tables runs on the auxiliary stream, behind this event
                    ev = torch.cuda.Event()
                    ev.record(main)
                    # no record_stream() on the decode outputs: `st` keeps them alive until collect() has waited for the processor's
                    # rows, i.e. past their last use on the auxiliary stream -- and a 267 MB block with a foreign-stream use recorded
                    # cannot be re-used by the caching allocator when it is freed: every batch then paid a hipMalloc (a device-wide
                    # synchronisation) for its successor (measured: 443-462 -> pages/s of the private loop)
                st["tsr"] = (pending, metas, offs, ev)

        staged_tsr = tsr_stage is not None and hasattr(tsr_stage, "start")      # Lore: start / process / collect; MtlTabNet: one synchronous call
        tsr_on_aux = bool(getattr(self, "tsr_on_aux", False))

        def process_tables(st):
            """device half 2 of the table stage for a batch whose decode was queued one step ago: cell counts from pinned memory (waits
            for THAT decode only), the processor over all its cells, rows on their way to pinned memory"""
            st["processed"] = None
            if not staged_tsr or st["tsr"][0] is None:
                return
            pending, ev = st["tsr"][0], st["tsr"][3]
            w0 = getattr(tsr_stage, "wait_s", 0.0)
            if tsr_on_aux:
                with torch.cuda.stream(aux):      # behind the decode of THESE tables only, beside whatever the main stream runs
                    aux.wait_event(ev)
                    st["processed"] = tsr_stage.process(pending)
            else:
                with stage_range("table_structure.processor"):
                    st["processed"] = tsr_stage.process(pending)
            host["tables.wait_decode"] = host.get("tables.wait_decode", 0.0) + getattr(tsr_stage, "wait_s", 0.0) - w0

        def collect(st) -> List[PageResult]:
            nonlocal total_lines
            t0 = time.perf_counter()
            texts = self._finish_rec(rec_stage, st["pages"], st["boxes"], st["rec"], rec_s)
            t1 = time.perf_counter()
            host["collect.texts"] = host.get("collect.texts", 0.0) + t1 - t0
            tsr = None
            if tsr_stage is not None and not staged_tsr:
                st["kept"] = [] if st["tb"] is not None else None
                tb = self._drop_empty_crops(st["tb"] if st["tb"] is not None else self._layout_table_boxes(st["layout"]), st["shape"], st["kept"])
                st["tb"] = tb
                # a table stage without start / process / collect halves (MtlTabNet) decodes synchronously and polls its stream every few steps;
                # on the main stream every poll would wait for the detection / recognition work of the NEXT batches already queued there and
                # the software pipeline would run serially.  It gets its own stream behind this batch's upload instead.
                if getattr(self, "_table_stream", None) is None:
                    self._table_stream = shared_stream(dev, "table")
                ts = self._table_stream
                with torch.cuda.stream(ts):
                    ts.wait_event(st["uploaded"])
                    tsr = tsr_stage(st["pages"], tb, page_frame=True)
                st["pages"].record_stream(ts)
            elif tsr_stage is not None:
                pending, metas, offs, ev = st["tsr"]
                flat = []
                if pending is not None:
                    if "processed" not in st:
                        process_tables(st)
                    flat = tsr_stage.collect(st["processed"], metas, offs)
                    host["collect.tsr_collect"] = host.get("collect.tsr_collect", 0.0) + time.perf_counter() - t1
                tsr = tsr_stage.regroup(flat, st["tb"])
            if tsr is not None and self.table_html:
                self._attach_html(tsr, st["tb"], st["boxes"], texts)
            tsr = self._realign_tables(tsr, st.get("kept"))
            out = []
            for k in range(st["n"]):
                boxes = st["boxes"][k]
                pts = order_points(boxes) if len(boxes) else np.zeros((0, 4, 2), np.float32)
                ocr = [{"index": j + 1, "text": t, "bbox": pts[j]} for j, t in enumerate(texts[k])]
                total_lines += len(ocr)
                out.append(PageResult(det_result=boxes, ocr_result=ocr, layout_result=None if st["layout"] is None else st["layout"][k],
                                      table_structure_result=None if tsr is None else tsr[k]))
            return out

        gpu_marks = [] if os.environ.get("PT_PIPE_GPU_TRACE") else None      # diagnostics: (phase, begin event, end event) on the main stream

        def timed(name, fn, *a):
            with stage_range("predict_stream." + name):      # roctx span of the phase (rocprofv3 --marker-trace); the stages inside name their own
                return timed_(name, fn, *a)

        def timed_(name, fn, *a):
            t0 = time.perf_counter()
            if gpu_marks is not None and name in ("queue_first", "queue_second", "process_tables"):
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(main)
                r = fn(*a)
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(main)
                gpu_marks.append((name, e0, e1))
            else:
                r = fn(*a)
            host[name] += time.perf_counter() - t0
            return r

        # stages of the software pipeline a batch walks through, one per step: [0] layout + detection queued, [1] host halves done,
        # recognition + table decode queued, [2] (single-stream schedule only) processor queued, then collected and yielded
        depth = 2 if (tsr_on_aux or not staged_tsr) else 3

        # Schedule.  Batch j walks through: first (layout + detection queued) at step j, host halves at the END of step j + a - 1, second
        # (recognition + table decode queued) at step j + a, processor at step j + 2a, collect at step j + 2a + 1, with a = self.lookahead.
        # a = 1 (default): results three batches behind the input.  a = 2: every wait of the host targets device work queued two steps
        # earlier, so the enqueue thread may run two steps ahead of the GPU (five batches of latency) -- built while hunting a 490-vs-580
        # pages/s run-to-run bimodality; the cause turned out to be a late-issued 7 MB copy stuck behind other copies on the DMA engine
        # (DetStage.forward), and with that fixed a = 1 measures 587-592 against 579-583 for a = 2 (one more drain step in 20).
        a_ = max(1, int(getattr(self, "lookahead", 1)))
        if tsr_on_aux or not staged_tsr:
            a_ = 1
        flight: Dict[int, dict] = {}
        k = 0

        def step(j: int, cur):
            """one step of the schedule at stream position j (cur: the batch that arrived, or None while draining)"""
            if cur is not None:
                flight[j] = cur
            st = flight.get(j - a_)
            if st is not None:
                timed("queue_second", queue_second, st)
            if depth == 3:
                st = flight.get(j - 2 * a_)
                if st is not None:
                    timed("process_tables", process_tables, st)
            done = flight.pop(j - 2 * a_ - (1 if depth == 3 else 0), None)
            res = timed("collect", collect, done) if done is not None else None
            st = flight.get(j - (a_ - 1))
            if st is not None and "boxes" not in st:
                timed("host_halves", host_halves, st)
            return res

        for batch in batches:
            cur = timed("queue_first", queue_first, batch, k)
            res = step(k, cur)
            k += 1
            if res is not None:
                yield res
        j = k
        while flight:
            res = step(j, None)
            j += 1
            if res is not None:
                yield res
        self.metric = {"use_time": time.time() - t_start, "batches": k, "text_recognition": {"total": total_lines},
                       "host_seconds": host}
        if gpu_marks:
            torch.cuda.synchronize(dev)
            busy: Dict[str, float] = {}
            idle: Dict[str, float] = {}
            for i, (name, e0, e1) in enumerate(gpu_marks):
                busy[name] = busy.get(name, 0.0) + e0.elapsed_time(e1)
                if i:
                    key = gpu_marks[i - 1][0] + "->" + name
                    idle[key] = idle.get(key, 0.0) + gpu_marks[i - 1][2].elapsed_time(e0)
            self.metric["gpu_ms_inside_phases"] = busy         # device time between the first and the last launch of a phase
            self.metric["gpu_ms_between_phases"] = idle        # device time between two phases' launches: the queue was empty (or copies ran)

    def _attach_html(self, tsr, tb, boxes, texts):
        """cells (page pixels since the stage shifts them) x the page's text lines (page pixels) -> HTML per table"""
        from .table_text_match import page_table_html
        for k in range(len(tsr)):
            for ti, table in enumerate(tsr[k]):
                if len(table.get("scores", [])) == 0:
                    table["table_html"], table["db_table_html"] = [], []
                    continue
                table["table_html"], table["db_table_html"] = page_table_html(
                    table["polygons"], table["logi"], tb[k][ti], boxes[k], texts[k])

    def _finish_rec(self, rec_stage, pages_t, boxes, state, stream) -> List[List[str]]:
        """-> texts; a device-side failure (PtError from pt_engine_check: the engine has switched to the streaming LSTM and
        cleared the flag) is logged and the batch is run once more; only then the reference's containment (empty strings,
        ocr_system_task.py:275-283) applies -- logged, never silent"""
        from .lib import PtError
        ctx = (lambda: torch.cuda.stream(stream)) if stream is not None else contextlib.nullcontext
        for attempt in (0, 1):
            try:
                with ctx():
                    if state is None:
                        state = rec_stage.start(pages_t, boxes)
                    return rec_stage.finish(state)
            except PtError as e:
                logger.warning("text recognition failed on the device (%s)%s", e, "; running the batch again" if attempt == 0 else "")
                state = None
            except Exception as e:     # noqa: BLE001 -- the reference contains every failure of a crop
                logger.warning("text recognition failed: %r", e)
                break
        logger.error("text recognition of %d lines yields empty strings", sum(len(b) for b in boxes))
        return [[""] * len(b) for b in boxes]
