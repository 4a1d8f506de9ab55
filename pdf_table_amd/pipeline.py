"""``OcrTablePipeline.predict()`` -- the batched page-inference façade named by BASELINE.json's north star.

The reference has no class of this name (SURVEY.md finding F1); its semantics are those of
``OcrSystemTask.__call__`` (src/pdftable/model/ocr_pdf/ocr_system_task.py:549-734) restricted to the vision path of
an *image* page: text detection (:629, :148-166, incl. the reading-order sort) -> text recognition (:630, :296-336).
-> table structure (:192-198, Lore) on the page's table regions.  Layout (PicoDet) is the SURVEY section 8 row that is
not built yet: asking for it raises, and the table regions the reference takes from the layout stage
(``label == "table"``, pdf_table/table_common.py:1287-1301) are passed to ``predict(..., table_boxes=...)`` instead.

What is different from the reference, by design: pages are processed as a batch (the reference is batch 1 and
synchronous), crops never leave the GPU, and a failed page/line follows the reference's containment rules
(a failed crop yields ``""``: ocr_system_task.py:275-283).
"""
from __future__ import annotations

import contextlib
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .det_stage import DetConfig, DetStage, sort_boxes_reading_order
from .engine import HipEngine
from .ocr_detection_task import OcrDetectionTask, _read_image
from .layout_stage import layout_tables
from .ocr_layout_task import OcrLayoutTask
from .ocr_recognition_task import OcrRecognitionTask
from .ocr_table_structure_task import OcrTableStructureTask

__all__ = ["OcrTablePipeline", "PageResult"]


@dataclass
class PageResult:
    """Subset of OcrSystemModelOutput (ocr_output.py:25-61) the built stages fill."""
    det_result: np.ndarray                       # [n, 8] boxes in reading order (ocr_system_task.py:159-162)
    ocr_result: List[Dict] = field(default_factory=list)   # [{"index", "text", "bbox"}] like modeling_ocr_pdf.py:286-291
    layout_result: Optional[list] = None
    table_structure_result: Optional[list] = None
    text_upright: Optional[bool] = None          # text_line_orientation's vote (ocr_system_task.py:395-439); None = not run
    text_line_orientation: Optional[list] = None  # per detected line: {"class_ids", "scores", "label_names"}


class OcrTablePipeline:
    def __init__(self, device: int = 0, detect_model: str = "db", recognizer: str = "CRNN", thresh: float = 0.2,
                 synthetic_seed: Optional[int] = None, det_task_path: Optional[str] = None,
                 rec_task_path: Optional[str] = None, layout: bool = False, table_structure: bool = False,
                 table_structure_model: str = "Lore", table_structure_task_type: str = "wtw",
                 tsr_task_path: Optional[str] = None, layout_model: str = "picodet", layout_task_type: str = "en",
                 layout_task_path: Optional[str] = None, text_orientation: bool = False,
                 orientation_task_path: Optional[str] = None, table_html: bool = False, overlap_rec: bool = True, **kwargs):
        self.engine = HipEngine(device)
        # the recogniser of a page batch runs on a second stream beside the layout and table-structure stages of the same
        # batch (every stage owns its arena inside the engine); the results do not depend on it
        self.overlap_rec = overlap_rec
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
        for shape, idxs in groups.items():
            batch = torch.from_numpy(np.stack([imgs[i] for i in idxs])).to(self.engine._tdev)
            a = time.time()
            stage: DetStage = self.text_detector._stage
            prob, bitmap, ev = stage.forward(batch)
            boxes = stage.boxes(prob, bitmap, shape[:2], ev)
            boxes = [sort_boxes_reading_order(b) for b in boxes]
            b_ = time.time()
            rec_stage = self.text_recognizer._stage
            side = None
            if self.overlap_rec and (self.layout_task is not None or self.table_structure_task is not None):
                if self._rec_stream is None:
                    self._rec_stream = torch.cuda.Stream(device=self.engine._tdev)
                side = self._rec_stream
                side.wait_stream(torch.cuda.current_stream(self.engine._tdev))
                batch.record_stream(side)

            def on_side():
                return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

            rec_state, texts = None, None
            try:
                with on_side():
                    rec_state = rec_stage.start(batch, boxes)        # asynchronous: crops, CRNN, arg-max
                    if side is None:
                        texts = rec_stage.finish(rec_state)
            except Exception:                      # reference: a failing recognition yields empty strings
                rec_state, texts = None, [[""] * len(b) for b in boxes]

            def finish_rec_and_orientation():
                nonlocal texts
                if texts is None:
                    try:
                        with on_side():
                            texts = rec_stage.finish(rec_state)
                    except Exception:
                        texts = [[""] * len(b) for b in boxes]
                if self.orientation_task is None:
                    return None
                res_all = []
                with on_side():          # shares the crop buffers with the recogniser: same stream, behind it
                    flat, _ = self.orientation_task.lines(batch, boxes)
                o = 0
                for b in boxes:       # the reference votes per page; it then rotates a non-upright page by 180 degrees and
                    res = flat[o:o + len(b)]      # detects again (:471-478) -- left to the caller, who holds the pages
                    o += len(b)
                    res_all.append((res, self.orientation_task._stage.orientation_vote(res)))
                return res_all

            ori = finish_rec_and_orientation() if side is None else None
            c = time.time()
            lay = self.layout_task.detect_pages(batch) if self.layout_task is not None else None
            tsr = None
            if self.table_structure_task is not None:
                if table_boxes is not None:
                    tb = [np.asarray(table_boxes[i]).reshape(-1, 4) for i in idxs]
                else:
                    # layout regions labelled "table", score >= 0.2, top to bottom, cropped at rounded coordinates
                    # (ocr_system_task.py:184-198, crop_image_by_box utils/ocr/ocr_common_utils.py:279-280)
                    tb = []
                    for k in range(len(idxs)):
                        bx = [[round(float(v)) for v in t["bbox"]] for t in layout_tables(lay[k], "table", 0.2)]
                        bx = [b for b in bx if b[2] > b[0] and b[3] > b[1]]
                        tb.append(np.array(bx, dtype=np.int64).reshape(-1, 4))
                tsr = self.table_structure_task.recognize_tables(batch, tb)
            if side is not None:
                ori = finish_rec_and_orientation()
            if tsr is not None and self.table_html:
                from .table_html import table_cells_from_logits
                from .table_text_match import cells_to_html, match_table_cells_and_text, text_boxes, texts_in_table
                for k in range(len(idxs)):
                    tbx = text_boxes(boxes[k]) if len(boxes[k]) else np.zeros((0, 4))
                    for ti, table in enumerate(tsr[k]):
                        if len(table.get("scores", [])) == 0:
                            table["table_html"], table["db_table_html"] = [], []
                            continue
                        cells = table_cells_from_logits(table["polygons"], table["logi"])
                        inside = texts_in_table([float(v) for v in tb[k][ti]], tbx, diff=2) if len(tbx) else np.zeros(0, np.int64)
                        # image pages: ocr_post_process=True (ocr_table_to_html_task.py:93)
                        res = match_table_cells_and_text(cells, tbx[inside], [texts[k][i] for i in inside], post_process=True)
                        table["table_html"], table["db_table_html"] = cells_to_html(res)
            d_ = time.time()
            t_det += b_ - a
            t_rec += c - b_
            t_tsr += d_ - c
            for k, i in enumerate(idxs):
                ocr = [{"index": j + 1, "text": t, "bbox": boxes[k][j].reshape(4, 2)} for j, t in enumerate(texts[k])]
                results[i] = PageResult(det_result=boxes[k], ocr_result=ocr, layout_result=None if lay is None else lay[k],
                                        table_structure_result=None if tsr is None else tsr[k],
                                        text_upright=None if ori is None else ori[k][1],
                                        text_line_orientation=None if ori is None else ori[k][0])
        self.metric = {"use_time": time.time() - t0, "text_detection": {"use_time": t_det},
                       "text_recognition": {"use_time": t_rec, "total": sum(len(r.ocr_result) for r in results)},
                       "table_structure": {"use_time": t_tsr}}
        return results
