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
import logging
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
from .rec_stage import order_points
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
                 rotate_upside_down: bool = True, **kwargs):
        self.engine = HipEngine(device)
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

            def detect(pages_t):
                prob, bitmap, ev = stage.forward(pages_t)
                return [sort_boxes_reading_order(b) for b in stage.boxes(prob, bitmap, shape[:2], ev)]

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
                    self._rec_stream = torch.cuda.Stream(device=self.engine._tdev)
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
                """-> texts; a device-side failure (PtError from pt_engine_check: the engine has switched to the streaming
                LSTM and cleared the flag) is logged and the batch is run once more; only then the reference's containment
                (empty strings, ocr_system_task.py:275-283) applies -- logged, never silent"""
                from .lib import PtError
                state = rec_state
                for attempt in (0, 1):
                    try:
                        with on_side():
                            if state is None:
                                state = rec_stage.start(batch, boxes)
                            return rec_stage.finish(state)
                    except PtError as e:
                        logger.warning("text recognition failed on the device (%s)%s", e, "; running the batch again" if attempt == 0 else "")
                        state = None
                    except Exception as e:     # noqa: BLE001 -- the reference contains every failure of a crop
                        logger.warning("text recognition failed: %r", e)
                        break
                logger.error("text recognition of %d lines yields empty strings", sum(len(b) for b in boxes))
                return [[""] * len(b) for b in boxes]

            rec_state = None
            try:
                rec_state = recognise()
            except Exception as e:                 # noqa: BLE001
                logger.warning("text recognition could not be queued: %r", e)
            texts = finish_rec() if side is None else None
            c = time.time()
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
                    # layout regions labelled "table", score >= 0.2, top to bottom, cropped at rounded coordinates
                    # (ocr_system_task.py:184-198, crop_image_by_box utils/ocr/ocr_common_utils.py:279-280)
                    tb = []
                    for k in range(len(idxs)):
                        bx = [[round(float(v)) for v in t["bbox"]] for t in layout_tables(lay[k], "table", 0.2)]
                        bx = [b for b in bx if b[2] > b[0] and b[3] > b[1]]
                        tb.append(np.array(bx, dtype=np.int64).reshape(-1, 4))
                tsr = self.table_structure_task.recognize_tables(batch, tb)
            if side is not None:
                texts = finish_rec()
            if tsr is not None and self.table_html:
                from .table_text_match import page_table_html
                for k in range(len(idxs)):
                    for ti, table in enumerate(tsr[k]):
                        if len(table.get("scores", [])) == 0:
                            table["table_html"], table["db_table_html"] = [], []
                            continue
                        # cells (page pixels since recognize_tables shifts them) x the page's text lines (page pixels)
                        table["table_html"], table["db_table_html"] = page_table_html(
                            table["polygons"], table["logi"], tb[k][ti], boxes[k], texts[k])
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
