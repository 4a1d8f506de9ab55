"""``OcrSystemTask`` call shape over the batched engine pipeline -- the reference's top-level entry for one page.

Reference: src/pdftable/model/ocr_pdf/ocr_system_task.py:549-734: ``task(inputs, save_result=True, src_id=None, page=None)
-> (OcrSystemModelOutput, metric)`` with the output fields of model/ocr_pdf/ocr_output.py:25-61.  This mirror serves IMAGE
inputs (path / PIL / RGB ndarray) through ``OcrTablePipeline`` -- orientation vote and 180-degree re-detection (:441-491),
layout (:203-215), detection + reading order (:148-166), recognition (:296-336), Lore table structure on the layout's
tables merged into page coordinates (:170-213 -> ``convert_table_sep_to_merge``, pdf_table/table_common.py:1795-1880),
text <-> cell HTML -- and fills the fields those stages produce; everything else of the reference's task (PDF rendering and
pdfminer text, database rows, debug drawings, HTML / Excel files: SURVEY.md section 2, out of scope) stays ``None``, and a PDF
input raises.  ``predict_pages`` is the batched form the engine is built for."""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .pipeline import OcrTablePipeline, PageResult

__all__ = ["OcrSystemModelOutput", "OcrSystemTask"]


@dataclass
class OcrSystemModelOutput:
    """field names and order of the reference's dataclass (ocr_output.py:25-50)"""
    file_name: Optional[str] = None
    src_id: Optional[int] = None
    page: Optional[int] = None
    run_time: Optional[str] = None
    metric: Optional[Dict] = None
    image_full: Optional[np.ndarray] = None
    table_cell_result: Optional[List] = None
    det_result: Union[List, np.ndarray, None] = None
    layout_result: Optional[List] = None
    ocr_result: Optional[List] = None
    raw_filename: Optional[str] = None
    save_html_file: Optional[str] = None
    ocr_cell_content: Optional[List] = None
    merge_ocr_cells: Optional[List] = None
    pdf_html: Optional[List] = None
    image_shape: Optional[List] = None
    image_scalers: Optional[List] = None
    pdf_scalers: Optional[List] = None
    is_pdf: Optional[bool] = None
    table_structure_result: Union[List, Dict, None] = None
    image_rotate: Optional[bool] = None
    all_table_valid_check: Optional[bool] = None
    image_name: Optional[str] = None
    use_master: Optional[bool] = None

    def get_table_structure_bboxs(self):
        bboxs = self.table_structure_result
        if isinstance(bboxs, dict):
            bboxs = bboxs["polygons"]
            if isinstance(bboxs, list):
                bboxs = np.concatenate(bboxs, axis=0)
        return bboxs


def merge_tables(tables: Sequence[Dict]) -> Dict:
    """the dict ``convert_table_sep_to_merge`` builds from the per-table results (table_common.py:1795-1880): polygons
    (already in page pixels here) and logical locations concatenated, plus the per-table lists"""
    valid = [t for t in tables if len(t.get("scores", [])) > 0]
    return {"polygons": np.concatenate([t["polygons"] for t in valid], 0) if valid else np.zeros((0, 8)),
            "structure_str_list": [t.get("structure_str_list", []) for t in tables],
            "logi": np.concatenate([t["logi"] for t in valid], 0) if valid else np.zeros((0, 4)),
            "polygons_sep": [np.asarray(t["polygons"]) for t in tables],
            "logi_sep": [t["logi"] for t in tables],
            "table_html": [t.get("table_html") for t in tables],
            "db_table_html": [t.get("db_table_html") for t in tables],
            "table_cell_metric": {}}


class OcrSystemTask:
    def __init__(self, task="ocr_system", model="ocr_system", device: int = 0, detect_model: str = "db", recognizer: str = "CRNN",
                 layout_model: str = "picodet", table_structure_model: str = "Lore", table_structure_task_type: str = "wtw",
                 text_orientation: bool = True, **kwargs):
        self.task, self.model = task, model
        self.pipeline = OcrTablePipeline(device=device, detect_model=detect_model, recognizer=recognizer, layout=True,
                                         layout_model=layout_model, table_structure=True,
                                         table_structure_model=table_structure_model,
                                         table_structure_task_type=table_structure_task_type,
                                         text_orientation=text_orientation, table_html=True, **kwargs)

    def _output(self, r: PageResult, name, shape, src_id, page) -> OcrSystemModelOutput:
        raw = None if not isinstance(name, str) else name.rsplit("/", 1)[-1].rsplit(".", 1)[0]
        return OcrSystemModelOutput(file_name=name if isinstance(name, str) else None, src_id=src_id, page=page,
                                    run_time=time.strftime("%Y%m%d_%H%M%S"), raw_filename=raw, image_shape=list(shape), is_pdf=False,
                                    image_name=name if isinstance(name, str) else None, det_result=r.det_result,
                                    layout_result=r.layout_result, ocr_result=r.ocr_result,
                                    table_structure_result=None if r.table_structure_result is None else merge_tables(r.table_structure_result),
                                    image_rotate=r.rotated_180)

    def predict_pages(self, pages: Sequence, src_id=None) -> List[Tuple[OcrSystemModelOutput, Dict]]:
        from .ocr_detection_task import _read_image
        imgs = [_read_image(p) for p in pages]
        res = self.pipeline.predict(imgs)
        m = self.pipeline.metric
        out = []
        for k, (p, r, im) in enumerate(zip(pages, res, imgs)):
            o = self._output(r, p, im.shape, src_id, k)
            o.metric = dict(m)
            out.append((o, o.metric))
        return out

    def __call__(self, inputs, save_result=True, src_id=None, page=None, **kwargs) -> Tuple[OcrSystemModelOutput, Dict]:
        if isinstance(inputs, str) and inputs.lower().endswith(".pdf"):
            raise RuntimeError("PDF inputs (rendering, pdfminer text) are outside the engine's scope (SURVEY.md section 8): "
                               "render the page to an image first")
        o, metric = self.predict_pages([inputs], src_id=src_id)[0]
        o.page = page
        return o, metric
