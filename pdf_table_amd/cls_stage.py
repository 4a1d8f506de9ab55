"""Batched PP-LCNet image classification on the HIP engine: task tables and the host-side post-processors.

Reference: model/cls/configuration_cls_pulc.py:20-39 (class counts, stride lists), model/cls/image_processing_pplcnet.py
:40-107 (label maps, input sizes, top-k), :109-152 (``TableAttribute``), :155-192 (``Topk``).  The device half
(Pillow-exact resize, normalisation, LCNet, pooled head) is ``pt_cls_forward*``; soft-max / top-k over <= 10 classes is
host work."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .engine import HipEngine

__all__ = ["CLS_TASKS", "CLASS_ID_MAP", "ClsStage", "topk_postprocess", "table_attribute_postprocess"]

CLS_TASKS = {
    "table_attribute": {"class_num": 6, "textline": False, "size": (224, 224), "topk": None},
    "text_image_orientation": {"class_num": 4, "textline": False, "size": (224, 224), "topk": 2},
    "textline_orientation": {"class_num": 2, "textline": True, "size": (80, 160), "topk": 1},
    "language_classification": {"class_num": 10, "textline": True, "size": (80, 160), "topk": 2},
}
CLASS_ID_MAP = {
    "text_image_orientation": {0: "0", 1: "90", 2: "180", 3: "270"},
    "textline_orientation": {0: "0_degree", 1: "180_degree"},
    "language_classification": {0: "arabic", 1: "chinese_cht", 2: "cyrillic", 3: "devanagari", 4: "japan", 5: "ka",
                                6: "korean", 7: "ta", 8: "te", 9: "latin"},
}


def topk_postprocess(logits: np.ndarray, task: str) -> List[Dict]:
    """``Topk.__call__``: F.softmax, ``argsort()[-k:][::-1]``, scores rounded to 5 decimals, label names"""
    k = CLS_TASKS[task]["topk"]
    probs = torch.softmax(torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32)), dim=-1).numpy()
    order = np.argsort(probs, axis=1)[:, -k:][:, ::-1].astype("int32")
    names = CLASS_ID_MAP[task]
    res = []
    for p, idx in zip(probs, order):
        res.append({"class_ids": [int(i) for i in idx], "scores": np.around([p[i].item() for i in idx], decimals=5).tolist(),
                    "label_names": [names[int(i)] for i in idx]})
    return res


def table_attribute_postprocess(outputs: np.ndarray, thresholds: Sequence[float] = (0.5,) * 6) -> List[Dict]:
    """``TableAttribute.__call__``: the raw network outputs against 0.5; `obstruction` and `angle` use number_threshold for
    the label (the reference's own slip, :141-144) and their own threshold in `output`"""
    pairs = (("Scanned", "Photo"), ("Little", "Numerous"), ("Black-and-White", "Multicolor"), ("Clear", "Blurry"),
             ("Without-Obstacles", "With-Obstacles"), ("Horizontal", "Tilted"))
    label_thr = (thresholds[0], thresholds[1], thresholds[2], thresholds[3], thresholds[1], thresholds[1])
    res = []
    for row in np.asarray(outputs).tolist():
        res.append({"attributes": [pairs[i][0] if row[i] > label_thr[i] else pairs[i][1] for i in range(6)],
                    "output": (np.array(row) > np.array(thresholds)).astype(np.int8).tolist()})
    return res


class ClsStage:
    """one classifier task bound to an engine slot"""

    def __init__(self, eng: HipEngine, task: str, slot: int = 0):
        if task not in CLS_TASKS:
            raise KeyError(f"unknown classification task '{task}' (one of {sorted(CLS_TASKS)})")
        self.eng, self.task, self.slot = eng, task, slot
        self.cfg = CLS_TASKS[task]

    def post(self, logits: torch.Tensor) -> List[Dict]:
        x = logits.cpu().numpy()
        return table_attribute_postprocess(x) if self.task == "table_attribute" else topk_postprocess(x, self.task)

    def top1(self, logits: torch.Tensor):
        """vectorised Topk(topk=1) for large batches: (class ids int64 [n], scores f32 [n] rounded to 5 decimals)"""
        p = torch.softmax(logits.cpu().float(), dim=-1).numpy()      # host soft-max over <= 10 classes, like Topk (no device op here)
        ids = np.argsort(p, axis=1)[:, -1]          # same tie rule as Topk: the LAST of equal maxima in argsort order
        return ids, np.around(p[np.arange(len(p)), ids], decimals=5)

    def vote_top1(self, ids: np.ndarray, scores: np.ndarray, score_threshold: float = 0.9) -> bool:
        """orientation_vote on top1() arrays (class 0 = "0_degree")"""
        ok = scores > score_threshold
        return int(np.sum(ok & (ids == 0))) > int(np.sum(ok & (ids != 0)))

    def images(self, images: Sequence[np.ndarray]) -> List[Dict]:
        """RGB uint8 host images of any sizes"""
        if not len(images):
            return []
        return self.post(self.eng.cls_forward(images, self.cfg["size"], self.slot, self.cfg["textline"]))

    def pages(self, pages: torch.Tensor) -> List[Dict]:
        """uint8 [n,h,w,3] pages resident on the device"""
        return self.post(self.eng.cls_forward_pages(pages, self.cfg["size"], self.slot, self.cfg["textline"]))

    def lines(self, pages: torch.Tensor, lines: np.ndarray) -> List[Dict]:
        """text lines (REC_LINE_DTYPE records, pdf_table_amd.rec_stage.build_lines) cut from resident pages"""
        if not len(lines):
            return []
        return self.post(self.eng.cls_forward_lines(pages, lines, self.cfg["size"], self.slot, self.cfg["textline"]))

    def orientation_vote(self, results: List[Dict], score_threshold: float = 0.9) -> bool:
        """OcrSystemTask.text_line_orientation's decision (ocr_system_task.py:418-431): True = upright page"""
        up = sum(1 for r in results if r["scores"][0] > score_threshold and r["label_names"][0] == "0_degree")
        down = sum(1 for r in results if r["scores"][0] > score_threshold and r["label_names"][0] != "0_degree")
        return up > down
