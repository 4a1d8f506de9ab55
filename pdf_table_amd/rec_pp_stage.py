"""PP-OCR recognition pre-processor on the HIP engine -- the ``PPOcrRecPreProcessor`` of the reference
(model/ocr_rec_pp/processor_ocr_rec_pp.py:24-135), which is the pre-processing of the recogniser its system path
actually selects (``fix_model_names`` forces PP-OCRv4, model/ocr_pdf/configuration_ocr_document.py:138-141).

Host: the reference's own plan -- aspect ratios, ``np.argsort`` (the same numpy call, so ties fall the same way),
mini-batches of ``rec_batch_num``, the padded width of each mini-batch and every crop's resized width
(``resize_norm_img`` :43-67).  Device (``pt_rec_pp_preprocess*``): crop (for lines of resident pages), cv2-exact 8-bit
bilinear resize to 48 x resized_w, ``(x / 255 - 0.5) / 0.5``, zero padding -- written straight in the reference's NCHW
mini-batch layout, so every ``data_batch[i]["image"]`` is a view of one device buffer.

The PP-OCR recognition NETWORK is an ONNX file the reference downloads; it is not in the tree, so this
module stops where the reference hands ``image`` to onnxruntime.  What is deliberately different: the reference's system
loop calls the pre-processor with ONE crop at a time (ocr_system_task.py:309-320) and its post-processor is only correct for
that (processor_ocr_rec_pp.py:158); here all lines of a page batch go through one call.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import REC_LINE_DTYPE, REC_PP_ITEM_DTYPE, HipEngine

__all__ = ["PPOcrRecConfig", "PPOcrRecPreProcessor", "rec_pp_plan"]


class PPOcrRecConfig:
    """the four fields of PPOcrRecognitionConfig (configuration_ocr_recognition_pp.py:43-58) the pre-processor reads"""

    def __init__(self, rec_image_shape: str = "3, 48, 320", limited_max_width: int = 1280, limited_min_width: int = 16,
                 rec_batch_num: int = 6):
        self.rec_image_shape = [int(v.strip()) for v in rec_image_shape.split(",")]
        self.limited_max_width = limited_max_width
        self.limited_min_width = limited_min_width
        self.rec_batch_num = rec_batch_num


def rec_pp_plan(crop_w: Sequence[int], crop_h: Sequence[int], config: Optional[PPOcrRecConfig] = None):
    """-> (items REC_PP_ITEM_DTYPE in processing order, batches [(batch_beg_img_no, n, img_w, float offset)], total floats).
    The arithmetic of PPOcrRecPreProcessor.__call__ (:100-126) and resize_norm_img (:43-58), in Python floats as there."""
    cfg = config or PPOcrRecConfig()
    imgC, imgH, imgW0 = cfg.rec_image_shape
    n = len(crop_w)
    width_list = [int(w) / float(int(h)) for w, h in zip(crop_w, crop_h)]
    indices = np.argsort(np.array(width_list))               # "Sorting can speed up the recognition process"
    items = np.zeros(n, dtype=REC_PP_ITEM_DTYPE)
    batches = []
    off = 0
    for beg in range(0, n, cfg.rec_batch_num):
        end = min(n, beg + cfg.rec_batch_num)
        max_wh_ratio = 0
        for ino in range(beg, end):
            i = indices[ino]
            max_wh_ratio = max(max_wh_ratio, int(crop_w[i]) * 1.0 / int(crop_h[i]))
        max_wh_ratio = max(max_wh_ratio, imgW0 / imgH)
        img_w = int((imgH * max_wh_ratio))
        img_w = max(min(img_w, cfg.limited_max_width), cfg.limited_min_width)
        batches.append((beg, end - beg, img_w, off))
        for ino in range(beg, end):
            i = indices[ino]
            ratio = int(crop_w[i]) / float(int(crop_h[i]))
            ratio_img_h = max(math.ceil(imgH * ratio), cfg.limited_min_width)
            items[ino] = (i, img_w if ratio_img_h > img_w else int(ratio_img_h), img_w, 0, off)
            off += imgC * imgH * img_w
    return items, batches, off


class PPOcrRecPreProcessor:
    def __init__(self, config: Optional[PPOcrRecConfig] = None, engine: Optional[HipEngine] = None, device: int = 0):
        self.config = config or PPOcrRecConfig()
        self.rec_image_shape = self.config.rec_image_shape
        self.rec_batch_num = self.config.rec_batch_num
        self.limited_max_width = self.config.limited_max_width
        self.limited_min_width = self.config.limited_min_width
        self.engine = engine or HipEngine(device)

    def _batches(self, flat: torch.Tensor, items: np.ndarray, batches) -> List[Dict]:
        imgC, imgH, _ = self.rec_image_shape
        indices = items["line"].astype(np.int64)
        return [{"image": flat[off:off + n * imgC * imgH * img_w].view(n, imgC, imgH, img_w), "indices": indices,
                 "batch_beg_img_no": beg} for (beg, n, img_w, off) in batches]

    def __call__(self, inputs) -> List[Dict]:
        """reference call shape: a crop or a list of crops (path / PIL / RGB ndarray; gray ndarrays are replicated to 3
        channels like cv2.COLOR_GRAY2RGB) -> [{'image' f32 [b,3,48,imgW] ON THE DEVICE, 'indices', 'batch_beg_img_no'}]"""
        from .ocr_detection_task import _read_image
        if not isinstance(inputs, list):
            inputs = [inputs]
        crops = []
        for item in inputs:
            if isinstance(item, np.ndarray):
                img = np.repeat(item[:, :, None], 3, 2) if item.ndim == 2 else item
            elif isinstance(item, str) or hasattr(item, "convert"):
                img = _read_image(item)
            else:
                raise TypeError(f"inputs should be either (a list of) str, PIL.Image, np.array, but got {type(item)}")
            crops.append(np.ascontiguousarray(img[:, :, :3], dtype=np.uint8))
        if not crops:
            return []
        lines = np.zeros(len(crops), dtype=REC_LINE_DTYPE)
        lines["crop_w"] = [c.shape[1] for c in crops]
        lines["crop_h"] = [c.shape[0] for c in crops]
        items, batches, total = rec_pp_plan(lines["crop_w"], lines["crop_h"], self.config)
        flat = self.engine.rec_pp_preprocess(None, lines, items, total, self.rec_image_shape[1],
                                             crops_flat=np.concatenate([c.reshape(-1) for c in crops]))
        return self._batches(flat, items, batches)

    def lines(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray]) -> List[Dict]:
        """batched form: every detected box of a page batch (boxes [k,8] per page, source pixels) is cut out of the resident
        pages on the device (order_point + crop_image, as the recognition stage does) and pre-processed in one call"""
        from .rec_stage import build_lines
        lines = build_lines(boxes_per_page)
        keep = (lines["crop_w"] > 0) & (lines["crop_h"] > 0)
        if not keep.any():
            return []
        lines = lines[keep]
        items, batches, total = rec_pp_plan(lines["crop_w"], lines["crop_h"], self.config)
        flat = self.engine.rec_pp_preprocess(pages, lines, items, total, self.rec_image_shape[1])
        return self._batches(flat, items, batches)
