"""``OcrLayoutTask`` on the HIP engine -- drop-in for the reference's stage-1 plug-in (model="picodet").

Reference: src/pdftable/model/ocr_pdf/ocr_layout_task.py:30-157.  Same constructor (``task, model, task_type``), same
result: one list per input image of ``{"bbox": ndarray [4] (x1, y1, x2, y2 in source pixels; float64 holding float32-rounded corners divided by the scale, as in the reference), "label", "score",
"category_id"}`` (:125-141, picodet/processor_picodet.py:286-296).  The reference runs an ONNX export of PaddleDetection's
picodet_lcnet_x1_0 layout model; here the in-tree LCNet + CSP-PAN + PicoHead graph (assumed hyper-parameters, see
pdf_table_amd.synth_weights.picodet_state_dict) runs on the engine.  ``DocXLayout`` is not built and fails loudly.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List

import numpy as np
import torch

from . import lib as L
from .base_infer_task import BaseInferTask
from .engine import HipEngine
from .layout_stage import LayoutStage, PicodetConfig
from .ocr_detection_task import _read_image
from .weights import pack_picodet

__all__ = ["OcrLayoutTask"]


class OcrLayoutTask(BaseInferTask):
    def __init__(self, task="ocr_layout", model="picodet", engine: HipEngine = None, **kwargs):
        super().__init__(task=task, model=model, **kwargs)
        if model not in ["picodet", "DocXLayout"]:
            raise RuntimeError(f"current model is not supported: {model}")
        if model != "picodet":
            raise RuntimeError(f"layout model '{model}' is not built on the HIP engine; only 'picodet' is")
        tt = kwargs.get("task_type", "en")
        self._config = PicodetConfig(task_type="ch" if tt == "zh" else tt)
        self.model_provider = "PaddleOCR"
        self._engine = engine
        try:
            self._config.model_path = self.get_model_name_or_path()
        except Exception:                       # registry lookups for unknown task types: keep the local path
            self._config.model_path = self._task_path
        self._get_inference_model()

    def _construct_model(self, model):
        if self._engine is None:
            self._engine = HipEngine(int(str(self.device).split(":")[-1]) if ":" in str(self.device) else 0)
        ncls = len(self._config.labels)
        if self.synthetic_seed is not None:
            from .synth_weights import picodet_state_dict
            sd = picodet_state_dict(seed=int(self.synthetic_seed), num_classes=ncls)
        else:
            path = os.path.join(str(self._config.model_path), "pytorch_model.bin")
            if not os.path.exists(path):
                raise RuntimeError(f"no PicoDet checkpoint under {self._config.model_path}: the reference would download an ONNX "
                                   "export from the hub (no network here, and the ONNX importer is SURVEY.md section 8f-3); pass "
                                   "task_path=<dir with a PicoDet state_dict> or synthetic_seed=<int>")
            sd = torch.load(path, map_location="cpu", weights_only=True)
        self._engine.load_weights(L.PT_MODEL_PICODET, pack_picodet(sd, ncls))
        self._model = self._predict

    def _build_processor(self):
        self._stage = LayoutStage(self._engine, self._config)

    def _predict(self, images: List[np.ndarray]) -> List[List[Dict]]:
        out = []
        for img in images:            # images of a call may differ in size: one batch per image, like the reference
            out.append(self._stage(torch.from_numpy(np.ascontiguousarray(img)[None]).to(self._engine._tdev))[0])
        return out

    def detect_pages(self, pages: torch.Tensor) -> List[List[Dict]]:
        """batched door: equally sized pages resident on the device"""
        return self._stage(pages)

    def _preprocess(self, inputs, **kwargs):
        if not isinstance(inputs, list):
            inputs = [inputs]
        return {"inputs": [{"image": _read_image(it), "image_file": it if isinstance(it, str) else ""} for it in inputs]}

    def _run_model(self, inputs, **kwargs):
        begin = time.time()
        res, elapse = self.infer({"images": [it["image"] for it in inputs["inputs"]]})
        inputs["results"] = [{"results": r, "elapse": elapse} for r in res]
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs):
        return [r["results"] for r in inputs["results"]]
