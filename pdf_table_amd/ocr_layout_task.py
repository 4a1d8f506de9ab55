"""``OcrLayoutTask`` on the HIP engine -- drop-in for the reference's stage-1 plug-in (model="picodet").

Reference: src/pdftable/model/ocr_pdf/ocr_layout_task.py:30-157.  Same constructor (``task, model, task_type``), same
result: one list per input image of ``{"bbox": ndarray [4] (x1, y1, x2, y2 in source pixels; float64 holding float32-rounded corners divided by the scale, as in the reference), "label", "score",
"category_id"}`` (:125-141, picodet/processor_picodet.py:286-296).  The reference runs an ONNX export of PaddleDetection's
picodet_lcnet_x1_0 layout model; here the in-tree LCNet + CSP-PAN + PicoHead graph (assumed hyper-parameters, see
pdf_table_amd.synth_weights.picodet_state_dict) runs on the engine.  ``DocXLayout`` is not built and fails loudly.

ONNX door (the reference's layout stage is ONNX-only: "pytorch model not support yet!", ocr_layout_task.py:82-123): a ``model.onnx`` /
``inference.onnx`` under ``task_path`` (or ``task_path`` = the file) is parsed by ``pdf_table_amd.onnx_import`` and run layer by layer by
``HipGraphExecutor`` behind the engine's own pre-processing kernel; its 2 L outputs are split like ``get_onnx_output_dict`` (:159-175: first
half per-level scores [B, A_l, classes], second half box distributions [B, A_l, 32]) and decoded by the same host post-processor.
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
            self._engine = self._new_engine()
        ncls = len(self._config.labels)
        self._exec = None
        onnx_path = self._onnx_file()
        if onnx_path is not None:
            from .onnx_exec import HipGraphExecutor
            from .onnx_import import UnsupportedOnnxGraph
            self._exec = HipGraphExecutor(onnx_path, engine=self._engine, precision=self._exec_precision)      # raises UnsupportedOnnxGraph naming an operator without a kernel
            if len(self._exec.outputs) % 2 or not self._exec.outputs:
                raise UnsupportedOnnxGraph(f"{onnx_path}: a PicoDet export returns 2 L tensors (L score maps, then L box distributions), "
                                           f"this graph returns {self._exec.outputs}")
            self._model = self._predict_onnx
            return
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
        # stage_precision="fp32": this stage alone in the pair mode (LayoutStage.precision) -- its blob is then a bf16 blob with the pair tiles whatever
        # the engine's own precision is
        sp = str(self.kwargs.get("stage_precision") or "").lower()
        self._stage_precision = L.PT_PRECISION_BF16X3 if sp in ("fp32", "bf16x3", "float32") else None
        if sp and self._stage_precision is None:
            raise ValueError(f"stage_precision={sp!r}: only 'fp32' (the pair mode for this stage alone) is supported")
        if self._stage_precision is not None:
            self._engine.load_weights(L.PT_MODEL_PICODET, pack_picodet(sd, ncls, x3=True, fmt="bf16"))
        else:
            self._engine.load_weights(L.PT_MODEL_PICODET, pack_picodet(sd, ncls, fmt=self._engine.weight_fmt))
        self._model = self._predict

    def _build_processor(self):
        self._stage = LayoutStage(self._engine, self._config, precision=getattr(self, "_stage_precision", None))

    def _predict(self, images: List[np.ndarray]) -> List[List[Dict]]:
        out = []
        for img in images:            # images of a call may differ in size: one batch per image, like the reference
            out.append(self._stage(torch.from_numpy(np.ascontiguousarray(img)[None]).to(self._engine._tdev))[0])
        return out

    def get_onnx_output_dict(self, outputs):
        """ocr_layout_task.py:159-175: the first half of the graph outputs are the per-level scores, the second half the box distributions
        (the reference's dict keys are kept: ``boxes`` holds the scores, ``boxes_num`` the distributions)"""
        if self._exec is None:
            return None
        half = len(outputs) // 2
        return dict(boxes=list(outputs[:half]), boxes_num=list(outputs[half:half * 2]))

    def _predict_onnx(self, images: List[np.ndarray]) -> List[List[Dict]]:
        out = []
        for img in images:
            out.extend(self.detect_pages(torch.from_numpy(np.ascontiguousarray(img)[None]).to(self._engine._tdev)))
        return out

    def detect_pages(self, pages: torch.Tensor) -> List[List[Dict]]:
        """batched door: equally sized pages resident on the device"""
        if getattr(self, "_exec", None) is None:
            return self._stage(pages)
        cfg = self._config
        ncls = len(cfg.labels)
        x = self._engine.layout_preprocess(pages, cfg.img_height, cfg.img_width)
        acts = self._exec.run_device_graphed(x, 3)      # the layer walk replayed from a captured HIP graph per batch shape
        outs = []
        for a in acts:
            if not a.seq:
                from .onnx_import import UnsupportedOnnxGraph
                raise UnsupportedOnnxGraph(f"layout graph output of shape {a.shape()}: [B, anchors, channels] tensors are expected")
            outs.append(self._exec.values(a)[:, 0].cpu().numpy())
        d = self.get_onnx_output_dict(outs)
        if any(s_.shape[-1] != ncls for s_ in d["boxes"]) or any(b_.shape[-1] != 32 for b_ in d["boxes_num"]):
            from .onnx_import import UnsupportedOnnxGraph
            raise UnsupportedOnnxGraph(f"layout graph outputs {[o.shape for o in outs]}: {ncls} class scores and 32 distribution bins per anchor "
                                       f"are expected for task_type '{cfg.task_type}'")
        return [self._stage.decode_outputs([s_[i] for s_ in d["boxes"]], [b_[i] for b_ in d["boxes_num"]], tuple(pages.shape[1:3]))
                for i in range(pages.shape[0])]

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
