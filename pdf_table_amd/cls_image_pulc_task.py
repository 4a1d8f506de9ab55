"""``ClsImagePulcTask`` on the HIP engine -- drop-in for the reference's PP-LCNet classification plug-in.

Reference: src/pdftable/model/ocr_pdf/cls_image_pulc_task.py:25-100.  Same constructor (``task, model, task_type``) and
result: for one input the post-processor's dict (``{"class_ids", "scores", "label_names"}`` or ``{"attributes",
"output"}``), for a list one dict per image.  The reference runs one image per ``infer()``; here a list is one batched
launch.  ``lines(pages, quads)`` is the batched form of OcrSystemTask.text_line_orientation's per-line loop
(ocr_system_task.py:395-439)."""
from __future__ import annotations

import os
import time
from typing import List

import numpy as np
import torch

from . import lib as L
from .base_infer_task import BaseInferTask
from .cls_stage import CLS_TASKS, ClsStage
from .engine import HipEngine
from .ocr_detection_task import _read_image
from .weights import pack_pplcnet

__all__ = ["ClsImagePulcTask"]


class ClsImagePulcTask(BaseInferTask):
    def __init__(self, task="cls_image", model="PPLCNet", task_type="text_image_orientation", engine: HipEngine = None,
                 slot: int = 0, **kwargs):
        super().__init__(task=task, model=model, task_type=task_type, **kwargs)
        if model != "PPLCNet":
            raise RuntimeError(f"current model is not supported: {model}")
        if task_type not in CLS_TASKS:
            raise KeyError(task_type)                # CLS_PULC_TASK_CONFIG[self.task_type] in the reference
        self.task_type = task_type
        self.slot = slot
        self.model_provider = "PaddleOCR"
        self._engine = engine

        class _Cfg:
            backbone = model
            model_path = ""
        self._config = _Cfg()
        self._config.model_path = self.get_model_name_or_path()
        self._get_inference_model()

    def _construct_model(self, model):
        if self._engine is None:
            self._engine = self._new_engine()
        ncls = CLS_TASKS[self.task_type]["class_num"]
        self._exec, self._softmax_in_graph = None, False
        onnx_path = self._onnx_file()
        if onnx_path is not None:
            # the reference's ONNX mode (DeployUtils.prepare_onnx_model, utils/deploy_utils.py:243-280): a PP-LCNet export the importer
            # recognises runs on the dedicated launch graph, any other classifier layer by layer on the generic executor
            from .onnx_import import UnsupportedOnnxGraph, load_onnx, recognise
            graph = load_onnx(onnx_path)
            sd = None
            try:
                arch, sd_ = recognise(graph)
                if arch == "pplcnet":
                    sd = sd_
            except UnsupportedOnnxGraph:
                pass
            if sd is None:
                from .onnx_exec import HipGraphExecutor
                self._exec = HipGraphExecutor(graph, engine=self._engine, precision=self._exec_precision)
                if len(self._exec.outputs) != 1:
                    raise UnsupportedOnnxGraph(f"{onnx_path}: a classifier returns one [B, classes] tensor, this graph returns {self._exec.outputs}")
                self._softmax_in_graph = any(l.op == "act" and l.attrs.get("kind") == "softmax" for l in self._exec.layers[-2:])
                self._model = self._predict_onnx
                return
        elif self.synthetic_seed is not None:
            from .synth_weights import pplcnet_state_dict
            sd = pplcnet_state_dict(seed=int(self.synthetic_seed), class_num=ncls)
        if onnx_path is None and self.synthetic_seed is None:
            path = os.path.join(str(self._config.model_path), "pytorch_model.bin")
            if not os.path.exists(path):
                raise RuntimeError(f"no PP-LCNet checkpoint at {path}: the reference would download "
                                   f"'{self._config.model_path}' from the hub (no network here); pass task_path=<dir> or "
                                   "synthetic_seed=<int>")
            sd = torch.load(path, map_location="cpu", weights_only=True)
            sd = sd.get("state_dict", sd)
        if sd["fc.weight"].shape[0] != ncls:
            raise RuntimeError(f"checkpoint has {sd['fc.weight'].shape[0]} classes, task '{self.task_type}' needs {ncls}")
        self._engine.load_weights(L.PT_MODEL_PPLCNET + self.slot, pack_pplcnet(sd, fmt=self._engine.weight_fmt))
        self._model = self._predict

    def _build_processor(self):
        self._stage = ClsStage(self._engine, self.task_type, self.slot)

    def _predict(self, images: List[np.ndarray]):
        cfg = self._stage.cfg
        return self._engine.cls_forward(images, cfg["size"], self.slot, cfg["textline"])

    def _predict_onnx(self, images: List[np.ndarray]):
        """generic executor behind the engine's Pillow-exact resize + normalise kernel -> logits (or log-probabilities when the graph ends in
        its own Softmax, so that the post-processor's soft-max gives the graph's probabilities back)"""
        cfg = self._stage.cfg
        x = self._engine.cls_preprocess(images, cfg["size"])
        ncls = CLS_TASKS[self.task_type]["class_num"]
        rows = []
        for i in range(x.shape[0]):              # one image per run: an export with static shapes has its batch size baked in (the reference
            (a,) = self._exec.run_device_graphed(x[i:i + 1], 3)      # runs one image per infer() too, cls_image_pulc_task.py:70-84); replayed from a HIP graph
            if not a.flat or a.c != ncls:
                from .onnx_import import UnsupportedOnnxGraph
                raise UnsupportedOnnxGraph(f"classifier output of shape {a.shape()}: [B, {ncls}] is expected for task '{self.task_type}'")
            rows.append(self._exec.values(a)[:, 0, 0].clone())      # (the activation is the graph's own buffer: the next replay overwrites it)
        y = torch.cat(rows, 0)
        return torch.log(y.clamp_min(1e-30)) if self._softmax_in_graph and self.task_type != "table_attribute" else y

    def _preprocess(self, inputs, **kwargs):
        if not isinstance(inputs, list):
            inputs = [inputs]
        return {"inputs": [{"image": _read_image(item)} for item in inputs]}

    def _run_model(self, inputs, **kwargs):
        begin = time.time()
        logits, elapse = self.infer({"images": [b["image"] for b in inputs["inputs"]]})
        inputs["results"] = [{"results": logits, "elapse": elapse}]
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs):
        results = self._stage.post(inputs["results"][0]["results"])
        return results[0] if len(results) == 1 else results

    # ---- batched forms of the reference's per-item loops -------------------------------------------------------------
    def pages(self, pages: torch.Tensor):
        return self._stage.pages(pages)

    def lines(self, pages: torch.Tensor, quads_per_page):
        """quads_per_page: per page an array [k, 8] of detected text boxes -> (results per line, upright?)"""
        from .rec_stage import build_lines
        lines = build_lines(quads_per_page)
        res = self._stage.lines(pages, lines)
        return res, self._stage.orientation_vote(res)
