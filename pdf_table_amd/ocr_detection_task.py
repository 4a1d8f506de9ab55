"""``OcrDetectionTask`` on the HIP engine -- drop-in for the reference's stage-2 plug-in.

Reference: src/pdftable/model/ocr_pdf/ocr_detection_task.py:29-141.  Same constructor
(``task, model in {"db","db_pp"}, backbone, thresh, **kwargs``), same five-method split, same result
(``[np.ndarray (n, 8)]`` per input, x1,y1..x4,y4 in source pixels), same ``RuntimeError`` for an unknown
model name (:58).  What changes is what runs: pages of equal size are batched, the network + bitmap run as
HIP kernels, and the per-contour scoring runs on the device.

Model availability (SURVEY.md finding F2): ``model="db"`` is the in-tree DB-ResNet18 -- fully served.
``model="db_pp"`` selects PP-OCR ONNX graphs that exist neither in the reference tree nor offline; with
``allow_stand_in=True`` the PP-OCR pre/post-processing is run around the DB-ResNet18 network (what the
benchmark does); without it construction fails loudly, naming the hub id that would have been fetched.
"""
from __future__ import annotations

import os
import time
from typing import List

import numpy as np
import torch

from . import lib as L
from .base_infer_task import BaseInferTask
from .det_stage import DetConfig, DetStage
from .engine import HipEngine
from .weights import pack_db_nas, pack_db_resnet18

__all__ = ["OcrDetectionTask"]


class _DetCfg:
    def __init__(self, backbone, thresh, lang="en"):
        self.backbone = backbone
        self.thresh = thresh
        self.lang = lang
        self.model_path = ""


def _read_image(item) -> np.ndarray:
    """str path / PIL image / RGB ndarray -> RGB uint8 ndarray (processor_ocr_db_pp.py:113-122)."""
    if isinstance(item, np.ndarray):
        return item
    import PIL.Image
    if isinstance(item, PIL.Image.Image):
        return np.array(item.convert("RGB"))
    if isinstance(item, str):
        return np.array(PIL.Image.open(item).convert("RGB"))
    raise TypeError(f"inputs should be either str, PIL.Image, np.array, but got {type(item)}")


class OcrDetectionTask(BaseInferTask):
    def __init__(self, task="ocr_detection", model="db", backbone: str = "resnet18", thresh: float = 0.2,
                 engine: HipEngine = None, **kwargs):
        super().__init__(task=task, model=model, **kwargs)
        if model == "db":
            self._config = _DetCfg(backbone, thresh)
            self.model_provider = "model_scope"
        elif model == "db_pp":
            lang = self.lang if self.lang in ["ch", "en"] else "ml"
            self._config = _DetCfg(backbone, thresh, lang)
            self.model_provider = "PaddleOCR"
        else:
            raise RuntimeError(f"current model is not supported: {model}")
        self._engine = engine
        self._det_cfg = DetConfig(flavour=model, thresh=thresh,
                                  box_thresh=kwargs.get("box_thresh", 0.6), unclip_ratio=kwargs.get("unclip_ratio", 1.5),
                                  use_dilation=kwargs.get("use_dilation", False))
        self._config.model_path = self.get_model_name_or_path()
        self._get_inference_model()

    def _construct_model(self, model):
        if self._engine is None:
            self._engine = self._new_engine()
        onnx_path = self._onnx_file()
        if model == "db_pp" and onnx_path is None and not self.kwargs.get("allow_stand_in", False):
            raise RuntimeError(f"'{self._config.model_path}' is an ONNX graph that is not part of the reference tree; pass "
                               "task_path=<dir with model.onnx> (imported by pdf_table_amd.onnx_import when its architecture "
                               "is one the engine runs) or allow_stand_in=True to run the PP-OCR pre/post-processing around "
                               "DB-ResNet18")
        if onnx_path is not None:
            # the reference's ONNX mode (BaseInferTask._prepare_onnx_mode, base_infer_task.py:139-144 ->
            # DeployUtils.prepare_onnx_model, utils/deploy_utils.py:243-280): parse the graph, map it to the engine's
            # DB-ResNet18 launch graph, load its weights; any other architecture raises with the graph's inventory
            from .onnx_import import UnsupportedOnnxGraph, load_onnx, recognise
            graph = load_onnx(onnx_path)
            try:
                arch, sd = recognise(graph)
            except UnsupportedOnnxGraph:
                arch, sd = "generic", None
            if arch == "db_resnet18":
                self._engine.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd, fmt=self._engine.weight_fmt))
            elif arch == "generic":
                # an architecture without a dedicated launch graph (the real PP-OCR detectors: PP-LCNetV3 / MobileNetV3 +
                # RSE-FPN + DB head): the layer list runs operator by operator between the engine's pre-processing and
                # its bitmap / box-score kernels; an operator without a kernel is named when the first batch reaches it
                from .onnx_exec import HipGraphExecutor
                ex = HipGraphExecutor(graph, engine=self._engine, precision=self._exec_precision)
                if len(ex.outputs) != 1:
                    raise UnsupportedOnnxGraph(f"{onnx_path}: a text detector returns one probability map, this graph returns {ex.outputs}")

                def net(x4, _ex=ex):
                    (a,) = _ex.run_device_graphed(x4, 3)      # one captured HIP graph per input shape (eager beyond eight shapes)
                    if a.c != 1:
                        raise UnsupportedOnnxGraph(f"{onnx_path}: the output has {a.c} channels, a probability map has one")
                    return _ex.values(a)[..., 0].clone()      # (the graph's own buffer: the next replay overwrites it)
                self._net = net
            else:
                raise UnsupportedOnnxGraph(f"{onnx_path} is a '{arch}' network, not a text detector the engine runs")
            self._model = self._predict
            return
        nas = model == "db" and self._config.backbone == "proxylessnas"      # DBNasModel, modeling_db_net.py:47-49
        if model == "db" and self._config.backbone not in ("resnet18", "proxylessnas"):
            raise TypeError(f"detector backbone should be either resnet18, proxylessnas, but got {self._config.backbone}")
        if self.synthetic_seed is not None:
            from .synth_weights import db_nas_state_dict, db_resnet18_state_dict
            sd = (db_nas_state_dict if nas else db_resnet18_state_dict)(seed=int(self.synthetic_seed))
        else:
            path = os.path.join(self._config.model_path, "pytorch_model.pt")   # modeling_db_net.py:53-56
            if not os.path.exists(path):
                raise RuntimeError(f"no checkpoint at {path}: the reference would download "
                                   f"'{self._config.model_path}' from the hub (no network here); pass task_path=<dir> "
                                   "or synthetic_seed=<int>")
            sd = torch.load(path, map_location="cpu", weights_only=True)
        if nas:
            self._engine.load_weights(L.PT_MODEL_DB_NAS, pack_db_nas(sd, fmt=self._engine.weight_fmt))
        else:
            self._engine.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd, fmt=self._engine.weight_fmt))
        self._model = self._predict

    def _onnx_file(self):
        """model.onnx / fp16_model.onnx under task_path (prepare_onnx_model's layout), or task_path itself if it is a file"""
        tp = getattr(self, "_task_path", None)
        if not tp or self.synthetic_seed is not None:
            return None
        if os.path.isfile(tp) and tp.endswith(".onnx"):
            return tp
        # a directory: the ONNX route is the reference's for model="db_pp" (BaseInferTask._prepare_onnx_mode); for model="db" with a
        # pytorch_model.pt beside an exported graph the checkpoint wins unless the caller asks (use_onnx=True)
        if self.model == "db" and not self.kwargs.get("use_onnx", False) and os.path.isfile(os.path.join(tp, "pytorch_model.pt")):
            return None
        for cand in ("model.onnx", "fp16_model.onnx", "inference.onnx"):
            if os.path.isfile(os.path.join(tp, cand)):
                return os.path.join(tp, cand)
        return None

    def _build_processor(self):
        self._stage = DetStage(self._engine, self._det_cfg, net=getattr(self, "_net", None))

    def _predict(self, pages: torch.Tensor):
        return self._stage.forward(pages)

    def _preprocess(self, inputs, **kwargs):
        if not isinstance(inputs, list):
            inputs = [inputs]
        batch = []
        for item in inputs:
            img = _read_image(item)
            batch.append({"image": img, "org_shape": img.shape, "inputs": item})
        return {"inputs": batch}

    def _run_model(self, inputs, **kwargs):
        items = inputs["inputs"]
        begin = time.time()
        results = [None] * len(items)
        # group pages of identical size into one device batch (the reference runs them one by one)
        groups = {}
        for i, it in enumerate(items):
            groups.setdefault(it["image"].shape, []).append(i)
        for shape, idxs in groups.items():
            pages = torch.from_numpy(np.stack([items[i]["image"] for i in idxs])).to(self._engine._tdev)
            (prob, bitmap, ev), elapse = self.infer({"pages": pages})
            boxes = self._stage.boxes(prob, bitmap, shape[:2], ev)
            for k, i in enumerate(idxs):
                results[i] = {"results": boxes[k], "elapse": elapse, "org_shape": items[i]["org_shape"],
                              "inputs": items[i]["inputs"]}
        inputs["results"] = results
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs) -> List[np.ndarray]:
        return [r["results"] for r in inputs["results"]]
