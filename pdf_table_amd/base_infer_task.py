"""``BaseInferTask`` -- the stage plug-in contract of the reference, with a ``"hip"`` predictor.

Mirrors src/pdftable/model/ocr_pdf/base_infer_task.py:30-125 (constructor kwargs, the five abstract
methods), :171-303 (model-id resolution against ``TABLE_MODEL_DICT``), :311-315 (``__call__`` =
``_preprocess -> _run_model -> _postprocess``) and :366-381 (``infer`` dispatch on ``_predictor_type``).

Differences, all deliberate:
  * ``predictor_type`` defaults to ``"hip"``: ``infer()`` hands the batch to the MI355X engine
    (libpdftable_hip.so) instead of PyTorch-eager / onnxruntime.  ``"pytorch"`` and ``"onnx"`` raise: this
    build carries no CPU or framework fallback (a missing engine must fail loudly).  ``"trt"`` raises the
    reference's own message (base_infer_task.py:143-144).
  * weights: there is no hub access offline.  ``task_path=<dir>`` is honoured like the reference's custom
    model path (base_infer_task.py:81-83); ``synthetic_seed=<int>`` loads the seeded random-init checkpoint
    with the reference's state_dict layout; otherwise ``get_model_name_or_path()`` returns the hub id the
    reference would download and construction stops with a clear error.
  * default ``precision`` is ``"bf16"`` (the reference's is ``"fp16"``, base_infer_task.py:56); ``precision="fp16"`` selects the engine's
    single-pass IEEE-half mode (PT_PRECISION_F16), ``"fp32"`` the three-pass pair mode (PT_PRECISION_BF16X3).
"""
from __future__ import annotations

import math
import os
import time
from abc import ABCMeta, abstractmethod
from multiprocessing import cpu_count
from typing import Dict

from .ocr_table_model_config import TABLE_MODEL_DICT

__all__ = ["BaseInferTask"]


class BaseInferTask(metaclass=ABCMeta):
    def __init__(self, model, task, priority_path=None, **kwargs):
        self.model = model
        self.is_static_model = kwargs.get("is_static_model", False)
        self.model_provider = kwargs.get("model_provider", "model_scope")
        self.task = task
        self.kwargs = kwargs
        self._priority_path = priority_path
        self._usage = ""
        self._model = None
        self._pre_processor = None
        self._post_processor = None
        self._config = None
        self._custom_model = False
        self._num_threads = kwargs.get("num_threads", math.ceil(cpu_count() / 2))
        self._infer_precision = kwargs.get("precision", "bf16")
        # arithmetic of the in-tree networks on an engine this task creates itself (_new_engine): the reference's "fp16" (its default,
        # base_infer_task.py:56-57 -> model.half(), utils/deploy_utils.py:227-240) is the engine's single-pass IEEE-half mode, "fp32" the
        # three-pass pair mode that holds 1e-3 against an fp32 run, anything else bf16.  A shared engine keeps the precision its owner set.
        _p = str(self._infer_precision).lower()
        self._engine_precision = "f16" if _p in ("fp16", "f16", "half", "float16") else "bf16x3" if _p in ("fp32", "bf16x3", "float32") else "bf16"
        self._exec_precision = self._engine_precision       # the generic ONNX executor computes in the engine's arithmetic
        self._predictor_type = kwargs.get("predictor_type", "hip")
        self._home_path = kwargs.get("home_path", os.path.expanduser("~/.cache/pdftable/outputs"))
        self._task_flag = kwargs.get("task_flag", self.model)
        self.device = kwargs.get("device", "cuda:0")
        self.output_dir = kwargs.get("output_dir", None)
        self.debug = kwargs.get("debug", True)
        self.lang = kwargs.get("lang", "en")
        self.task_type = kwargs.get("task_type", "wtw")
        self.server_model = kwargs.get("server_model", False)
        self.use_modelscope_hub = kwargs.get("use_modelscope_hub", False)
        self.synthetic_seed = kwargs.get("synthetic_seed", None)
        if "task_path" in kwargs:
            self._task_path = kwargs["task_path"]
            self._custom_model = True
        elif self._priority_path:
            self._task_path = os.path.join(self._home_path, "taskflow", self._priority_path)
        else:
            self._task_path = os.path.join(self._home_path, "taskflow", self.task, self.model)
        self.model_dict = TABLE_MODEL_DICT
        self.predictor = None

    # ---- the five methods a stage implements (base_infer_task.py:95-125) --------------------------
    @abstractmethod
    def _construct_model(self, model):
        """Load weights into the engine."""

    @abstractmethod
    def _build_processor(self):
        """Create pre/post processors."""

    @abstractmethod
    def _preprocess(self, inputs, **kwargs):
        """-> {"inputs": [per-item dict, ...]}"""

    @abstractmethod
    def _run_model(self, inputs, **kwargs):
        """-> inputs + {"results": [...], "use_time": seconds}"""

    @abstractmethod
    def _postprocess(self, inputs, **kwargs):
        """-> list of per-item results"""

    def _new_engine(self):
        """the task's own engine on ``self.device`` in the arithmetic ``precision=`` asked for; weight blobs are then packed for it
        (``fmt=engine.weight_fmt``)"""
        from . import lib as L
        from .engine import HipEngine
        eng = HipEngine(int(str(self.device).split(":")[-1]) if ":" in str(self.device) else 0)
        eng.set_precision({"f16": L.PT_PRECISION_F16, "bf16x3": L.PT_PRECISION_BF16X3, "bf16": L.PT_PRECISION_BF16}[self._engine_precision])
        return eng

    # ---- predictor preparation (base_infer_task.py:127-169) -----------------------------------------
    def _prepare_hip_mode(self):
        self.predictor = self._model

    def _prepare_trt_mode(self):
        raise RuntimeError("TensorRt infer not supported!!!")

    def _get_inference_model(self):
        if self._predictor_type == "hip":
            self._construct_model(self.model)
            self._build_processor()
            self._prepare_hip_mode()
        elif self._predictor_type == "trt":
            self._prepare_trt_mode()
        elif self._predictor_type in ("pytorch", "onnx", "other"):
            raise RuntimeError(f"predictor_type={self._predictor_type!r} is not available in the MI355X build: "
                               "the stage runs on the HIP engine only (predictor_type='hip'), there is no fallback")
        else:
            raise RuntimeError(f"unknown predictor_type {self._predictor_type!r}")

    # ---- model-id resolution (base_infer_task.py:171-303) ---------------------------------------------
    def get_model_id(self, config: Dict):
        if config is None:
            return None
        if self.server_model and not self.use_modelscope_hub and "hf_server_model" in config:
            return config["hf_server_model"]
        if self.server_model and "server_model" in config:
            return config["server_model"]
        if not self.use_modelscope_hub and "hf_model" in config:
            return config["hf_model"]
        return config["model"]

    def _scope_or_other_config(self):
        md = self.model_dict[self.model_provider]
        if self.task == "ocr_detection":
            return md["detection"][self._config.backbone]["general"]
        if self.task == "ocr_recognition":
            return md["recognition"][self._config.recognizer][self._config.task_type]
        if self.task == "ocr_table_structure":
            return md["table_structure"][self._config.model_name][self._config.task_type]
        if self.task == "ocr_layout":
            return md["layout"][self._config.model_name][self._config.task_type]
        return None

    def get_model_path_from_model_scope(self):
        return self.get_model_id(self._scope_or_other_config())

    def get_model_path_from_other(self):
        return self.get_model_id(self._scope_or_other_config())

    def get_model_path_from_paddleocr(self):
        md = self.model_dict[self.model_provider]
        backbone = self._config.backbone
        lang = self.lang
        name = {"ocr_detection": "detection", "ocr_recognition": "recognition", "ocr_table_structure": "table_structure",
                "cls_image": "cls_image", "ocr_layout": "layout"}[self.task]
        if name == "recognition" and backbone in ["PP-OCRv4"] and lang not in ["ch", "en"]:
            backbone = "PP-OCRv3"
        raw = md[name][backbone]
        if name == "detection" and lang not in ["ch", "en", "ml"]:
            lang = "ml"
        elif name == "recognition" and lang not in ["ch", "en", "chinese_cht", "korean", "japan"]:
            lang = "en"
        elif name in ["cls_image", "layout"]:
            lang = self.task_type
        return self.get_model_id(raw.get(lang, "en"))

    def get_model_name_or_path(self):
        """Local directory if weights are there, else the hub id the reference would fetch."""
        if self._custom_model or os.path.exists(os.path.join(self._task_path, "pytorch_model.bin")) \
                or os.path.exists(os.path.join(self._task_path, "pytorch_model.pt")):
            return self._task_path
        if self.model_provider == "model_scope":
            return self.get_model_path_from_model_scope()
        if self.model_provider == "PaddleOCR":
            return self.get_model_path_from_paddleocr()
        if self.model_provider == "Other":
            return self.get_model_path_from_other()
        return self._task_path

    def _onnx_file(self):
        """model.onnx / fp16_model.onnx / inference.onnx under task_path (DeployUtils.prepare_onnx_model's layout, utils/deploy_utils.py:
        243-280), or task_path itself if it is an .onnx file; None when the task was given seeded weights"""
        tp = getattr(self, "_task_path", None)
        if not tp or self.synthetic_seed is not None:
            return None
        if os.path.isfile(tp) and str(tp).endswith(".onnx"):
            return tp
        for cand in ("model.onnx", "fp16_model.onnx", "inference.onnx"):
            if os.path.isfile(os.path.join(tp, cand)):
                return os.path.join(tp, cand)
        return None

    # ---- run -----------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        inputs = self._preprocess(*args, **kwargs)
        outputs = self._run_model(inputs, **kwargs)
        return self._postprocess(outputs, **kwargs)

    def infer(self, input_dict: dict, generate=False):
        """(result, elapsed seconds) -- base_infer_task.py:366-381 with the HIP engine as the predictor."""
        start = time.time()
        if self._predictor_type == "hip":
            result = self.predictor(**input_dict)
        elif self._predictor_type == "trt":
            self._prepare_trt_mode()
        else:
            raise RuntimeError(f"predictor_type={self._predictor_type!r} has no runtime in this build")
        return result, time.time() - start

    def help(self):
        print("Examples:\n{}".format(self._usage))
