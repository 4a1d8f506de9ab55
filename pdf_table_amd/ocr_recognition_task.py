"""``OcrRecognitionTask`` on the HIP engine -- drop-in for the reference's stage-3 plug-in.

Reference: src/pdftable/model/ocr_pdf/ocr_recognition_task.py:28-136.  Same constructor (``task, model, task_type``),
same result (one string per input crop), same ``RuntimeError`` for an unknown model (:47).  ``model="CRNN"`` and
``model="ConvNextViT"`` (the two in-tree torch recognisers of BASELINE.json's configs; ConvNextViT with the chunking
pre-processor: 32 x 804, three 300-px chunks, 201 tokens per line, vocabulary from class 2) are served from checkpoints.
``model="PP-OCRv4" / "PP-OCRv3" / "PP-Table"`` -- the ONNX recognisers ``fix_model_names()`` selects for every language
(model/ocr_pdf/configuration_ocr_document.py:138-141) -- are served from a ``model.onnx`` / ``inference.onnx`` under ``task_path``: the
engine's ``PPOcrRecPreProcessor`` kernel (48-px, width-sorted mini-batches), the generic graph executor (conv backbone, SVTR-type
attention / LayerNorm neck, CTC head with its Softmax: pdf_table_amd/onnx_exec.py) and ``CTCLabelDecode``; without a file they fail
loudly, naming the hub id the reference would have fetched.  ``LightweightEdge`` is not built.

Two ways in:
  * reference-shaped: ``task(crop_or_list_of_crops)`` -- every crop is an RGB image (path / PIL / ndarray) exactly as
    ``OcrSystemTask.text_recognition`` passes it (ocr_system_task.py:309-320); all crops of the call go through one
    ``pt_rec_forward_crops`` (ragged batch);
  * batched: ``task.recognize_quads(pages_gpu, boxes_per_page)`` -- quads are cut out of the resident pages on the
    device (no host crop at all); this is what ``OcrTablePipeline`` and ``bench.py`` use.
"""
from __future__ import annotations

import os
import time
from typing import List, Sequence

import numpy as np
import torch

from . import lib as L
from .base_infer_task import BaseInferTask
from .engine import HipEngine
from .ocr_detection_task import _read_image
from .rec_stage import RecStage
from .weights import pack_convnext_vit, pack_crnn

__all__ = ["OcrRecognitionTask"]


class _RecCfg:
    def __init__(self, recognizer, task_type):
        self.recognizer = recognizer
        self.task_type = "general" if recognizer in ("CRNN", "LightweightEdge") else task_type
        self.backbone = recognizer
        self.model_path = ""


class OcrRecognitionTask(BaseInferTask):
    def __init__(self, task="ocr_recognition", model="CRNN", task_type="document", engine: HipEngine = None, **kwargs):
        super().__init__(task=task, model=model, **kwargs)
        if model in ["ConvNextViT", "CRNN", "LightweightEdge"]:
            self._config = _RecCfg(model, task_type)
            self.model_provider = "model_scope"
        elif model in ["PP-OCRv4", "PP-OCRv3", "PP-Table"]:
            self._config = _RecCfg(model, task_type)
            self.model_provider = "PaddleOCR"
        else:
            raise RuntimeError(f"current model is not supported: {model}")
        self._engine = engine
        self._config.model_path = self.get_model_name_or_path()
        self._get_inference_model()

    def _construct_pp(self):
        """ONNX mode of the PP-OCR recognisers (ocr_recognition_task.py:81-116 -> DeployUtils.prepare_onnx_model)"""
        from .onnx_exec import HipGraphExecutor
        from .rec_postprocess import CTCLabelDecode
        from .rec_pp_stage import PPOcrRecConfig, PPOcrRecPreProcessor
        onnx_path = self._onnx_file()
        if onnx_path is None:
            raise RuntimeError(f"recogniser '{self.model}': no model.onnx / inference.onnx under {self._task_path!r} -- the reference would "
                               f"download '{self._config.model_path}' from the hub (no network here); pass task_path=<dir or file>")
        if self._engine is None:
            self._engine = self._new_engine()
        self._exec = HipGraphExecutor(onnx_path, engine=self._engine, precision=self._exec_precision)
        if len(self._exec.outputs) != 1:
            from .onnx_import import UnsupportedOnnxGraph
            raise UnsupportedOnnxGraph(f"{onnx_path}: a CTC recogniser returns one [B, T, classes] tensor, this graph returns {self._exec.outputs}")
        # a static export ([1, 3, 48, 320] like the shipped PP-OCR files) has its width baked into the Reshape constants: the pre-processor then
        # runs PaddleOCR's static-shape setting of the same config fields (rec_image_shape = the graph's, limited_max_width = its width: lines
        # wider than imgW are resized to imgW, resize_norm_img processor_ocr_rec_pp.py:43-58); a dynamic-width graph keeps the defaults
        shp = list(getattr(self._exec.inputs[0], "shape", []) or [])
        static = len(shp) == 4 and all(isinstance(d, int) and d > 0 for d in shp[1:])
        self._batch1 = len(shp) == 4 and isinstance(shp[0], int) and shp[0] == 1       # batch size baked in: one walk per line
        cfg = PPOcrRecConfig(rec_image_shape=f"{shp[1]}, {shp[2]}, {shp[3]}", limited_max_width=int(shp[3])) if static else PPOcrRecConfig()
        self._pp = PPOcrRecPreProcessor(cfg, engine=self._engine)
        d = self.kwargs.get("character_dict_path")
        if d is None:
            base = onnx_path if os.path.isdir(onnx_path) else os.path.dirname(onnx_path)
            d = next((os.path.join(base, c) for c in ("ppocr_keys_v1.txt", "en_dict.txt", "dict.txt", "vocab.txt") if os.path.isfile(os.path.join(base, c))), None)
        self._ctc = CTCLabelDecode(d, use_space_char=bool(self.kwargs.get("use_space_char", True)))
        self._vocab = None
        self.last_scores: List[float] = []
        self._model = self._predict_pp

    def _predict_pp(self, crops):
        """crops -> PPOcrRecPreProcessor mini-batches on the device -> graph -> CTCLabelDecode; results in the order of the crops (the
        reference's PPOcrRecPostProcessor scatters correctly only for one crop per call, processor_ocr_rec_pp.py:149-171: not reproduced)"""
        from .onnx_import import UnsupportedOnnxGraph
        texts, scores = [""] * len(crops), [0.0] * len(crops)
        for b in self._pp(list(crops)):
            img = b["image"]                                     # f32 [n, 3, 48, imgW] on the device
            # static exports have their batch size baked in: one graph walk per line, the mini-batch's walks captured into one HIP graph
            x = img.permute(0, 2, 3, 1).contiguous()
            if not self._exec.split:                         # the tolerance mode takes the fp32 image and splits it into (hi, lo) itself
                x = x.to(self._exec.adt)
            outs = self._exec.run_lines_graphed(x, 3) if self._batch1 else [self._exec.run_device_graphed(x, 3)]    # dynamic batch: one walk
            probs = []
            for (a,) in outs:
                if not a.seq or a.c != len(self._ctc.character):
                    raise UnsupportedOnnxGraph(f"recogniser output of shape {a.shape()}: [B, T, {len(self._ctc.character)}] (blank + dictionary"
                                               " + space) is expected")
                v = self._exec.values(a)                         # fp32 probabilities [b, 1, T, classes] (the executor keeps a final Softmax in fp32)
                probs.append(v.reshape(v.shape[0], v.shape[-2], v.shape[-1]))
            conf, ids = torch.cat(probs).max(-1)
            # one device -> host copy per mini-batch (all lines of a mini-batch share imgW, hence T)
            conf_h, ids_h = conf.cpu().numpy(), ids.cpu().numpy()
            for i, (text, sc) in enumerate(self._ctc.decode_ids(ids_h, conf_h)):
                k = int(b["indices"][b["batch_beg_img_no"] + i])
                texts[k], scores[k] = text, float(sc)
        self.last_scores = scores
        return texts

    def _construct_model(self, model):
        if model in ("PP-OCRv4", "PP-OCRv3", "PP-Table"):
            return self._construct_pp()
        if model not in ("CRNN", "ConvNextViT"):
            raise RuntimeError(f"recogniser '{model}' ({self._config.model_path}) is not built on the HIP engine yet; "
                               "only the in-tree CRNN and ConvNextViT are (SURVEY.md section 8f)")
        if self._engine is None:
            self._engine = self._new_engine()
        vocab = None
        if self.synthetic_seed is not None:
            from .synth_weights import convnext_vit_state_dict, crnn_state_dict
            sd = (convnext_vit_state_dict if model == "ConvNextViT" else crnn_state_dict)(seed=int(self.synthetic_seed))
        elif model == "CRNN" and any(os.path.isfile(os.path.join(self._config.model_path, c)) for c in ("model.onnx", "fp16_model.onnx")):
            # an exported CRNN (DeployUtils.export_onnx -> torch.onnx.export, utils/deploy_utils.py:197-224): the importer
            # restores the state_dict (ONNX LSTM gate order i, o, f, c -> torch's i, f, g, o) and the usual path takes over
            from .onnx_import import UnsupportedOnnxGraph, load_onnx, recognise
            mp = self._config.model_path
            arch, sd = recognise(load_onnx(mp))
            if arch != "crnn":
                raise UnsupportedOnnxGraph(f"the ONNX graph under {mp} is a '{arch}' network, not a recogniser the engine runs")
            vp = os.path.join(mp, "vocab.txt")
            if os.path.exists(vp):
                with open(vp, "r", encoding="utf-8") as f:
                    vocab = [ln.strip("\n") for ln in f.readlines()]
        else:
            mp = self._config.model_path
            path = os.path.join(mp, "pytorch_model.bin")
            if not os.path.exists(path):
                path = os.path.join(mp, "pytorch_model.pt")          # modeling_ocr_recognition.py:102-105
            if not os.path.exists(path):
                raise RuntimeError(f"no checkpoint under {mp}: the reference would download it from the hub (no network "
                                   "here); pass task_path=<dir> or synthetic_seed=<int>")
            raw = torch.load(path, map_location="cpu", weights_only=True)
            sd = {k.replace("recognizer.", "").replace("module.", ""): v for k, v in raw.items()}   # :108-111
            with open(os.path.join(mp, "vocab.txt"), "r", encoding="utf-8") as f:
                vocab = [ln.strip("\n") for ln in f.readlines()]
        if model == "ConvNextViT":
            self._engine.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(sd, fmt=self._engine.weight_fmt))
        else:
            self._engine.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd, fmt=self._engine.weight_fmt))
        self._vocab = vocab
        self._model = self._predict

    def _build_processor(self):
        if self.model in ("PP-OCRv4", "PP-OCRv3", "PP-Table"):
            self._stage = None
            return
        self._stage = RecStage(self._engine, self._vocab, recognizer=self._config.recognizer)

    def _predict(self, crops):
        """already-cropped line images: resize + network + arg-max on the device, CTC collapse + vocabulary on the host"""
        from .rec_stage import ctc_collapse
        if self._config.recognizer == "ConvNextViT":
            ids, _ = self._engine.rec_cvit_forward_crops(crops)
        else:
            ids, _ = self._engine.rec_forward_crops(crops)
        toks = ctc_collapse(ids.cpu().numpy()) if len(crops) else []
        return ["".join(self._stage.label.get(t, "") for t in row) for row in toks]

    def recognize_quads(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray]) -> List[List[str]]:
        return self._stage(pages, boxes_per_page)

    def _preprocess(self, inputs, **kwargs):
        if not isinstance(inputs, list):
            inputs = [inputs]
        return {"inputs": [{"image": _read_image(it)} for it in inputs]}

    def _run_model(self, inputs, **kwargs):
        begin = time.time()
        crops = [it["image"] for it in inputs["inputs"]]
        (texts), elapse = self.infer({"crops": crops})
        inputs["results"] = [{"results": t, "elapse": elapse} for t in texts]
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs) -> List[str]:
        return [r["results"] for r in inputs["results"]]
