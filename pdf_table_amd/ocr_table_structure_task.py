"""``OcrTableStructureTask`` on the HIP engine -- drop-in for the reference's stage-4 plug-in (model="Lore").

Reference: src/pdftable/model/ocr_pdf/ocr_table_structure_task.py:47-271.  Same constructor (``task, model, task_type``,
``assert`` on the model name :53-54, ``PubTabNet`` -> ``ptn`` :66-67), same result list: one dict per input image with
``polygons`` float32 [n, 8] (cell quads in source pixels), ``logi`` [n, 4] (integer-valued logical locations) and
``inputs`` (TableLorePostProcessor.__call__, lore/processer_lore.py:163-188).  ``model="Lore"`` is served for all three
task types (wtw / ptn: DLA-34 + DCN detector, wireless: ResNet-18 detector); the other structure models the reference lists fail loudly.

Two ways in:
  * reference-shaped: ``task(image_or_list)`` -- path / PIL / ndarray, one table image each;
  * batched: ``task.recognize_tables(pages_gpu, boxes_per_page)`` -- table crops are warped out of resident pages
    on the device; this is what ``OcrTablePipeline`` and ``bench.py`` use.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import lib as L
from .base_infer_task import BaseInferTask
from .engine import HipEngine
from .ocr_detection_task import _read_image
from .tsr_stage import LoreConfig, TsrStage
from .weights import pack_lore_dla34, pack_lore_processor, pack_lore_wireless

__all__ = ["OcrTableStructureTask"]

_MODELS = ["CenterNet", "SLANet", "Lore", "Lgpma", "MtlTabNet", "TableMaster", "LineCell", "LineCellPdf"]


class OcrTableStructureTask(BaseInferTask):
    def __init__(self, task="ocr_table_structure", model="CenterNet", engine: HipEngine = None, **kwargs):
        super().__init__(task=task, model=model, **kwargs)
        assert model in _MODELS
        if model != "Lore":
            raise RuntimeError(f"table-structure model '{model}' is not built on the HIP engine; only 'Lore' is "
                               "(SURVEY.md section 8a stage 4)")
        if self.task_type == "PubTabNet":
            self.task_type = "ptn"
        self._config = LoreConfig(task_type=self.task_type or "wtw")
        self.model_provider = "model_scope"
        self._engine = engine
        self._config.model_path = self.get_model_name_or_path()
        self._get_inference_model()

    def _construct_model(self, model):
        if self._engine is None:
            self._engine = HipEngine(int(str(self.device).split(":")[-1]) if ":" in str(self.device) else 0)
        cfg = self._config
        if self.synthetic_seed is not None:
            from .synth_weights import lore_dla34_state_dict, lore_processor_state_dict, lore_wireless_state_dict
            det_sd = (lore_wireless_state_dict if cfg.backbone == "ResNet-18" else lore_dla34_state_dict)(seed=int(self.synthetic_seed))
            proc_sd = lore_processor_state_dict(seed=int(self.synthetic_seed) + 1, layers=cfg.tsfm_layers,
                                                stacking_layers=cfg.stacking_layers)
        else:
            # LoreModel.load_model (lore/modeling_lore.py:103-123): pytorch_model.pt holds both parts under the
            # prefixes model. / processor.; otherwise model_best.pth + processor_best.pth ('state_dict' entries)
            mp = cfg.model_path
            one = os.path.join(mp, "pytorch_model.pt")
            if os.path.exists(one):
                sd = torch.load(one, map_location="cpu", weights_only=True)["state_dict"]
                det_sd = {k[6:]: v for k, v in sd.items() if k.startswith("model.")}
                proc_sd = {k[10:]: v for k, v in sd.items() if k.startswith("processor.")}
            elif os.path.exists(os.path.join(mp, "model_best.pth")):
                strip = lambda d: {(k[7:] if k.startswith("module") and not k.startswith("module_list") else k): v
                                   for k, v in d.items()}
                det_sd = strip(torch.load(os.path.join(mp, "model_best.pth"), map_location="cpu", weights_only=True)["state_dict"])
                proc_sd = strip(torch.load(os.path.join(mp, "processor_best.pth"), map_location="cpu",
                                           weights_only=True)["state_dict"])
            else:
                raise RuntimeError(f"no Lore checkpoint under {mp}: the reference would download it from the hub (no "
                                   "network here); pass task_path=<dir> or synthetic_seed=<int>")
        if cfg.backbone == "ResNet-18":
            self._engine.load_weights(L.PT_MODEL_LORE_RESNET18, pack_lore_wireless(det_sd))
        else:
            self._engine.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(det_sd))
        self._engine.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(proc_sd))
        self._model = self._predict

    def _build_processor(self):
        # tables per DLA-34 launch chain: 8-table launches leave most of the 256 CUs idle in the coarse levels (measured
        # with the bench's 80); TsrStage balances the last micro-batch (87 tables -> 44 + 43)
        self._stage = TsrStage(self._engine, self._config, micro_batch=int(os.environ.get("PT_TSR_MICROBATCH", "128")))

    def _predict(self, images: List[np.ndarray]) -> List[Dict]:
        """one table image each (RGB ndarray): the whole image is the crop"""
        out = []
        for img in images:
            page = torch.from_numpy(np.ascontiguousarray(img)[None]).to(self._engine._tdev)
            h, w = img.shape[:2]
            out.append(self._stage(page, [np.array([[0, 0, w, h]])])[0][0])
        return out

    def recognize_tables(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray], page_frame: bool = True) -> List[List[Dict]]:
        """tables of resident pages; quads come back in PAGE pixels (page_frame=True), as the reference's system path
        delivers them after shifting every per-crop result by the crop's rounded corner (ocr_system_task.py:190-199 ->
        TableProcessUtils.convert_table_sep_to_merge, pdf_table/table_common.py:1811-1825)"""
        return self._stage(pages, boxes_per_page, page_frame=page_frame)

    def _preprocess(self, inputs, **kwargs):
        if not isinstance(inputs, list):
            inputs = [inputs]
        return {"inputs": [{"image": _read_image(it), "inputs": it} for it in inputs]}

    def _run_model(self, inputs, **kwargs):
        begin = time.time()
        res, elapse = self.infer({"images": [it["image"] for it in inputs["inputs"]]})
        inputs["results"] = [{"results": r, "elapse": elapse, "inputs": it["inputs"]} for r, it in zip(res, inputs["inputs"])]
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs) -> List[Dict]:
        out = []
        for r in inputs["results"]:
            d = {"polygons": r["results"]["polygons"], "logi": r["results"]["logi"]}
            if r["inputs"] is not None:
                d["inputs"] = r["inputs"]
            # the reference adds these only when an output_dir is set (show_results, :262-268); they are cheap here
            if "structure_str_list" in r["results"]:
                from .table_html import table_cells_from_logits
                d["structure_str_list"] = r["results"]["structure_str_list"]
                d["table_cells"] = table_cells_from_logits(d["polygons"], d["logi"])
            out.append(d)
        return out
