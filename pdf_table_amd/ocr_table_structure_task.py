"""``OcrTableStructureTask`` on the HIP engine -- drop-in for the reference's stage-4 plug-in (model="Lore", "MtlTabNet", "TableMaster").

Reference: src/pdftable/model/ocr_pdf/ocr_table_structure_task.py:47-271.  Same constructor (``task, model, task_type``,
``assert`` on the model name :53-54, ``PubTabNet`` -> ``ptn`` :66-67), same result list: one dict per input image with
``polygons`` float32 [n, 8] (cell quads in source pixels), ``logi`` [n, 4] (integer-valued logical locations) and
``inputs`` (TableLorePostProcessor.__call__, lore/processer_lore.py:163-188).  ``model="Lore"`` is served for all three
task types (wtw / ptn: DLA-34 + DCN detector, wireless: ResNet-18 detector).  ``model="MtlTabNet"`` (BASELINE.json configs[4], SURVEY.md
section 8f-4: ResNet-GC backbone + three KV-cached decoders on the engine, label convertor + HTML post-processor on the host,
``pdf_table_amd/mtl_stage.py``) returns what ``MtlTabNetPostProcessor.__call__`` returns (model/mtl_tabnet/processor_mtl_tabnet.py:108-131):
``polygons`` int32 [n, 8], ``structure_str_list``, ``structure_str``, ``html_context``, ``inputs``.  ``model="TableMaster"`` (table_master_config.py,
``TableMasterDecoder`` master_decoder.py:532-645): the same backbone, layers and host half without the cell-content decoder (the blob says "0 cell
classes"; ``TableMasterConvertor``).  The other structure models the reference lists fail loudly.

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
from .mtl_stage import MtlStage, MtlTabNetConvertor, MtlTabnetConfig, TableMasterConvertor
from .tsr_stage import LoreConfig, TsrStage
from .weights import pack_lore_dla34, pack_lore_processor, pack_lore_wireless, pack_mtl_backbone, pack_mtl_decoder

__all__ = ["OcrTableStructureTask"]

_MODELS = ["CenterNet", "SLANet", "Lore", "Lgpma", "MtlTabNet", "TableMaster", "LineCell", "LineCellPdf"]


class OcrTableStructureTask(BaseInferTask):
    def __init__(self, task="ocr_table_structure", model="CenterNet", engine: HipEngine = None, **kwargs):
        super().__init__(task=task, model=model, **kwargs)
        assert model in _MODELS
        if model not in ("Lore", "MtlTabNet", "TableMaster"):
            raise RuntimeError(f"table-structure model '{model}' is not built on the HIP engine; 'Lore', 'MtlTabNet' and 'TableMaster' are "
                               "(SURVEY.md section 8a stage 4, 8f-4)")
        self._engine = engine
        if model in ("MtlTabNet", "TableMaster"):
            self._config = MtlTabnetConfig(model_name=model, task_type=self.task_type)
            # sequence limits are configuration (mtl_tabnet_config.py:12-18); tests shorten them
            self._config.max_seq_len = int(kwargs.get("max_seq_len", self._config.max_seq_len))
            self._config.max_seq_len_cell = int(kwargs.get("max_seq_len_cell", self._config.max_seq_len_cell))
            self.model_provider = self._config.model_provider
            self._config.model_path = self.get_model_name_or_path()
            self._get_inference_model()
            return
        if self.task_type == "PubTabNet":
            self.task_type = "ptn"
        self._config = LoreConfig(task_type=self.task_type or "wtw")
        self.model_provider = "model_scope"
        self._config.model_path = self.get_model_name_or_path()
        self._get_inference_model()

    def _construct_model(self, model):
        if self._engine is None:
            self._engine = self._new_engine()
        cfg = self._config
        if model in ("MtlTabNet", "TableMaster"):
            return self._construct_mtl()
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
            self._engine.load_weights(L.PT_MODEL_LORE_RESNET18, pack_lore_wireless(det_sd, fmt=self._engine.weight_fmt))
        else:
            self._engine.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(det_sd, fmt=self._engine.weight_fmt))
        self._engine.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(proc_sd, fmt=self._engine.weight_fmt))
        self._model = self._predict

    def _construct_mtl(self):
        """``build(model config) + load_checkpoint`` (ocr_table_structure_task.py:106-113): one mmcv-style checkpoint, keys ``backbone.*``
        and ``decoder.*`` (``pytorch_model.bin`` in a directory, or a ``.pth`` / ``.bin`` file; ``state_dict`` entry if present --
        table/lgpma/checkpoint.py:39-53).  The network was trained on cv2-read (BGR) images and this engine's pages are RGB: conv1's input
        channels are swapped once at load time, which is exact."""
        cfg = self._config
        master = self.model == "TableMaster"      # table_master_config.py: the same backbone and layers, TableMasterDecoder (no cell-content decoder)
        self._convertor = TableMasterConvertor(max_seq_len=cfg.max_seq_len) if master else \
            MtlTabNetConvertor(max_seq_len=cfg.max_seq_len, max_seq_len_cell=cfg.max_seq_len_cell)
        if self.synthetic_seed is not None:
            from .synth_weights import mtl_tabnet_backbone_state_dict, mtl_tabnet_decoder_state_dict, table_master_decoder_state_dict
            bb = mtl_tabnet_backbone_state_dict(seed=int(self.synthetic_seed))
            dec = table_master_decoder_state_dict(seed=int(self.synthetic_seed) + 1, num_classes=self._convertor.num_classes()) if master else \
                mtl_tabnet_decoder_state_dict(seed=int(self.synthetic_seed) + 1, num_classes=self._convertor.num_classes(),
                                              num_classes_cell=self._convertor.num_classes_cell())
        else:
            mp = cfg.model_path
            f = mp if str(mp).endswith((".pth", ".bin")) else os.path.join(mp, "pytorch_model.bin")
            if not os.path.exists(f):
                raise RuntimeError(f"no MtlTabNet checkpoint at {f}: the reference would download it from the hub (no network here); "
                                   "pass task_path=<dir or file> or synthetic_seed=<int>")
            ck = torch.load(f, map_location="cpu", weights_only=True)
            sd = ck["state_dict"] if "state_dict" in ck else ck
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
            bb = {k[9:]: v for k, v in sd.items() if k.startswith("backbone.")}
            dec = {k[8:]: v for k, v in sd.items() if k.startswith("decoder.")}
            if not bb or not dec:
                raise RuntimeError(f"{f} holds no 'backbone.' / 'decoder.' tensors (an MtlTabNet checkpoint has both)")
            if master == ("cell_fc.weight" in dec):
                raise RuntimeError(f"{f} is {'an MtlTabNet' if master else 'a TableMaster'} checkpoint (cell-content decoder "
                                   f"{'present' if master else 'absent'}); model='{self.model}' was asked for")
            ncell = dec["cell_fc.weight"].shape[0] if not master else 0
            if dec["cls_fc.weight"].shape[0] != self._convertor.num_classes() or ncell != self._convertor.num_classes_cell():
                raise RuntimeError("the checkpoint's class counts do not match the PubTabNet vocabularies "
                                   f"({dec['cls_fc.weight'].shape[0]} / {ncell} vs "
                                   f"{self._convertor.num_classes()} / {self._convertor.num_classes_cell()})")
        bb = dict(bb)
        bb["conv1.weight"] = bb["conv1.weight"][:, [2, 1, 0]].contiguous()
        self._engine.load_weights(L.PT_MODEL_MTL_BACKBONE, pack_mtl_backbone(bb, fmt=self._engine.weight_fmt))
        self._engine.load_weights(L.PT_MODEL_MTL_DECODER, pack_mtl_decoder(dec, self._convertor.decoder_cfg(), fmt=self._engine.weight_fmt))
        self._model = self._predict

    def _build_processor(self):
        if self.model in ("MtlTabNet", "TableMaster"):
            # the reference-shaped door keeps the reference's IndexError for a table without a surviving box; the batched door does not
            self._stage = MtlStage(self._engine, self._convertor, size=self._config.size, micro_batch=int(os.environ.get("PT_MTL_MICROBATCH", "32")))
            return
        # tables per DLA-34 launch chain: 8-table launches leave most of the 256 CUs idle in the coarse levels (measured
        # with the bench's 80); TsrStage balances the last micro-batch (87 tables -> 44 + 43)
        self._stage = TsrStage(self._engine, self._config, micro_batch=int(os.environ.get("PT_TSR_MICROBATCH", "128")))

    def _predict(self, images: List[np.ndarray]) -> List[Dict]:
        """one table image each (RGB ndarray): the whole image is the crop"""
        out = []
        if self.model in ("MtlTabNet", "TableMaster"):
            self._stage.post.strict = True
            try:
                for img in images:
                    page = torch.from_numpy(np.ascontiguousarray(img)[None]).to(self._engine._tdev)
                    out.append(self._stage(page, [np.array([[0, 0, img.shape[1], img.shape[0]]])])[0][0])
            finally:
                self._stage.post.strict = False
            return out
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
        if self.model in ("MtlTabNet", "TableMaster"):
            # mmcv's imread hands an ndarray on AS IS and reads a file as BGR (table/lgpma/base_utils.py:689-739); the engine holds the
            # network with conv1's channels swapped (it eats RGB), so an ndarray is flipped here and a file is read as RGB
            return {"inputs": [{"image": np.ascontiguousarray(it[..., ::-1]) if isinstance(it, np.ndarray) else _read_image(it), "inputs": it}
                               for it in inputs]}
        return {"inputs": [{"image": _read_image(it), "inputs": it} for it in inputs]}

    def _run_model(self, inputs, **kwargs):
        begin = time.time()
        res, elapse = self.infer({"images": [it["image"] for it in inputs["inputs"]]})
        inputs["results"] = [{"results": r, "elapse": elapse, "inputs": it["inputs"]} for r, it in zip(res, inputs["inputs"])]
        inputs["use_time"] = time.time() - begin
        return inputs

    def _postprocess(self, inputs, **kwargs) -> List[Dict]:
        out = []
        if self.model in ("MtlTabNet", "TableMaster"):
            for r in inputs["results"]:
                d = {k: r["results"][k] for k in ("polygons", "structure_str_list", "structure_str", "html_context")}
                d["inputs"] = r["inputs"]
                out.append(d)
            return out
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
