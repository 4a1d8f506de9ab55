#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REFERENCE's own Python modules.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported as a whole (``import pdftable`` needs pdfminer, cv2,
onnxruntime ... -- SURVEY.md section 8c), so single files are imported by path underneath empty
stand-in parent packages (only ``__path__`` is set; no reference ``__init__`` runs).  Only DATA is
written here: seeded inputs and the tensors the reference modules produce for them.  Weights are
NOT stored; they are regenerated from a seed by ``pdf_table_amd.synth_weights`` and loaded into
the reference modules with ``strict=True`` (which also pins every key name and shape).
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
sys.path.insert(0, REPO)


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def ref_import(modname):
    """import pdftable.<...> by file without running any package __init__."""
    parts = modname.split(".")
    for i in range(1, len(parts)):
        pk = ".".join(parts[:i])
        if pk not in sys.modules:
            _pkg(pk, os.path.join(REF_SRC, *parts[:i]))
    return importlib.import_module(modname)


def gen_db_resnet18():
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    dbnet = ref_import("pdftable.model.db_net.dbnet")
    torch.manual_seed(0)
    model = dbnet.DBModel().eval()
    sd = db_resnet18_state_dict(seed=11)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(101)
    out = {}
    for tag, (h, w) in {"a": (64, 96), "b": (128, 128)}.items():
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            xt = torch.from_numpy(x)
            c2, c3, c4, c5 = model.backbone(xt)
            y = model(xt)
        out[f"x_{tag}"] = x
        out[f"prob_{tag}"] = y.numpy()
        out[f"c2_{tag}"] = c2.numpy()
        out[f"c5_{tag}"] = c5.numpy()
    out["seed"] = np.array(11)
    np.savez_compressed(os.path.join(HERE, "db_resnet18.npz"), **out)
    print("db_resnet18.npz", {k: v.shape for k, v in out.items()})


def gen_db_nas():
    from pdf_table_amd.synth_weights import db_nas_state_dict
    dbnet = ref_import("pdftable.model.db_net.dbnet")
    torch.manual_seed(0)
    model = dbnet.DBNasModel().eval()
    sd = db_nas_state_dict(seed=13)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(113)
    out = {}
    for tag, (h, w) in {"a": (64, 96), "b": (160, 128)}.items():
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            xt = torch.from_numpy(x)
            c2, c3, c4, c5 = model.backbone(xt)
            y = model(xt)
        out[f"x_{tag}"] = x
        out[f"prob_{tag}"] = y.numpy()
        out[f"c2_{tag}"] = c2.numpy()
        out[f"c5_{tag}"] = c5.numpy()
    out["seed"] = np.array(13)
    # the module's own parameter inventory: the synthetic checkpoint must have exactly these keys and shapes
    ref_sd = dbnet.DBNasModel().state_dict()
    out["keys"] = np.array(list(ref_sd.keys()))
    out["shapes"] = np.array([",".join(str(d) for d in v.shape) for v in ref_sd.values()])
    np.savez_compressed(os.path.join(HERE, "db_nas.npz"), **out)
    print("db_nas.npz", {k: v.shape for k, v in out.items()})


def gen_pplcnet():
    """reference PPLCNet (model/cls/cls_pp_lcnet.py) logits, the reference's Topk / TableAttribute post-processors and its
    PPLCNetImageProcessor (-> Pillow bilinear resize) on the seeded inputs of tests/cls_synth.py"""
    import importlib.util
    sys.path.insert(0, os.path.dirname(HERE))
    from cls_synth import CLS_GOLDEN_TASKS, PIL_CASES, PRE_CASES, cls_inputs, u8_image
    from pdf_table_amd.synth_weights import pplcnet_state_dict
    stub_env()      # image_processing_pplcnet.py imports cv2 (only to read image paths)
    spec = importlib.util.spec_from_file_location("cls_pp_lcnet", os.path.join(REF_SRC, "pdftable/model/cls/cls_pp_lcnet.py"))
    net_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(net_mod)
    spec = importlib.util.spec_from_file_location("image_processing_pplcnet",
                                                  os.path.join(REF_SRC, "pdftable/model/cls/image_processing_pplcnet.py"))
    ip = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ip)
    out = {}
    post = {}
    for task, (cn, textline, hw, seed) in CLS_GOLDEN_TASKS.items():
        strides = [2, [2, 1], [2, 1], [2, 1], [2, 1]] if textline else [2, 2, 2, 2, 2]
        net = net_mod.PPLCNet(class_num=cn, stride_list=strides).eval()
        net.load_state_dict(pplcnet_state_dict(seed, cn), strict=True)
        with torch.no_grad():
            y = net(torch.from_numpy(cls_inputs(seed, 5, hw)))
        out[f"logits_{task}"] = y.numpy()
        post[task] = ip.PPLCNetImagePostProcessor(task=task)({"results": y})
    # pre-processor: uint8 RGB images of odd sizes through the reference's own processor (Pillow resize inside)
    for i, (task, (h, w)) in enumerate(PRE_CASES):
        pv = ip.PPLCNetImageProcessor(task=task)(u8_image(i, h, w))["pixel_values"][0]
        out[f"pre_out_{i}"] = np.asarray(pv, dtype=np.float32)
    # Pillow itself on more shapes (the resize is third-party code: pinned to the library in this image)
    import PIL
    from PIL import Image
    for i, (h, w, oh, ow) in enumerate(PIL_CASES):
        out[f"pil_out_{i}"] = np.array(Image.fromarray(u8_image(100 + i, h, w)).resize((ow, oh), resample=Image.BILINEAR))
    out["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(HERE, "pplcnet.npz"), **out)
    with open(os.path.join(HERE, "pplcnet_post.json"), "w") as f:
        json.dump(post, f, indent=1)
    print("pplcnet.npz", {k: v.shape for k, v in out.items()})


def _matcher_env():
    """import environment for the reference's OcrTableToHtmlTask (third-party libraries stubbed, reference files real)"""
    from collections import OrderedDict
    import transformers  # noqa: F401
    stub_env()
    for m in ["pypdf", "PyPDF2", "pdf2image", "docx", "ghostscript", "requests", "tqdm", "xlsxwriter", "openpyxl", "bs4", "lxml",
              "Levenshtein", "apted", "apted.helpers", "distance", "pdfminer.high_level", "pdfminer.pdfpage",
              "pdfminer.pdfinterp", "pdfminer.converter", "pdfminer.pdfdocument", "pdfminer.pdfparser", "pdfminer.utils",
              "pdfminer.image", "pdfminer.pdftypes", "pdfminer.psparser", "pdfminer.pdfdevice", "pdfminer.pdffont",
              "pdfminer.pdfcolor", "skimage", "skimage.draw", "PIL.ImageDraw", "matplotlib", "matplotlib.pyplot",
              "matplotlib.patches"]:
        sys.modules.setdefault(m, _Stub(m))
    for sub in ("model", "model/pdf_table", "entity", "model/ocr_pdf", "model/ocr_pdf/table", "utils/table"):
        name = "pdftable." + sub.replace("/", ".")
        if name not in sys.modules:
            _pkg(name, os.path.join(REF_SRC, "pdftable", sub))
    pl = types.ModuleType("pdfminer.layout")
    for nme in ("LTChar", "LTTextLineHorizontal", "LTAnno", "LTImage", "LTTextLineVertical", "LTTextLine", "LTTextBoxHorizontal",
                "LTFigure", "LTRect", "LTLine", "LTCurve", "LTTextBox", "LTPage", "LAParams", "LTContainer", "LTTextContainer"):
        setattr(pl, nme, type(nme, (), {}))
    sys.modules["pdfminer.layout"] = pl
    ent = sys.modules["pdftable.entity"]
    ee = ref_import("pdftable.entity.enum_entity")
    for k in dir(ee):
        if not k.startswith("_"):
            setattr(ent, k, getattr(ee, k))
    pu = sys.modules["pdftable.utils"]
    pu.MathUtils = _Stub("MathUtils")
    pu.Constants = ref_import("pdftable.utils.constant").Constants
    pu.MatchUtils = ref_import("pdftable.utils.match_utils").MatchUtils
    pu.CommonUtils = ref_import("pdftable.utils.common_utils").CommonUtils
    ut = sys.modules["pdftable.utils.table"]
    for nme in ("GhostscriptBackend", "PopplerBackend", "ImageConversionBackend"):
        setattr(ut, nme, _Stub(nme))
    te = ref_import("pdftable.entity.table_entity")
    ref_import("pdftable.entity.ie_entity")
    pu.PdfUtils = ref_import("pdftable.utils.pdf_utils").PdfUtils
    tc = ref_import("pdftable.model.pdf_table.table_common")
    core = ref_import("pdftable.model.pdf_table.table_core")
    pm = sys.modules["pdftable.model"]
    pm.Cell, pm.box_in_other_box, pm.distance, pm.compute_iou_v2 = core.Cell, tc.box_in_other_box, tc.distance, tc.compute_iou_v2
    pm.TableProcessUtils = tc.TableProcessUtils
    oo = types.ModuleType("pdftable.model.ocr_pdf.ocr_output")
    oo.OcrSystemModelOutput = type("OcrSystemModelOutput", (), {})
    sys.modules["pdftable.model.ocr_pdf.ocr_output"] = oo
    sys.modules["pdftable.model.ocr_pdf.table"].TableMatch = lambda **k: None        # only constructed, never used on this path
    tm = types.ModuleType("pdftable.model.ocr_pdf.table.table_master_match")
    tm.TableMasterMatcher = lambda **k: None
    sys.modules["pdftable.model.ocr_pdf.table.table_master_match"] = tm
    t2h = ref_import("pdftable.model.ocr_pdf.ocr_table_to_html_task")
    return t2h.OcrTableToHtmlTask, tc.TableProcessUtils, te.OcrCell


def gen_table_text_match():
    """the reference's own OcrTableToHtmlTask.match_table_cell_and_text_cell (ocr_table_to_html_task.py:178-243, with
    find_top1_mach_box :48-77 and get_one_cell_text :297-330), get_text_in_table_bbox (table_common.py:1303-1325) and
    cell_to_html with text and widths on the seeded tables / OCR lines of tests/lore_synth.py"""
    sys.path.insert(0, os.path.dirname(HERE))
    from lore_synth import synth_table_grids, synth_table_texts
    Task, T, OcrCell = _matcher_env()
    task = Task(output_dir="/tmp/pt_golden_html")
    out = {"seed": 17, "cases": []}
    for ci, (polys, logi) in enumerate(synth_table_grids(17)):
        boxes, texts = synth_table_texts(ci, polys)
        for post in (False, True):
            cells = T.get_table_cell_from_table_logit(table_bboxs=polys, logits=logi, save_html_file=None)
            ocr = [OcrCell(raw_data={"index": i + 1, "text": t, "bbox": b}) for i, (b, t) in enumerate(zip(boxes, texts))]
            bbox = [float(polys[:, 0::2].min()), float(polys[:, 1::2].min()), float(polys[:, 0::2].max()), float(polys[:, 1::2].max())]
            inside, remain = T.get_text_in_table_bbox(bbox=bbox, ocr_results=ocr, diff=2)
            top1 = [int(task.find_top1_mach_box(text_box=c.to_bbox(), table_bboxs=cells)) for c in inside]
            res, html, metric, db_html = task.match_table_cell_and_text_cell(table_idx=0, table_cells=cells, text_bboxs=inside,
                                                                             raw_filename="g", ocr_post_process=post)
            out["cases"].append({"case": ci, "ocr_post_process": post, "bbox": bbox, "inside": [int(c.index) for c in inside],
                                 "top1": top1, "html": html, "db_html": db_html,
                                 "cells": [[float(c.row_index), float(c.col_index), c.text] for c in res]})
    with open(os.path.join(HERE, "table_text_match.json"), "w") as f:
        json.dump(out, f)
    print("table_text_match.json", len(out["cases"]), "cases;", out["cases"][0]["html"][:6])


def gen_crnn():
    from pdf_table_amd.synth_weights import crnn_state_dict
    crnn = ref_import("pdftable.model.crnn.modeling_crnn")
    model = crnn.CRNN().eval()
    sd = crnn_state_dict(seed=12)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(102)
    x = rng.uniform(0, 1, (2, 3, 32, 96)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x))
    # logits are [2, 24, 7644]; keep every 16th class column + argmax to stay small
    ynp = y.numpy()
    np.savez_compressed(os.path.join(HERE, "crnn.npz"), x=x, seed=np.array(12),
                        logits_sub=ynp[:, :, ::16], argmax=ynp.argmax(-1).astype(np.int32),
                        maxval=ynp.max(-1))
    print("crnn.npz logits", ynp.shape)


def gen_registry_hash():
    cfg = ref_import("pdftable.model.ocr_pdf.ocr_table_model_config")
    blob = json.dumps(cfg.TABLE_MODEL_DICT, sort_keys=True, ensure_ascii=False).encode("utf-8")
    leaves = []

    def walk(d, path):
        if isinstance(d, dict):
            for k, v in d.items():
                walk(v, path + (str(k),))
        else:
            leaves.append(("/".join(path), d))
    walk(cfg.TABLE_MODEL_DICT, ())
    info = {"sha256": hashlib.sha256(blob).hexdigest(), "n_leaves": len(leaves),
            "providers": sorted(cfg.TABLE_MODEL_DICT.keys())}
    with open(os.path.join(HERE, "registry_hash.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print("registry", info)


def gen_ctc():
    """CTCLabelDecode known answers (ocr_rec_pp/rec_postprocess.py:126-191)."""
    rp = ref_import("pdftable.model.ocr_rec_pp.rec_postprocess")
    chars = list("abcdefghij")
    dict_path = os.path.join(HERE, "_tmp_dict.txt")
    with open(dict_path, "w") as f:
        f.write("\n".join(chars) + "\n")
    dec = rp.CTCLabelDecode(character_dict_path=dict_path, use_space_char=True)
    rng = np.random.default_rng(103)
    ncls = len(dec.character)
    probs = rng.uniform(0, 1, (6, 40, ncls)).astype(np.float32)
    # make repeats and blanks frequent
    for b in range(6):
        for t in range(40):
            if rng.uniform() < 0.35:
                probs[b, t, 0] = 2.0
            elif t > 0 and rng.uniform() < 0.4:
                probs[b, t] = probs[b, t - 1]
    res = dec(probs)
    os.remove(dict_path)
    with open(os.path.join(HERE, "ctc_decode.json"), "w") as f:
        json.dump({"chars": chars, "character": dec.character,
                   "texts": [r[0] for r in res], "confs": [float(r[1]) for r in res]}, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "ctc_decode.npz"), probs=probs)
    print("ctc", [r[0] for r in res])


def gen_rec_pp():
    """The reference's PPOcrRecPreProcessor (model/ocr_rec_pp/processor_ocr_rec_pp.py) on seeded crops, one call with all
    crops (mini-batches of 6, width-sorted) and one call per crop (how the system loop uses it).  cv2 is not installed: the
    module's ``cv2.resize`` is this repository's restatement of OpenCV's 8-bit bilinear resize (oracle/db_pre.py) -- every
    other line that runs is the reference's."""
    from oracle.db_pre import cv2_resize_linear_u8
    sys.path.insert(0, os.path.dirname(HERE))
    from rec_synth import rec_pp_crops
    stub_env()
    cv2 = types.ModuleType("cv2")
    cv2.resize = lambda img, size: cv2_resize_linear_u8(np.ascontiguousarray(img), int(size[0]), int(size[1]))
    cv2.COLOR_GRAY2RGB = 8
    sys.modules["cv2"] = cv2
    # the module's config class cannot be defined under this image's transformers (PretrainedConfig is a dataclass there,
    # and `attribute_map: Dict = {...}` is a mutable default): the pre-processor only reads four attributes of its config
    cm = types.ModuleType("pdftable.model.ocr_rec_pp.configuration_ocr_recognition_pp")
    cm.PPOcrRecognitionConfig = type("PPOcrRecognitionConfig", (), {})
    for pk in ("pdftable", "pdftable.model", "pdftable.model.ocr_rec_pp"):
        if pk not in sys.modules:
            _pkg(pk, os.path.join(REF_SRC, *pk.split(".")))
    sys.modules[cm.__name__] = cm
    mod = ref_import("pdftable.model.ocr_rec_pp.processor_ocr_rec_pp")
    mod.cv2 = cv2
    cfg = types.SimpleNamespace(rec_image_shape=[3, 48, 320], rec_batch_num=6, limited_max_width=1280, limited_min_width=16)
    pre = mod.PPOcrRecPreProcessor(cfg)
    crops = rec_pp_crops()
    out = {"seed": np.array(107)}
    batches = pre(list(crops))
    out["indices"] = np.asarray(batches[0]["indices"])
    for b, d in enumerate(batches):
        out[f"batch{b}"] = d["image"]
        out[f"beg{b}"] = np.array(d["batch_beg_img_no"])
    out["n_batches"] = np.array(len(batches))
    for i in (0, 4, 5, 6):          # one crop per call: its own width, never padded beyond it
        out[f"single{i}"] = pre(crops[i])[0]["image"]
    np.savez_compressed(os.path.join(HERE, "rec_pp.npz"), **out)
    print("rec_pp.npz", {k: v.shape for k, v in out.items()})


class _Stub(types.ModuleType):
    """Stand-in for third-party modules that are not installed (cv2, pyclipper, shapely ...).  Only used
    so that reference files IMPORT; any golden value below comes from reference code paths that do not
    touch these stand-ins (except cv2.resize, which records the requested size and returns zeros)."""
    calls = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Stub(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        if self.__name__ == "cv2.resize":
            w, h = a[1]
            _Stub.calls.append((int(w), int(h)))
            return np.zeros((int(h), int(w), 3), dtype=np.uint8)
        return None


def stub_env():
    import logging
    for m in ["cv2", "pyclipper", "shapely", "shapely.geometry", "fitz", "pdfminer", "dotenv"]:
        sys.modules.setdefault(m, _Stub(m))
    try:
        import transformers.onnx  # noqa: F401  (removed in transformers 5; the reference subclasses OnnxConfig)
    except Exception:
        tm = types.ModuleType("transformers.onnx")
        tm.OnnxConfig = type("OnnxConfig", (), {})
        sys.modules["transformers.onnx"] = tm
    if "pdftable.utils" not in sys.modules:
        u = _pkg("pdftable", os.path.join(REF_SRC, "pdftable")) if "pdftable" not in sys.modules else None
        pu = _pkg("pdftable.utils", os.path.join(REF_SRC, "pdftable", "utils"))
        pu.logger = logging.getLogger("ref")
        pu.BaseUtil = type("BaseUtil", (), {})
        pu.FileUtils = _Stub("FileUtils")
        pu.Constants = _Stub("Constants")
        pu.TimeUtils = _Stub("TimeUtils")
        pu.CommonUtils = _Stub("CommonUtils")
        pu.MatchUtils = _Stub("MatchUtils")
        # utils/ocr/__init__.py drags in pdfminer through pdftable.entity: load the one file we need instead
        po = _pkg("pdftable.utils.ocr", os.path.join(REF_SRC, "pdftable", "utils", "ocr"))
        po.OcrCommonUtils = importlib.import_module("pdftable.utils.ocr.ocr_common_utils").OcrCommonUtils
        po.OcrInferUtils = _Stub("OcrInferUtils")


def gen_db_host_numpy():
    """Pure-numpy pieces of the detection path, produced by the reference's own code:
    DetResizeForTest size arithmetic, filter_tag_det_res/order_points_clockwise, order_point, the
    text_detection sort key."""
    stub_env()
    ops = ref_import("pdftable.model.db_pp.image_operators")
    rz = ops.DetResizeForTest(limit_side_len=960, limit_type="max")
    sizes = [(1024, 1024), (640, 640), (1000, 700), (333, 517), (2000, 1500), (31, 47), (960, 1280), (1920, 1920),
             (975, 975), (976, 944)]
    plan = []
    for (h, w) in sizes:
        _Stub.calls.clear()
        _, (rh, rw) = rz.resize_image_type0(np.zeros((h, w, 3), np.uint8))
        nw, nh = _Stub.calls[-1]
        plan.append([h, w, nh, nw, rh, rw])
    # (the torch-flavour OCRDetectionPreprocessor cannot be imported: its PretrainedConfig subclass breaks
    #  under transformers 5 -- its size rule is pinned by hand-computed answers in the tests instead)
    # DbPPConfig subclasses transformers.PretrainedConfig with a mutable class attribute, which transformers 5
    # rejects at class creation; the post-processor methods used below never touch the config.
    cm = types.ModuleType("pdftable.model.db_pp.configuration_db_pp")
    cm.DbPPConfig = type("DbPPConfig", (), {})
    cm.__all__ = ["DbPPConfig"]
    sys.modules["pdftable.model.db_pp.configuration_db_pp"] = cm
    post = ref_import("pdftable.model.db_pp.processor_ocr_db_pp")
    pp = object.__new__(post.PPOcrDetectionPostProcessor)
    rng = np.random.default_rng(104)
    boxes = []
    for bi in range(40):
        c = rng.uniform(-20, 700, 2)
        wh = rng.uniform(1, 8, 2) if bi < 12 else rng.uniform(1, 120, 2)
        ang = rng.uniform(0, np.pi)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        q = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]]) * wh / 2 @ R.T + c
        boxes.append(np.round(q[rng.permutation(4)]).astype(np.int16))
    boxes = np.array(boxes, dtype=np.int16)
    filt = pp.filter_tag_det_res(boxes.copy(), (640, 672, 3))
    ocu = ref_import("pdftable.utils.ocr.ocr_common_utils")
    op_in = rng.uniform(0, 500, (30, 8)).astype(np.float32)
    op_out = np.stack([ocu.OcrCommonUtils.order_point(r) for r in op_in])
    det = rng.integers(0, 900, (50, 8)).astype(np.float32)
    lst = sorted(det.tolist(), key=lambda x: 0.01 * sum(x[::2]) / 4 + sum(x[1::2]) / 4)  # ocr_system_task.py:161
    np.savez_compressed(os.path.join(HERE, "db_host_numpy.npz"), plan=np.array(plan, dtype=np.float64),
                        boxes=boxes, filtered=filt.astype(np.float32),
                        filter_shape=np.array([640, 672, 3]), order_point_in=op_in, order_point_out=op_out,
                        sort_in=det, sort_out=np.array(lst, dtype=np.float32))
    print("db_host_numpy.npz", len(filt), "of", len(boxes), "boxes kept")


def lore_env():
    """Make lore/*.py importable: parent packages as empty stand-ins (their __init__ pull in transformers configs,
    cv2 ...) and ``torchvision.ops.deform_conv2d`` -- torchvision is not installed -- provided by the oracle's
    restatement, which is itself pinned against the reference's vendored DCNv2 C++ (tests/test_oracle_lore.py)."""
    import transformers  # noqa: F401  (must be imported before the fake torchvision appears)
    from oracle import lore_net
    stub_env()
    for sub in ("model", "model/center_net", "model/lore"):
        name = "pdftable." + sub.replace("/", ".")
        if name not in sys.modules:
            _pkg(name, os.path.join(REF_SRC, "pdftable", sub))
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvo = types.ModuleType("torchvision.ops")

        def deform_conv2d(x, offset, weight, bias, stride, padding, dilation, mask):
            return lore_net.deform_conv2d(x, offset, mask, weight, bias, stride[0], padding[0], dilation[0])
        tvo.deform_conv2d = deform_conv2d
        tv.ops = tvo
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = tvo


def gen_lore_dla34():
    """Outputs of the reference ``DLASeg`` (lore_dla_34.py:137-196) for seeded weights and inputs."""
    from pdf_table_amd.synth_weights import LORE_HEADS, lore_dla34_state_dict
    lore_env()
    m = ref_import("pdftable.model.lore.lore_dla_34")
    model = m.get_dla_dcn(34, dict(LORE_HEADS), head_conv=256).eval()
    sd = lore_dla34_state_dict(seed=21)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(105)
    out = {"seed": np.array(21)}
    for tag, (h, w) in {"a": (128, 160), "b": (64, 64)}.items():
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            z = model(torch.from_numpy(x))[0]
        out[f"x_{tag}"] = x
        for k, v in z.items():
            v = v.numpy()
            out[f"{k}_{tag}"] = v[:, ::8] if v.shape[1] == 256 else v     # every 8th channel of ax / cr
    np.savez_compressed(os.path.join(HERE, "lore_dla34.npz"), **out)
    print("lore_dla34.npz", {k: v.shape for k, v in out.items()})


def gen_lore_wireless():
    """Outputs of the reference ``LoreDetectModel`` (lore_detector.py:155-389) for seeded weights and inputs."""
    from pdf_table_amd.synth_weights import lore_wireless_state_dict
    lore_env()
    m = ref_import("pdftable.model.lore.lore_detector")
    model = m.LoreDetectModel().eval()
    sd = lore_wireless_state_dict(seed=23)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(108)
    out = {"seed": np.array(23)}
    for tag, (h, w) in {"a": (128, 192), "b": (64, 128)}.items():   # sizes must be multiples of the total stride 64
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            z = model(torch.from_numpy(x))[0]
        out[f"x_{tag}"] = x
        for k, v in z.items():
            v = v.numpy()
            out[f"{k}_{tag}"] = v[:, ::8] if v.shape[1] == 256 else v
    np.savez_compressed(os.path.join(HERE, "lore_wireless.npz"), **out)
    print("lore_wireless.npz", sum(p_.numel() for p_ in model.parameters()), "parameters")


def gen_lore_decode():
    """Outputs of the reference's own process_detect_output / process_logic_output (lineless_table_process.py:592-663)
    on seeded head maps.  cv2.getAffineTransform and shapely are not installed: the oracle's stand-ins are injected
    (those two stay PARITY UNPINNED); every other line that runs is the reference's."""
    lore_env()
    from oracle import lore_decode as od
    sys.path.insert(0, os.path.dirname(HERE))
    from lore_synth import synth_lore_heads          # tests/lore_synth.py: inputs are regenerated from the seed
    cv2 = sys.modules["cv2"]
    cv2.getAffineTransform = lambda s, d: od.get_affine_transform_3pt(np.asarray(s), np.asarray(d))

    class Polygon:
        def __init__(self, pts):
            self.p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)

    class Point:
        def __init__(self, xy):
            self.x, self.y = float(xy[0]), float(xy[1])

        def within(self, poly):
            return od.point_strictly_in_polygon(self.x, self.y, poly.p)
    sg = types.ModuleType("shapely.geometry")
    sg.Polygon, sg.Point, sg.MultiPoint = Polygon, Point, object
    sh = types.ModuleType("shapely")
    sh.geometry = sg
    sys.modules["shapely"] = sh
    sys.modules["shapely.geometry"] = sg
    torch.Tensor.cuda = lambda self, *a, **k: self            # the decode calls .cuda() on helper tensors
    m = ref_import("pdftable.model.lore.lineless_table_process")
    out = {}
    # (K = 3000 cells / 5000 corners are hard-coded in the reference: maps need >= 5000 pixels)
    for tag, (seed, H, W, src_h, src_w, rev) in {"a": (1, 80, 80, 300, 420, True), "b": (2, 72, 96, 777, 512, True),
                                                 "c": (3, 80, 80, 256, 256, False), "d": (4, 96, 96, 500, 333, False)}.items():
        heads = synth_lore_heads(seed, H, W)
        ul = tag == "d"                             # the 'wireless' configuration: upper_left=True, wiz_rev=False
        _, meta = od.lore_preprocess_geometry(src_h, src_w, 4 * H, 4 * W, upper_left=ul)
        hd = {k: torch.from_numpy(v.copy()) for k, v in heads.items()}
        logi, ps, results, corner = m.process_detect_output(hd, torch.from_numpy(meta)[None], upper_left=ul,
                                                            wiz_rev=rev, vis_thresh=0.2)
        n = logi.shape[1]
        out[f"case_{tag}"] = np.array([seed, H, W, src_h, src_w, int(rev), int(ul)])
        out[f"meta_{tag}"] = meta
        out[f"logi_{tag}"] = logi.numpy()
        out[f"ps_{tag}"] = ps.numpy()
        out[f"results_{tag}"] = results[1]
        print("lore decode", tag, "cells", n)
    lg = torch.from_numpy(np.random.default_rng(9).uniform(-1, 12, (1, 50, 4)).astype(np.float32))
    lg[0, :4, 0] = torch.tensor([2.5, 3.5, 0.5, 7.500001])
    out["logic_in"] = lg.numpy()
    out["logic_out"] = m.process_logic_output(lg.clone()).numpy()
    np.savez_compressed(os.path.join(HERE, "lore_decode.npz"), **out)


def gen_lore_processor():
    """Outputs of the reference LoreProcessModel (lore/lore_processor.py:399-514) for seeded weights and features."""
    from pdf_table_amd.synth_weights import lore_processor_state_dict
    lore_env()
    # lore_processor.py imports these two by absolute name; only _tranpose_and_gather_feat & co. (training path) come from them
    tp = types.ModuleType("pdftable.model.center_net.table_process")
    tp._tranpose_and_gather_feat = None
    sys.modules["pdftable.model.center_net.table_process"] = tp
    cl = types.ModuleType("pdftable.model.lore.configuration_lore")
    cl.LoreConfig = type("LoreConfig", (), {})
    sys.modules["pdftable.model.lore.configuration_lore"] = cl
    torch.Tensor.cuda = lambda self, *a, **k: self
    m = ref_import("pdftable.model.lore.lore_processor")
    out = {}
    rng = np.random.default_rng(106)
    for tag, (L, n, with_dets) in {"wtw": (4, 137, False), "ptn": (3, 60, True)}.items():
        cfg = types.SimpleNamespace(stacking_layers=L, tsfm_layers=L, wiz_stacking=True, wiz_2dpe=with_dets)
        model = m.LoreProcessModel(cfg).eval()
        sd = lore_processor_state_dict(seed=31, layers=L, stacking_layers=L)
        model.load_state_dict(sd, strict=True)
        feat = rng.standard_normal((1, n, 256)).astype(np.float32)
        dets = rng.integers(0, 256, (1, n, 8)).astype(np.int64) if with_dets else None
        with torch.no_grad():
            logic, stacked = model(torch.from_numpy(feat), dets=None if dets is None else torch.from_numpy(dets))
        out[f"feat_{tag}"] = feat
        if dets is not None:
            out[f"dets_{tag}"] = dets
        out[f"logic_{tag}"] = logic.numpy()
        out[f"stacked_{tag}"] = stacked.numpy()
        out[f"layers_{tag}"] = np.array(L)
        print("lore processor", tag, logic.shape, float(logic.abs().max()), float(stacked.abs().max()))
    np.savez_compressed(os.path.join(HERE, "lore_processor.npz"), **out)


def gen_picodet():
    """(1) LCNet -> CSPPAN -> PicoHead.forward_eval(export_post_process=False) of the reference on seeded weights;
    (2) the reference OCRPicodetPostProcessor on seeded head outputs."""
    from pdf_table_amd.synth_weights import picodet_state_dict
    import transformers  # noqa: F401
    sys.path.insert(0, os.path.dirname(HERE))
    from lore_synth import synth_pico_heads      # tests/lore_synth.py: inputs are regenerated from the seed
    stub_env()
    if "torchvision" not in sys.modules:      # pico_utils.py imports two names from torchvision.ops at module level
        tv = types.ModuleType("torchvision")
        tvo = types.ModuleType("torchvision.ops")
        tvo.DeformConv2d = type("DeformConv2d", (torch.nn.Module,), {})
        tvo.nms = None
        tv.ops = tvo
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = tvo
    else:
        sys.modules["torchvision.ops"].DeformConv2d = type("DeformConv2d", (torch.nn.Module,), {})
        sys.modules["torchvision.ops"].nms = None
    for sub in ("model", "model/picodet"):
        name = "pdftable." + sub.replace("/", ".")
        if name not in sys.modules:
            _pkg(name, os.path.join(REF_SRC, "pdftable", sub))
    lc = ref_import("pdftable.model.picodet.lcnet")
    cp = ref_import("pdftable.model.picodet.csp_pan")
    ph = ref_import("pdftable.model.picodet.pico_head")
    bb = lc.LCNet(scale=1.0, feature_maps=[3, 4, 5]).eval()
    neck = cp.CSPPAN(in_channels=[128, 256, 512], out_channels=128, kernel_size=5, num_features=4, num_csp_blocks=1,
                     use_depthwise=True, act="hard_swish", spatial_scales=[0.125, 0.0625, 0.03125]).eval()
    head = ph.PicoHead(conv_feat=dict(feat_in=128, feat_out=128, num_fpn_stride=4, num_convs=4, norm_type="bn",
                                      share_cls_reg=True, act="hard_swish", use_se=False),
                       num_classes=5, fpn_stride=[8, 16, 32, 64], loss_class=dict(use_sigmoid=True, iou_weighted=True, loss_weight=1.0),
                       nms=dict(nms_top_k=1000, keep_top_k=100, score_threshold=0.025, nms_threshold=0.6), reg_max=7,
                       feat_in_chan=128, cell_offset=0.5).eval()
    sd = picodet_state_dict(seed=41, num_classes=5)
    for name, mod in (("backbone", bb), ("neck", neck), ("head", head)):
        mod.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")}, strict=True)
    rng = np.random.default_rng(107)
    out = {"seed": np.array(41)}
    for tag, (h, w) in {"a": (160, 128), "b": (224, 192)}.items():
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            f = bb(image=torch.from_numpy(x))
            n = neck(f)
            sc, bx = head(n, export_post_process=False)
        out[f"x_{tag}"] = x
        out[f"c5_{tag}"] = f[-1].numpy()
        out[f"p3_{tag}"] = n[0].numpy()[:, ::4]
        for i in range(4):
            out[f"score{i}_{tag}"] = sc[i].numpy()
            out[f"box{i}_{tag}"] = bx[i].numpy()
    # post-processor: its config class subclasses transformers.PretrainedConfig with list defaults (rejected by
    # transformers 5), so the instance is built without it and given the same attributes (processor_picodet.py:126-131)
    cm = types.ModuleType("pdftable.model.picodet.configuration_picodet")
    cm.PicodetConfig = type("PicodetConfig", (), {})
    sys.modules["pdftable.model.picodet.configuration_picodet"] = cm
    pp = ref_import("pdftable.model.picodet.processor_picodet")
    labels = ["text", "title", "list", "table", "figure"]
    post = object.__new__(pp.OCRPicodetPostProcessor)
    post.config = types.SimpleNamespace(id2label=dict(enumerate(labels)))
    post.strides, post.score_threshold, post.nms_threshold, post.nms_top_k, post.keep_top_k = [8, 16, 32, 64], 0.5, 0.5, 1000, 100
    for tag, (seed, tgt, org) in {"p": (1, (160, 128), (1024, 1024)), "q": (2, (800, 608), (1100, 850))}.items():
        sc, bx = synth_pico_heads(seed, tgt)
        sf = [float(tgt[0]) / org[0], float(tgt[1]) / org[1]]
        res = post({"boxes": sc, "boxes_num": bx, "org_shape": list(org), "scale_factor": sf, "target_shape": list(tgt)})
        out[f"post_case_{tag}"] = np.array([seed, tgt[0], tgt[1], org[0], org[1]])
        out[f"post_bbox_{tag}"] = np.array([r["bbox"] for r in res["bboxs"]], dtype=np.float64).reshape(-1, 4)
        out[f"post_score_{tag}"] = np.array([r["score"] for r in res["bboxs"]], dtype=np.float64)
        out[f"post_cls_{tag}"] = np.array([r["category_id"] for r in res["bboxs"]], dtype=np.int64)
        print("picodet post", tag, len(res["bboxs"]), "boxes")
    np.savez_compressed(os.path.join(HERE, "picodet.npz"), **out)
    print("picodet.npz", os.path.getsize(os.path.join(HERE, "picodet.npz")))


def gen_table_html():
    """cells and structure HTML from the reference's own get_table_cell_from_table_logit / cell_to_html
    (pdf_table/table_common.py:1643-1663, 578-669; entity/table_entity.py:569-657).  ``CommonUtils.sorted_dict`` (an
    OrderedDict of ``sorted(items)``, utils/common_utils.py:53-65) is stood in because utils/__init__ cannot be imported."""
    from collections import OrderedDict
    import transformers  # noqa: F401
    stub_env()
    sys.path.insert(0, os.path.dirname(HERE))
    from lore_synth import synth_table_grids
    for sub in ("model", "model/pdf_table", "entity"):
        name = "pdftable." + sub.replace("/", ".")
        if name not in sys.modules:
            _pkg(name, os.path.join(REF_SRC, "pdftable", sub))
    sys.modules["pdftable.entity"].LineDirectionType = ref_import("pdftable.entity.enum_entity").LineDirectionType
    pl = types.ModuleType("pdfminer.layout")
    for nme in ("LTChar", "LTTextLineHorizontal", "LTAnno", "LTImage", "LTTextLineVertical", "LTTextLine"):
        setattr(pl, nme, type(nme, (), {}))
    sys.modules["pdfminer.layout"] = pl
    pu = sys.modules["pdftable.utils"]
    pu.MathUtils = _Stub("MathUtils")

    class CU:
        @staticmethod
        def sorted_dict(label_dict, key=lambda x: x[1], reverse=True):
            d = OrderedDict()
            for k, v in sorted(label_dict.items(), key=key, reverse=reverse):
                d[k] = v
            return d
    pu.CommonUtils = CU
    T = ref_import("pdftable.model.pdf_table.table_common").TableProcessUtils
    out = {"seed": 11, "cases": []}
    for polys, logi in synth_table_grids(11):
        cells = T.get_table_cell_from_table_logit(table_bboxs=polys, logits=logi, save_html_file=None)
        html, _ = T.cell_to_html(table_cells=cells, first_header=False, add_width=False, add_text=False)
        s_ = "".join(html)
        s_ = f"<html><body>{s_}</body></html>".replace("<td >", "<td>").replace("<tbody>", "").replace("</tbody>", "")
        out["cases"].append({"html": s_, "cells": [[float(c.row_index), float(c.col_index), float(c.row_span), float(c.col_span),
                                                     float(c.x1), float(c.y1), float(c.x2), float(c.y2),
                                                     float(c.width_ratio), float(c.height_ratio)] for c in cells]})
    with open(os.path.join(HERE, "table_html.json"), "w") as f:
        json.dump(out, f)
    print("table_html.json", len(out["cases"]), "cases;", out["cases"][0]["html"][:120])


def gen_convnext_vit():
    """Outputs of the reference's ``ConvNextViT`` (convnext_vit/modeling_convnext_vit.py:20-45) for seeded weights and one
    line (three 300-px chunks).  ``ViTForSTR.forward_features`` raises under the transformers 5.15 of this image
    (``self.vit.get_head_mask`` is gone and ``ViTPatchEmbeddings.forward`` lost ``interpolate_pos_encoding``), so the fixture
    drives the module's OWN sub-modules in the order that method does (modeling_vit.py:64-86: patch embeddings, position
    embeddings [:, 1:], encoder layers, final LayerNorm) and then the stitching + classifier lines of ``forward``
    (modeling_vit.py:131-143) on the module's own ``classifier``.  The CNN half is the reference's ``ConvNextModel`` called
    as its forward calls it.  Weights: the synthetic 4.x-named state_dict renamed to the installed modules' 5.x names and
    loaded with strict=True."""
    import re
    from pdf_table_amd.synth_weights import convnext_vit_state_dict
    mod = ref_import("pdftable.model.convnext_vit.modeling_convnext_vit")
    torch.manual_seed(0)
    model = mod.ConvNextViT().eval()
    v4_to_v5 = [
        (r"vit\.encoder\.layer\.(\d+)\.attention\.attention\.query\.", r"vit.layers.\1.attention.q_proj."),
        (r"vit\.encoder\.layer\.(\d+)\.attention\.attention\.key\.", r"vit.layers.\1.attention.k_proj."),
        (r"vit\.encoder\.layer\.(\d+)\.attention\.attention\.value\.", r"vit.layers.\1.attention.v_proj."),
        (r"vit\.encoder\.layer\.(\d+)\.attention\.output\.dense\.", r"vit.layers.\1.attention.o_proj."),
        (r"vit\.encoder\.layer\.(\d+)\.intermediate\.dense\.", r"vit.layers.\1.mlp.fc1."),
        (r"vit\.encoder\.layer\.(\d+)\.output\.dense\.", r"vit.layers.\1.mlp.fc2."),
        (r"vit\.encoder\.layer\.(\d+)\.layernorm_", r"vit.layers.\1.layernorm_"),
    ]
    sd = {}
    for k, v in convnext_vit_state_dict(seed=29).items():
        for pat, rep in v4_to_v5:
            k = re.sub(pat, rep, k)
        sd[k] = v
    if not any(k.startswith("vitstr.vit.layers.") for k in model.state_dict()):
        sd = convnext_vit_state_dict(seed=29)            # an older transformers: the 4.x names are the module's own
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(131)
    img = rng.integers(0, 256, (3, 32, 300, 3), dtype=np.uint8)       # three chunks of one line, RGB
    img[:, :, 200:, :] //= 4
    x = torch.from_numpy(img).float().div(255.).permute(0, 3, 1, 2)
    with torch.no_grad():
        gray = x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140          # modeling_convnext_vit.py:40
        feats = model.cnn_model(gray).last_hidden_state                              # [3, 512, 1, 75]
        vit = model.vitstr.vit
        emb = vit.embeddings.patch_embeddings(feats)
        emb = emb + vit.embeddings.position_embeddings[:, 1:, :]
        hs = emb
        layers = vit.layers if hasattr(vit, "layers") else vit.encoder.layer
        for layer in layers:
            hs = layer(hs)
            hs = hs[0] if isinstance(hs, tuple) else hs
        seq = vit.layernorm(hs)
        b, s_, e = seq.shape
        ap = seq.view(b // 3, 3, 75, e)
        cat = torch.ones(b // 3, 201, e).type_as(seq)
        cat[:, :69, :] = ap[:, 0, :69, :]
        cat[:, 69:69 + 63, :] = ap[:, 1, 6:-6, :]
        cat[:, 69 + 63:, :] = ap[:, 2, 6:, :]
        logits = model.vitstr.classifier(cat.reshape(-1, e)).view(b // 3, 201, -1)
    top2 = torch.topk(logits, 2, dim=-1)
    out = {"img_u8": img, "seed": np.array(29), "feats_sub": feats[:, ::8, 0, ::5].numpy(), "seq_sub": seq[:, ::5, ::8].numpy(),
           "logits_sub": logits[:, :, ::97].numpy(), "top2_val": top2.values.numpy(), "top2_idx": top2.indices.numpy().astype(np.int32),
           "logits_abs_max": np.array(logits.abs().max().item(), np.float32)}
    np.savez_compressed(os.path.join(HERE, "convnext_vit.npz"), **out)
    print("convnext_vit.npz", {k: v.shape for k, v in out.items()}, "distinct ids", len(set(top2.indices[..., 0].flatten().tolist())))


def gen_mtl_tabnet_backbone():
    """Feature maps of the reference's own ``TableResNetExtra`` (table/mtl_tabnet/table_resnet_extra.py:205-318, configuration of
    mtl_tabnet_config.py:41-53) for seeded weights (loaded strict=True) and a seeded 96x128 input."""
    from pdf_table_amd.synth_weights import mtl_tabnet_backbone_state_dict
    mod = ref_import("pdftable.model.table.mtl_tabnet.table_resnet_extra")
    torch.manual_seed(0)
    gcb = dict(ratio=0.0625, headers=1, att_scale=False, fusion_type="channel_add", layers=[False, True, True, True])
    model = mod.TableResNetExtra(layers=[1, 2, 5, 3], input_dim=3, gcb_config=gcb).eval()
    sd = mtl_tabnet_backbone_state_dict(seed=41)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(141)
    x = rng.standard_normal((1, 3, 96, 128)).astype(np.float32)
    with torch.no_grad():
        f = model(torch.from_numpy(x))
    out = {"x": x, "seed": np.array(41), "f1_sub": f[0][:, ::8, ::3, ::3].numpy(), "f2_sub": f[1][:, ::4, ::2, ::2].numpy(), "f3": f[2][:, ::4].numpy(),
           "shapes": np.array([list(t.shape) for t in f]), "n_params": np.array(sum(p.numel() for p in model.parameters()))}
    np.savez_compressed(os.path.join(HERE, "mtl_tabnet_backbone.npz"), **out)
    print("mtl_tabnet_backbone.npz", {k: v.shape for k, v in out.items()}, int(out["n_params"]), "parameters")


MTL_DEC_CFG = dict(N=3, sos=40, eos=41, pad=42, max_len=12, sos_cell=57, eos_cell=58, pad_cell=59, max_len_cell=6, idx_tag_cell=[3, 5],
                   num_classes=43, num_classes_cell=60)


def gen_mtl_tabnet_decoder():
    """Greedy test-time decode of the reference's own ``MtlTabNetDecoder`` (table/mtl_tabnet/master_decoder.py:194-531; N = 3,
    d_model 512, 8 heads, d_ff 2024 as mtl_tabnet_config.py:59-77; small vocabularies and sequence limits) for seeded weights
    and a seeded feature sequence.  The nested config the reference builds from an mmcv ``ConfigDict`` (``decoder.size`` is read
    as an attribute, :227) is a dict with attribute access here -- the only stand-in."""
    from pdf_table_amd.synth_weights import mtl_tabnet_decoder_state_dict
    mod = ref_import("pdftable.model.table.mtl_tabnet.master_decoder")

    class AttrDict(dict):
        __getattr__ = dict.__getitem__

    c = MTL_DEC_CFG
    att = dict(headers=8, d_model=512, dropout=0.)
    dec = AttrDict(self_attn=dict(att), src_attn=dict(att), feed_forward=dict(d_model=512, d_ff=2024, dropout=0.), size=512, dropout=0.)
    torch.manual_seed(0)
    model = mod.MtlTabNetDecoder(N=c["N"], decoder=dec, d_model=512, num_classes=c["num_classes"], num_classes_cell=c["num_classes_cell"],
                                 start_idx=c["sos"], padding_idx=c["pad"], end_idx=c["eos"], max_seq_len=c["max_len"],
                                 start_idx_cell=c["sos_cell"], padding_idx_cell=c["pad_cell"], end_idx_cell=c["eos_cell"],
                                 max_seq_len_cell=c["max_len_cell"], idx_tag_cell=c["idx_tag_cell"]).eval()
    sd = mtl_tabnet_decoder_state_dict(seed=43, num_classes=c["num_classes"], num_classes_cell=c["num_classes_cell"])
    full = dict(sd)
    for k, v in model.state_dict().items():
        if k.endswith(".pe"):
            full[k] = v                       # the two position tables are buffers computed in __init__
    model.load_state_dict(full, strict=True)
    rng = np.random.default_rng(143)
    fmap = rng.standard_normal((2, 512, 3, 8)).astype(np.float32)       # a backbone feature map, before the positional encoding
    with torch.no_grad():
        feature = mod.PositionalEncoding(d_model=512).eval()(torch.from_numpy(fmap))
        out, box, cells = model(None, feature, None, None, train_mode=False)
    res = {"fmap": fmap, "seed": np.array(43), "tag_logits": out.numpy(), "boxes": box.numpy(), "n_cells": np.array(len(cells)),
           "feature_sub": feature[:, ::3, ::16].numpy()}
    for i, cl in enumerate(cells):
        res[f"cell_{i}"] = cl.numpy()
    np.savez_compressed(os.path.join(HERE, "mtl_tabnet_decoder.npz"), **res)
    print("mtl_tabnet_decoder.npz", {k: v.shape for k, v in res.items()}, "tags", out.argmax(-1).tolist())


def gen_table_master():
    """The reference's own ``TableMasterDecoder`` (table/mtl_tabnet/master_decoder.py:532-645; configuration of table_master_config.py:45-64 with a
    small sequence limit) on seeded weights (strict=True) and seeded feature maps, then ITS host half on those outputs: ``TableMasterConvertor.output_format``
    (master_convertor.py:1000-1030) and ``MasterPostProcessor.__call__`` on ``dict(text, score, bbox)`` as ``TableMaster.simple_test`` builds it
    (table_master.py:669-697) -- logits, boxes and the expected strings / HTML in one fixture."""
    from pdf_table_amd.synth_weights import table_master_decoder_state_dict
    stub_env()
    mod = ref_import("pdftable.model.table.mtl_tabnet.master_decoder")
    conv_mod = ref_import("pdftable.model.table.mtl_tabnet.master_convertor")
    post_mod = ref_import("pdftable.model.table.mtl_tabnet.master_post_processor")
    consts = ref_import("pdftable.model.table.mtl_tabnet.mtl_tabnet_constants")

    class AttrDict(dict):
        __getattr__ = dict.__getitem__

    conv = conv_mod.TableMasterConvertor(dict_file=list(consts.STRUCTURE_ALPHABET_PUBTABNET), max_seq_len=500, start_end_same=False, with_unknown=True)
    ncls, max_len = conv.num_classes(), 14
    att = dict(headers=8, d_model=512, dropout=0.)
    dec = AttrDict(self_attn=dict(att), src_attn=dict(att), feed_forward=dict(d_model=512, d_ff=2024, dropout=0.), size=512, dropout=0.)
    torch.manual_seed(0)
    model = mod.TableMasterDecoder(N=3, decoder=dec, d_model=512, num_classes=ncls, start_idx=conv.start_idx, padding_idx=conv.padding_idx,
                                   max_seq_len=max_len).eval()
    sd = table_master_decoder_state_dict(seed=47, num_classes=ncls)
    full = dict(sd)
    for k, v in model.state_dict().items():
        if k.endswith(".pe"):
            full[k] = v
    model.load_state_dict(full, strict=True)
    rng = np.random.default_rng(147)
    fmap = rng.standard_normal((3, 512, 3, 8)).astype(np.float32)
    with torch.no_grad():
        feature = mod.PositionalEncoding(d_model=512).eval()(torch.from_numpy(fmap))
        out, box = model(None, feature, None, None, train_mode=False)
    metas = [{"scale_factor": (0.75, 0.75), "pad_shape": (480, 480, 3), "ori_shape": (400, 640, 3), "img_shape": (300, 480, 3)} for _ in range(3)]
    # one sample per call, as the reference runs it (processor_mtl_tabnet.py:84-89; _get_pred_bbox_mask builds a ragged array for a mixed batch)
    strings, scores, bboxes = [], [], []
    for b in range(out.shape[0]):
        s1, c1, b1 = conv.output_format(out[b:b + 1], box[b:b + 1], metas[b:b + 1])
        strings.append(s1[0]), scores.append(c1[0]), bboxes.append(b1[0])
    host = []
    post = post_mod.MasterPostProcessor(output_dir=None)
    for s_, sc, bb in zip(strings, scores, bboxes):
        result = dict(text=s_, score=sc, bbox=bb)
        try:
            pred = post(result, file_name=None)
            host.append({"text": s_, "score": float(sc), "bbox_decoded": np.asarray(bb).tolist(), "new_bbox": np.asarray(pred["new_bbox"]).tolist(),
                         "html_context": pred["html_context"], "structure_str": pred["structure_str"], "structure_str_list": pred["structure_str_list"]})
        except IndexError:
            host.append({"raises": "IndexError", "text": s_, "score": float(sc), "bbox_decoded": np.asarray(bb).tolist()})
    res = {"fmap": fmap, "seed": np.array(47), "max_len": np.array(max_len), "tag_logits": out.numpy(), "boxes": box.numpy(),
           "ids": np.array([conv.start_idx, conv.end_idx, conv.padding_idx, ncls]), "host_json": np.array(json.dumps(host, ensure_ascii=False))}
    np.savez_compressed(os.path.join(HERE, "table_master_decoder.npz"), **res)
    print("table_master_decoder.npz", {k: getattr(v, "shape", None) for k, v in res.items()}, "tags", out.argmax(-1).tolist(), [h.get("raises", "ok") for h in host])


def gen_mtl_alphabet():
    """The two vocabularies MtlTabNet's checkpoints are trained against (table/mtl_tabnet/mtl_tabnet_constants.py: 39 structure
    tokens, 277 cell-content tokens) as a DATA file of the package -- a checkpoint's class ids mean nothing without them -- plus
    their sha256 in this directory."""
    c = ref_import("pdftable.model.table.mtl_tabnet.mtl_tabnet_constants")
    d = {"structure": list(c.STRUCTURE_ALPHABET_PUBTABNET), "cell": list(c.TEXTLINE_RECOGNITION_ALPHABET_PUBTABNET)}
    os.makedirs(os.path.join(REPO, "pdf_table_amd", "data"), exist_ok=True)
    blob = json.dumps(d, ensure_ascii=False, indent=0).encode("utf-8")
    with open(os.path.join(REPO, "pdf_table_amd", "data", "mtl_tabnet_alphabet.json"), "wb") as f:
        f.write(blob)
    with open(os.path.join(HERE, "mtl_tabnet_alphabet_hash.json"), "w") as f:
        json.dump({"sha256": hashlib.sha256(json.dumps(d, ensure_ascii=False, sort_keys=True).encode("utf-8")).hexdigest(),
                   "structure": len(d["structure"]), "cell": len(d["cell"])}, f)
    print("mtl_tabnet_alphabet.json", len(d["structure"]), len(d["cell"]))


def gen_mtl_tabnet_host():
    """The reference's own ``MtlTabNetConvertor.output_format`` (table/mtl_tabnet/master_convertor.py:756-784),
    ``MasterPostProcessor.__call__`` (master_post_processor.py:360-401) and ``MtlTabNetPostProcessor.__call__``
    (model/mtl_tabnet/processor_mtl_tabnet.py:108-131) on seeded decoder outputs (regenerated from the seed by the test: only the
    case names, the seed and the EXPECTED results are stored)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from mtl_synth import MTL_HOST_CASES, mtl_host_case_tensors
    stub_env()
    conv_mod = ref_import("pdftable.model.table.mtl_tabnet.master_convertor")
    post_mod = ref_import("pdftable.model.table.mtl_tabnet.master_post_processor")
    consts = ref_import("pdftable.model.table.mtl_tabnet.mtl_tabnet_constants")
    ocu = sys.modules["pdftable.utils.ocr"].OcrCommonUtils
    conv = conv_mod.MtlTabNetConvertor(dict_file=list(consts.STRUCTURE_ALPHABET_PUBTABNET), max_seq_len=500, start_end_same=False,
                                       with_unknown=True, cell_dict_file=list(consts.TEXTLINE_RECOGNITION_ALPHABET_PUBTABNET),
                                       max_seq_len_cell=150)
    like = {"char2idx": conv.char2idx, "char2idx_cell": conv.char2idx_cell, "end_idx": conv.end_idx, "end_idx_cell": conv.end_idx_cell,
            "padding_idx_cell": conv.padding_idx_cell}
    ids = {k: int(getattr(conv, k)) for k in ("start_idx", "end_idx", "padding_idx", "unknown_idx", "start_idx_cell", "end_idx_cell",
                                              "padding_idx_cell", "unknown_idx_cell")}
    ids.update(num_classes=conv.num_classes(), num_classes_cell=conv.num_classes_cell(), idx_tag_cell=conv.idx_tag_cell())
    out = {"seed": 5150, "convertor": ids, "cases": {}}
    post = post_mod.MasterPostProcessor(output_dir=None)
    for k, name in enumerate(MTL_HOST_CASES):
        tag, box, cell, meta = mtl_host_case_tensors(name, like, out["seed"] + k)
        strings, scores, bboxes, cell_strings, cell_scores = conv.output_format(torch.from_numpy(tag), torch.from_numpy(box),
                                                                                [torch.from_numpy(cell)], [meta])
        result = dict(text=strings[0], score=scores[0], bbox=bboxes[0], cell=cell_strings[0])
        try:
            pred = post(result, file_name=None)
        except IndexError:
            # no box survives `sum(item) > 1`: np.array([]) has no axis to index in box_transform (master_post_processor.py:360) --
            # the reference raises; what it had computed up to there is recorded
            out["cases"][name] = {"raises": "IndexError", "text": strings[0], "score": float(scores[0]), "cell": list(cell_strings[0]),
                                  "pred_html": result["pred_html"], "html_context": result["html_context"],
                                  "structure_str": result["structure_str"], "structure_str_list": result["structure_str_list"]}
            print(name, "-> IndexError in box_transform")
            continue
        polygons = ocu.box_list_two_point_to_four_point(pred["new_bbox"])
        out["cases"][name] = {"text": strings[0], "score": float(scores[0]), "bbox_decoded": np.asarray(bboxes[0]).tolist(),
                              "cell": list(cell_strings[0]), "cell_scores": [float(v) for v in cell_scores[0]],
                              "bbox_kept": np.asarray(pred["bbox"]).tolist(), "new_bbox": np.asarray(pred["new_bbox"]).tolist(),
                              "polygons": np.asarray(polygons).tolist(), "pred_html": pred["pred_html"], "html_context": pred["html_context"],
                              "structure_str": pred["structure_str"], "structure_str_list": pred["structure_str_list"]}
        print(name, "->", pred["pred_html"][:100], len(pred["new_bbox"]), "boxes")
    with open(os.path.join(HERE, "mtl_tabnet_host.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)
    print("mtl_tabnet_host.json", len(out["cases"]), "cases")


def gen_e2e_page():
    """Two 1024 x 1024 synthetic pages through the composed oracle chain (oracle/e2e.py: fp32, one call per line / table like the
    reference) -> tests/golden/e2e_page.npz: boxes, token ids + top-2 margins, cells + logical locations per table.  Minutes of CPU."""
    import time
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from e2e_synth import E2E_PAGES, e2e_state_dicts, e2e_table_boxes
    from oracle import e2e
    from pdf_table_amd.synth_pages import make_page
    torch.set_num_threads(os.cpu_count() or 1)
    sds = e2e_state_dicts()
    res = {"pages": np.array(E2E_PAGES)}
    # the reference's OWN structure + text -> HTML code over the oracle chain's cells, boxes and strings: what the engine's
    # table_html is compared with in tests/test_gpu_e2e.py / tests/e2e_agreement.py -- no product code on this side of the comparison
    Task, T, OcrCell = _matcher_env()
    task = Task(output_dir="/tmp/pt_golden_html")
    from oracle.crnn import ctc_greedy_ids
    vocab = {i + 1: chr(0x4E00 + i) for i in range(7643)}       # pdf_table_amd.rec_stage.synthetic_vocab: id k -> the k-th CJK code point (no vocab.txt offline)
    for pi, idx in enumerate(E2E_PAGES):
        page, meta = make_page(idx, 1024)
        t0 = time.time()
        r = e2e.page_chain(page, sds, None)          # table regions: the layout stage's own (the reference's flow)
        tb = r["table_boxes"]
        gen = e2e_table_boxes(meta)
        print(f"page {idx}: layout tables {tb.tolist()}, the generator's rectangles grown by 8 px {gen.tolist()}")
        print(f"page {idx}: {len(r['det_boxes'])} boxes, {len(r['layout'])} layout regions, tables {[t['n'] for t in r['tables']]} cells, "
              f"{r['det_prob_near_thresh']} prob pixels within 1e-3 of the threshold, {time.time() - t0:.0f} s")
        p = f"p{pi}_"
        res[p + "table_boxes"] = tb
        res[p + "det_boxes"] = r["det_boxes"].astype(np.float32)
        res[p + "rec_ids"], res[p + "rec_margin"], res[p + "rec_win"] = r["rec_ids"], r["rec_margin"].astype(np.float32), r["rec_win"].astype(np.float32)
        res[p + "layout_bbox"] = np.array([it["bbox"] for it in r["layout"]], np.float32).reshape(-1, 4)
        res[p + "layout_score"] = np.array([it["score"] for it in r["layout"]], np.float32)
        res[p + "layout_cat"] = np.array([it["category_id"] for it in r["layout"]], np.int32)
        res[p + "n_tables"] = np.array(len(r["tables"]))
        texts = ["".join(vocab.get(int(t_), "") for t_ in row) for row in ctc_greedy_ids(r["rec_ids"])]
        res[p + "rec_text"] = np.array(texts)
        for ti, t in enumerate(r["tables"]):
            for k in ("polys", "scores", "stacked", "logi", "fragile"):
                if k in t:
                    res[f"{p}t{ti}_{k}"] = t[k]
            if not t["n"]:
                continue
            # ocr_system_task.py:192-199 -> OcrTableToHtmlTask (ocr_table_to_html_task.py:79-176): cells in page pixels (shifted by the crop corner,
            # table_common.py:1811-1825), the page's lines whose centre lies in the layout box, image pages with ocr_post_process=True (:93)
            off = np.tile(tb[ti][:2].astype(np.float64), 4)[None]
            cells = T.get_table_cell_from_table_logit(table_bboxs=t["polys"].astype(np.float64) + off, logits=t["logi"], save_html_file=None)
            ocr = [OcrCell(raw_data={"index": i + 1, "text": tx, "bbox": np.asarray(q, np.float64).reshape(4, 2).tolist()}) for i, (q, tx) in enumerate(zip(r["det_boxes"], texts))]
            inside, _ = T.get_text_in_table_bbox(bbox=[float(v) for v in tb[ti]], ocr_results=ocr, diff=2)
            _, html, _, db_html = task.match_table_cell_and_text_cell(table_idx=ti, table_cells=cells, text_bboxs=inside, raw_filename="g",
                                                                      ocr_post_process=True)
            res[f"{p}t{ti}_html"] = np.array(list(html))
            fr = t["stacked"] - np.floor(t["stacked"])
            print(f"   table {ti}: {t['n']} cells, {len(inside)} lines inside, HTML {sum(len(x) for x in html)} characters; closest logical location to the "
                  f".5 boundary {float(np.abs(fr - 0.5).min()):.2e}; {int(t['fragile'].sum())} fragile peaks")
    np.savez_compressed(os.path.join(HERE, "e2e_page.npz"), **res)
    print("e2e_page.npz", os.path.getsize(os.path.join(HERE, "e2e_page.npz")) // 1024, "KiB")


def gen_mtl_lengths():
    """the oracle's greedy decode at the reference's configured sequence limits (500 structure / 150 cell positions) for the two tables of
    tests/test_gpu_mtl.py::test_decoders_at_the_configured_lengths_bf16x3 -> mtl_tabnet_lengths.npz (tag logits, boxes, cell logits).  ~3 minutes of CPU."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import test_gpu_mtl as T
    torch.set_num_threads(os.cpu_count() or 1)
    g, fmap, cfg, sd = T.lengths_case(HERE)
    want = T._oracle_decode(sd, fmap, cfg)
    res = {"seed": np.array(int(g["seed"])), "fmap_digest": T._digest(fmap)}
    for b, (tag, box, cells) in enumerate(want):
        res[f"tag{b}"], res[f"box{b}"], res[f"cells{b}"] = tag.astype(np.float32), box.astype(np.float32), cells.astype(np.float32)
        print(f"table {b}: {tag.shape[0]} structure positions, {cells.shape[0]} cells x {cells.shape[1]} positions")
    np.savez_compressed(os.path.join(HERE, "mtl_tabnet_lengths.npz"), **res)
    print("mtl_tabnet_lengths.npz", os.path.getsize(os.path.join(HERE, "mtl_tabnet_lengths.npz")) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["db", "crnn", "registry", "ctc", "host", "lore_dla", "lore_decode", "lore_processor", "picodet",
                             "table_html"]
    if "rec_pp" in which or not sys.argv[1:]:
        gen_rec_pp()
    if "table_html" in which:
        gen_table_html()
    if "picodet" in which:
        gen_picodet()
    if "lore_processor" in which:
        gen_lore_processor()
    if "lore_decode" in which:
        gen_lore_decode()
    if "lore_dla" in which:
        gen_lore_dla34()
    if "lore_wireless" in which or not sys.argv[1:]:
        gen_lore_wireless()
    if "host" in which:
        gen_db_host_numpy()
    if "db" in which:
        gen_db_resnet18()
    if "db_nas" in which or not sys.argv[1:]:
        gen_db_nas()
    if "pplcnet" in which or not sys.argv[1:]:
        gen_pplcnet()
    if "table_text_match" in which or not sys.argv[1:]:
        gen_table_text_match()
    if "crnn" in which:
        gen_crnn()
    if "registry" in which:
        gen_registry_hash()
    if "ctc" in which:
        gen_ctc()
    if "convnext_vit" in which or not sys.argv[1:]:
        gen_convnext_vit()
    if "mtl_tabnet" in which or not sys.argv[1:]:
        gen_mtl_tabnet_backbone()
    if "mtl_tabnet_decoder" in which or not sys.argv[1:]:
        gen_mtl_tabnet_decoder()
    if "table_master" in which or not sys.argv[1:]:
        gen_table_master()
    if "mtl_lengths" in which:   # minutes of CPU: only on request
        gen_mtl_lengths()
    if "e2e" in which:           # minutes of CPU: only on request
        gen_e2e_page()
    if "mtl_tabnet_host" in which or not sys.argv[1:]:
        gen_mtl_alphabet()
        gen_mtl_tabnet_host()
