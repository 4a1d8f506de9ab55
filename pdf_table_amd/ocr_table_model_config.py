"""Model registry: ``TABLE_MODEL_DICT[provider][task][model][lang | task_type] -> {hub ids ...}``.

Same content as the reference registry (src/pdftable/model/ocr_pdf/ocr_table_model_config.py:16-347; equality is
pinned by a sha256 over the canonical JSON in tests/golden/registry_hash.json), but generated from naming rules
rather than spelled out, and extended with the engine's own ``"hip"`` entries (which checkpoints the HIP engine
can serve today).  Look-up rules live in ``base_infer_task.BaseInferTask`` exactly as in the reference
(base_infer_task.py:228-303).
"""
from __future__ import annotations

__all__ = ["TABLE_MODEL_DICT", "HIP_SUPPORTED"]

_OSS = "https://modelscope.oss-cn-beijing.aliyuncs.com/test/images/"
_DUGUANG = "http://duguang-labelling.oss-cn-shanghai.aliyuncs.com/"
_REC_IMG = _DUGUANG + "mass_img_tmp_20220922/ocr_recognition.jpg"
_GDRIVE = "https://drive.google.com/file/d/{}/view?usp=sharing"
_ME = "cycloneboy/"


def _scope_entry(kind_key, kind_val, repo, image_url, org="damo/"):
    return {kind_key: kind_val, "model": org + repo, "hf_model": _ME + repo, "image_url": image_url}


def _model_scope():
    det = {bb: {"general": _scope_entry("backbone", bb, f"cv_{bb}_ocr-detection-db-line-level_damo",
                                        _OSS + "ocr_detection.jpg")}
           for bb in ("resnet18", "proxylessnas")}
    rec = {
        "CRNN": {"general": _scope_entry("recognizer", "CRNN", "cv_crnn_ocr-recognition-general_damo", _REC_IMG)},
        "LightweightEdge": {"general": _scope_entry("recognizer", "LightweightEdge",
                                                    "cv_LightweightEdge_ocr-recognitoin-general_damo", _REC_IMG)},
    }
    cvit_img = {
        "general": _REC_IMG,
        "handwritten": _DUGUANG + "mass_img_tmp_20220922/ocr_recognition_handwritten.jpg",
        "document": _DUGUANG + "mass_img_tmp_20220922/ocr_recognition_document.png",
        "licenseplate": _DUGUANG + "mass_img_licenseplate//ocr_recognition_licenseplate.jpg",
        "scene": _REC_IMG,
    }
    rec["ConvNextViT"] = {
        t: _scope_entry("recognizer", "ConvNextViT", f"cv_convnextTiny_ocr-recognition-{t}_damo", img, org=_ME)
        for t, img in cvit_img.items()}
    lineless = _OSS + "lineless_table_recognition.jpg"
    lore_repo = "cv_{}-transformer_table-structure-recognition_lore"
    lore = {
        "wireless": {
            "backbone": "ResNet-18",
            "model": _ME + lore_repo.format("resnet") + "_wireless",
            "model_hub": "damo/" + lore_repo.format("resnet"),
            "model_hub2": "iic/" + lore_repo.format("resnet"),
            "model_url": _GDRIVE.format("1cBaewRwlZF1tIZovT49HpJZ5wlb3nSCw"),
            "image_url": lineless,
        },
    }
    for task_type, suffix, gid in (("wtw", "wtw", "1n33c9jmGmjSfRbheleE1pqiIXBb_BCEw"),
                                   ("ptn", "ptn", "1hg5R42u_6xaoO-6Ft18Ctu86HB_N2Bzu"),
                                   ("PubTabNet", "ptn", "1hg5R42u_6xaoO-6Ft18Ctu86HB_N2Bzu")):
        lore[task_type] = {"backbone": "DLA-34", "model": _ME + lore_repo.format("dla34") + "_" + suffix,
                           "model_url": _GDRIVE.format(gid), "image_url": lineless}
    tsr = {
        "CenterNet": {"wtw": _scope_entry("backbone", "dla34", "cv_dla34_table-structure-recognition_cycle-centernet",
                                          _OSS + "table_recognition.jpg", org="iic/")},
        "Lore": lore,
    }
    layout = {"DocXLayout": {"general": {"backbone": "dla34", "model": _ME + "cv_dla34_layout-analysis_docxlayout_general",
                                         "image_url": _OSS + "table_recognition.jpg"}}}
    return {"detection": det, "recognition": rec, "table_structure": tsr, "layout": layout}


def _m(repo):
    return {"model": _ME + repo}


def _paddle():
    det = {
        "PP-OCRv4": {
            "ch": {"model": _ME + "ch_PP-OCRv4_det_infer", "server_model": _ME + "ch_PP-OCRv4_det_server_infer"},
            "en": _m("en_PP-OCRv3_det_infer"),          # sic: the "v4/en" entry points at the v3 English model
            "ml": _m("Multilingual_PP-OCRv3_det_infer"),
        },
        "PP-OCRv3": {"ch": _m("ch_PP-OCRv3_det_infer"), "en": _m("en_PP-OCRv3_det_infer"),
                     "ml": _m("Multilingual_PP-OCRv3_det_infer")},
        "PP-Table": {"en": _m("en_ppocr_mobile_v2.0_table_det_infer")},
    }
    langs = ["ch", "en", "korean", "japan", "chinese_cht", "ta", "te", "ka", "latin", "arabic", "cyrillic", "devanagari"]
    v3_only = {"chinese_cht", "latin", "cyrillic"}      # no v4 recogniser published for these
    rec_v4 = {}
    for lg in langs:
        rec_v4[lg] = _m(f"{lg}_PP-OCRv{3 if lg in v3_only else 4}_rec_infer")
    rec_v4["ch"]["server_model"] = _ME + "ch_PP-OCRv4_rec_server_infer"
    rec = {
        "PP-OCRv4": rec_v4,
        "PP-OCRv3": {lg: _m(f"{lg}_PP-OCRv3_rec_infer") for lg in langs},
        "PP-Table": {"en": _m("en_ppocr_mobile_v2.0_table_rec_infer")},
    }
    tsr = {"SLANet": {lg: _m(f"{lg}_ppstructure_mobile_v2.0_SLANet_infer") for lg in ("ch", "en")}}
    cls_image = {"PPLCNet": {k: _m("cv_cls_pulc_" + k) for k in
                             ("table_attribute", "text_image_orientation", "textline_orientation",
                              "language_classification")}}
    layout = {"LCNet": {"ch": _m("picodet_lcnet_x1_0_fgd_layout_cdla_infer"),
                        "en": _m("picodet_lcnet_x1_0_fgd_layout_infer"),
                        "table": _m("picodet_lcnet_x1_0_fgd_layout_table_infer")}}
    formula = {"latex": {"en": _m("rec_latex_ocr_infer"), "ch": _m("rec_latex_ocr_infer")}}
    return {"detection": det, "recognition": rec, "table_structure": tsr, "cls_image": cls_image, "layout": layout,
            "formula": formula}


def _other():
    def e(backbone, repo):
        return {"backbone": backbone, "model": _ME + repo}
    tsr = {
        "Lgpma": {"PubTabNet": e("ResNet", "en_table_structure_lgpma_pubtabnet")},
        "MtlTabNet": {"PubTabNet": e("TableResNetExtra", "en_table_structure_mtltabnet_pubtabnet"),
                      "FinTabNet": e("TableResNetExtra", "en_table_structure_mtltabnet_fintabnet")},
        "TableMaster": {"PubTabNet": e("TableResNetExtra", "en_table_structure_tablemaster_pubtabnet")},
        "LineCell": {"PubTabNet": e("opencv", "line_cell")},
        "LineCellPdf": {"PubTabNet": e("digital_pdf", "line_cell_pdf")},
    }
    return {"table_structure": tsr, "config": {"Pdftable": {"font": {"model": _ME + "pdftable_config"}}}}


TABLE_MODEL_DICT = {"model_scope": _model_scope(), "PaddleOCR": _paddle(), "Other": _other()}

# (provider, task, model) triples whose ARCHITECTURE the HIP engine implements (state_dict layouts of the in-tree
# torch modules).  The PaddleOCR / PicoDet entries are ONNX graphs that are not in the reference tree
# (SURVEY.md finding F2) and need the ONNX importer planned as section 8f-3.
HIP_SUPPORTED = {
    ("model_scope", "detection", "resnet18"): "PT_MODEL_DB_RESNET18",
    ("model_scope", "recognition", "CRNN"): "PT_MODEL_CRNN",
}
