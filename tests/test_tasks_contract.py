"""Plug-in contract of the stage tasks (mirror of base_infer_task.py / ocr_*_task.py in the reference): registry,
constructor errors, model-id resolution.  The registry hash is pinned to the reference's TABLE_MODEL_DICT."""
import hashlib
import json
import os

import numpy as np
import pytest

from pdf_table_amd.base_infer_task import BaseInferTask
from pdf_table_amd.ocr_table_model_config import HIP_SUPPORTED, TABLE_MODEL_DICT


def test_registry_equals_reference(golden_dir):
    want = json.load(open(os.path.join(golden_dir, "registry_hash.json")))
    blob = json.dumps(TABLE_MODEL_DICT, sort_keys=True, ensure_ascii=False).encode("utf-8")
    assert hashlib.sha256(blob).hexdigest() == want["sha256"]
    assert sorted(TABLE_MODEL_DICT) == want["providers"]
    for prov, task, model in HIP_SUPPORTED:
        assert model in TABLE_MODEL_DICT[prov][task]


class _Dummy(BaseInferTask):
    def _construct_model(self, model): self._model = lambda **kw: "ran"
    def _build_processor(self): pass
    def _preprocess(self, inputs, **kw): return {"inputs": [inputs]}
    def _run_model(self, inputs, **kw):
        r, el = self.infer({})
        inputs["results"] = [r]
        return inputs
    def _postprocess(self, inputs, **kw): return inputs["results"]


class _Cfg:
    backbone = "resnet18"
    recognizer = "CRNN"
    task_type = "general"
    model_name = "Lore"


def test_base_task_call_chain_and_predictors():
    t = _Dummy(model="db", task="ocr_detection")
    t._config = _Cfg()
    t._get_inference_model()
    assert t("x") == ["ran"]
    with pytest.raises(RuntimeError, match="TensorRt infer not supported"):
        _Dummy(model="db", task="ocr_detection", predictor_type="trt")._get_inference_model()
    for pt in ("pytorch", "onnx"):
        with pytest.raises(RuntimeError, match="no fallback"):
            _Dummy(model="db", task="ocr_detection", predictor_type=pt)._get_inference_model()


def test_model_id_resolution_follows_reference_rules():
    t = _Dummy(model="db", task="ocr_detection")
    t._config = _Cfg()
    assert t.get_model_name_or_path() == "cycloneboy/cv_resnet18_ocr-detection-db-line-level_damo"     # hf_model wins
    t.use_modelscope_hub = True
    assert t.get_model_name_or_path() == "damo/cv_resnet18_ocr-detection-db-line-level_damo"
    r = _Dummy(model="CRNN", task="ocr_recognition")
    r._config = _Cfg()
    assert r.get_model_name_or_path() == "cycloneboy/cv_crnn_ocr-recognition-general_damo"
    p = _Dummy(model="db_pp", task="ocr_detection", model_provider="PaddleOCR", lang="fr")
    p._config = type("C", (), {"backbone": "PP-OCRv4"})()
    assert p.get_model_name_or_path() == "cycloneboy/Multilingual_PP-OCRv3_det_infer"                   # lang -> "ml"
    q = _Dummy(model="PP-OCRv4", task="ocr_recognition", model_provider="PaddleOCR", lang="ka")
    q._config = type("C", (), {"backbone": "PP-OCRv4"})()
    assert q.get_model_name_or_path() == "cycloneboy/en_PP-OCRv3_rec_infer"      # v4 -> v3 for non ch/en, then lang -> en
    s = _Dummy(model="db_pp", task="ocr_detection", model_provider="PaddleOCR", lang="ch", server_model=True)
    s._config = type("C", (), {"backbone": "PP-OCRv4"})()
    assert s.get_model_name_or_path() == "cycloneboy/ch_PP-OCRv4_det_server_infer"


def test_unknown_models_raise_like_the_reference():
    from pdf_table_amd.ocr_detection_task import OcrDetectionTask
    from pdf_table_amd.ocr_recognition_task import OcrRecognitionTask
    with pytest.raises(RuntimeError, match="current model is not supported"):
        OcrDetectionTask(model="yolo")
    with pytest.raises(RuntimeError, match="current model is not supported"):
        OcrRecognitionTask(model="tesseract")


def test_lore_host_logic_matches_oracle():
    """tsr_stage's host pieces (affine geometry, int64 meta, quad back-projection, logical rounding, config presets)
    against the oracle restatements, which are pinned by the reference goldens."""
    import torch
    from oracle import lore_decode as od
    from oracle import lore_pre
    from pdf_table_amd import tsr_stage as ts
    for (h, w, ih, iw) in [(320, 360, 1024, 1024), (300, 420, 1024, 1024), (777, 512, 512, 512), (51, 1999, 1024, 1024), (1024, 1024, 1024, 1024)]:
        t1, m1 = od.lore_preprocess_geometry(h, w, ih, iw)
        mi, m2 = ts.lore_geometry(h, w, ih, iw)
        assert np.array_equal(m1, m2) and np.array_equal(lore_pre.invert_affine(t1), mi)
        q = np.random.default_rng(h).uniform(-5, iw / 4 + 5, (40, 8)).astype(np.float32)
        ref = np.concatenate([od.transform_preds(q[:, 2 * k:2 * k + 2], m1[:2], m1[2], (m1[6], m1[5])) for k in range(4)], 1)
        assert np.array_equal(ts.transform_quads(q, m2), ref.astype(np.float32))
    for (h, w, ih, iw) in [(500, 333, 768, 768), (300, 420, 768, 768)]:          # the 'wireless' upper-left geometry
        t1, m1 = od.lore_preprocess_geometry(h, w, ih, iw, upper_left=True)
        mi, m2 = ts.lore_geometry(h, w, ih, iw, upper_left=True)
        assert np.array_equal(m1, m2) and np.array_equal(lore_pre.invert_affine(t1), mi) and m2[:3].tolist() == [0, 0, max(h, w)]
        q = np.random.default_rng(w).uniform(-5, iw / 4 + 5, (40, 8)).astype(np.float32)
        ref = np.concatenate([od.transform_preds(q[:, 2 * k:2 * k + 2], m1[:2], m1[2], (m1[6], m1[5]), True) for k in range(4)], 1)
        assert np.array_equal(ts.transform_quads(q, m2, True), ref.astype(np.float32))
    lg = np.random.default_rng(1).uniform(-1, 12, (50, 4)).astype(np.float32)
    lg[:4, 0] = [2.5, 3.5, 0.5, 7.500001]
    assert np.array_equal(ts.process_logic_output(lg), od.process_logic_output(torch.from_numpy(lg)).numpy())
    wtw, ptn = ts.LoreConfig(task_type="wtw"), ts.LoreConfig(task_type="ptn")
    assert (wtw.resolution, wtw.wiz_rev, wtw.wiz_2dpe, wtw.vis_thresh, wtw.tsfm_layers) == ((1024, 1024), True, False, 0.2, 4)
    assert (ptn.resolution, ptn.wiz_rev, ptn.wiz_2dpe, ptn.vis_thresh, ptn.tsfm_layers) == ((512, 512), False, True, 0.35, 3)
    wl = ts.LoreConfig(task_type="wireless")
    assert (wl.backbone, wl.resolution, wl.upper_left, wl.wiz_rev, wl.wiz_2dpe) == ("ResNet-18", (768, 768), True, False, True)


def test_layout_decode_matches_oracle_postprocess():
    """LayoutStage.decode_page on the compacted candidates == the oracle's restatement of OCRPicodetPostProcessor on the
    full head outputs (pinned bit-exactly to the reference): same boxes, labels, scores, order."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lore_synth import synth_pico_heads
    from oracle import picodet as op
    from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig, layout_tables
    for seed, tgt, org in ((1, (160, 128), (1024, 1024)), (2, (800, 608), (1100, 850)), (7, (800, 608), (1024, 1024))):
        sc, bx = synth_pico_heads(seed, tgt)
        logits = [torch.logit(torch.from_numpy(s).double()).float().numpy() for s in sc]
        sc = [torch.sigmoid(torch.from_numpy(l)).numpy() for l in logits]              # scores as the net would emit them
        ref = op.picodet_postprocess(sc, bx, list(org), [float(tgt[0]) / org[0], float(tgt[1]) / org[1]], list(tgt), op.LABELS["en"])
        cfg = PicodetConfig(task_type="en")
        cfg.img_height, cfg.img_width = tgt
        st = LayoutStage(None, cfg)
        recs = []
        for l in range(4):
            keep = np.nonzero(sc[l][0].max(1) > cfg.score_threshold - 1e-3)[0]
            r = np.zeros((len(keep), 48), np.float32)
            r[:, 0] = np.full(len(keep), l, np.int32).view(np.float32)
            r[:, 1] = keep.astype(np.int32).view(np.float32)
            r[:, 2:7] = logits[l][0, keep]
            r[:, 7:39] = bx[l][0, keep]
            recs.append(r)
        rng = np.random.default_rng(seed)
        rec = np.concatenate(recs)[rng.permutation(sum(len(r) for r in recs))]         # device order is arbitrary
        got = st.decode_page(rec, org)
        assert len(got) == len(ref) > 10
        assert [g["category_id"] for g in got] == [r["category_id"] for r in ref]
        assert np.array_equal(np.array([g["bbox"] for g in got]), np.array([r["bbox"] for r in ref]))
        assert np.array_equal(np.array([g["score"] for g in got]), np.array([r["score"] for r in ref]))
    tabs = layout_tables(got, "table", 0.2)
    assert all(t["label"] == "table" for t in tabs) and [t["bbox"][1] for t in tabs] == sorted(t["bbox"][1] for t in tabs)


def test_table_html_equals_reference(golden_dir):
    """cells (order, indices, spans, geometry ratios) and the structure HTML equal the reference's own
    get_table_cell_from_table_logit + cell_to_html on seeded grids (tests/golden/table_html.json)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lore_synth import synth_table_grids
    from pdf_table_amd.table_html import cells_to_structure_html, structure_html, table_cells_from_logits
    with open(os.path.join(golden_dir, "table_html.json")) as f:
        gold = json.load(f)
    for (polys, logi), ref in zip(synth_table_grids(gold["seed"]), gold["cases"]):
        cells = table_cells_from_logits(polys, logi)
        got = [[float(c.row_index), float(c.col_index), float(c.row_span), float(c.col_span), float(c.x1), float(c.y1),
                float(c.x2), float(c.y2), float(c.width_ratio), float(c.height_ratio)] for c in cells]
        assert np.allclose(np.array(got), np.array(ref["cells"]), rtol=1e-6, atol=1e-6)
        assert np.array_equal(np.array(got)[:, :4], np.array(ref["cells"])[:, :4])
        assert cells_to_structure_html(cells) == ref["html"]
        assert structure_html(polys, logi) == ref["html"]          # the fast path used by TsrStage
    assert "colspan" in "".join(r["html"] for r in gold["cases"]) and "rowspan" in "".join(r["html"] for r in gold["cases"])


# ---- text <-> cell matching and table HTML ------------------------------------------------------------
def test_table_text_match_equals_reference(golden_dir):
    """pdf_table_amd.table_text_match against the reference's own OcrTableToHtmlTask on seeded tables and OCR lines
    (tests/golden/table_text_match.json, generator: tests/golden/make_golden.py table_text_match)"""
    import json
    import os
    import numpy as np
    from lore_synth import synth_table_grids, synth_table_texts
    from pdf_table_amd import table_text_match as M
    from pdf_table_amd.table_html import table_cells_from_logits
    with open(os.path.join(golden_dir, "table_text_match.json")) as f:
        gold = json.load(f)
    grids = synth_table_grids(gold["seed"])
    assert len(gold["cases"]) == 2 * len(grids)
    for g in gold["cases"]:
        polys, logi = grids[g["case"]]
        boxes, texts = synth_table_texts(g["case"], polys)
        cells = table_cells_from_logits(polys, logi)
        tb = M.text_boxes(boxes)
        inside = M.texts_in_table(g["bbox"], tb, diff=2)
        assert [int(i) + 1 for i in inside] == g["inside"]
        cb = np.array([[c.x1, c.y1, c.x2, c.y2] for c in cells], np.float64)
        assert M.find_top1_match(tb[inside], cb).tolist() == g["top1"]
        res = M.match_table_cells_and_text(cells, tb[inside], [texts[i] for i in inside], post_process=g["ocr_post_process"])
        assert [[float(c.row_index), float(c.col_index), c.text] for c in res] == g["cells"]
        html, db = M.cells_to_html(res)
        assert html == g["html"] and db == g["db_html"]


def test_ocr_post_process_known_answers():
    from pdf_table_amd.table_text_match import ocr_post_process
    assert ocr_post_process("o") == "0" and ocr_post_process(" O ") == "0"
    assert ocr_post_process("1.000.000") == "1,000.000" and ocr_post_process("1.5") == "1.5"
    assert ocr_post_process("total") == "total" and ocr_post_process("") == ""


def test_page_table_html_uses_one_coordinate_frame():
    """A 3x3 table whose crop starts at (300, 200) on the page: the structure stage's quads are relative to the crop,
    the OCR lines are in page pixels.  After the reference's shift by the crop corner (convert_table_sep_to_merge,
    pdf_table/table_common.py:1811-1825 -- what recognize_tables(page_frame=True) delivers) every line lands in the cell
    that contains it; with crop-relative quads (the round-1 bug) it does not."""
    import numpy as np
    from pdf_table_amd.table_text_match import page_table_html
    x0, y0, cw, ch = 300, 200, 100.0, 40.0
    polys, logi = [], []
    for r in range(3):
        for c in range(3):
            xa, ya, xb, yb = c * cw, r * ch, (c + 1) * cw, (r + 1) * ch
            polys.append([xa, ya, xb, ya, xb, yb, xa, yb])           # TL, TR, BR, BL in crop pixels
            logi.append([c, c, r, r])                                # left, right, top, bottom
    polys, logi = np.array(polys, np.float32), np.array(logi, np.float32)
    quads, texts = [], []
    for r in range(3):
        for c in range(3):
            xa, ya = x0 + c * cw + 10, y0 + r * ch + 10              # one line inside every cell, page pixels
            quads.append([xa, ya, xa + 60, ya, xa + 60, ya + 18, xa, ya + 18])
            texts.append(f"r{r}c{c}")
    quads = np.array(quads, np.float64)
    box = [x0, y0, x0 + 3 * cw, y0 + 3 * ch]
    page_polys = polys.astype(np.float64) + np.tile([x0, y0], 4)[None]
    html, db = page_table_html(page_polys, logi, box, quads, texts)
    tds = [row for row in html if row.startswith("<td")]
    assert len(tds) == 9
    for i, row in enumerate(tds):
        assert row.endswith(f">test_textr{i // 3}c{i % 3}</td>"), row   # Cell.text appends to the structure stage's "test_text"
    wrong, _ = page_table_html(polys, logi, box, quads, texts)          # crop-relative quads against page-frame text
    assert [r for r in wrong if r.startswith("<td")] != tds


def test_dropped_table_boxes_keep_the_callers_indexing():
    """A degenerate caller-supplied table box is skipped with a log line; the page's table results stay index-aligned with the caller's list
    (None at the dropped position) instead of silently shifting."""
    import numpy as np
    from pdf_table_amd.pipeline import OcrTablePipeline
    tb = [np.array([[10, 10, 200, 120], [300, 50, 300, 90], [40, 400, 500, 700]]), np.zeros((0, 4), np.int64), np.array([[5, 5, 60, 60]])]
    kept = []
    out = OcrTablePipeline._drop_empty_crops(tb, (1024, 1024), kept)
    assert [len(b) for b in out] == [2, 0, 1]
    tsr = [["t0", "t2"], [], ["u0"]]
    assert OcrTablePipeline._realign_tables(tsr, kept) == [["t0", None, "t2"], [], ["u0"]]
    assert OcrTablePipeline._realign_tables(tsr, None) is tsr
