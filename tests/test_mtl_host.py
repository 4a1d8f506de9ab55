"""Host half of MtlTabNet (pdf_table_amd/mtl_stage.py: label convertor + HTML post-processor) against what the reference's OWN
``MtlTabNetConvertor.output_format`` / ``MasterPostProcessor.__call__`` / ``box_list_two_point_to_four_point`` produced for the same
seeded decoder outputs (tests/golden/mtl_tabnet_host.json, make_golden.py::gen_mtl_tabnet_host)."""
import hashlib
import json
import os

import numpy as np
import pytest

from mtl_synth import MTL_HOST_CASES, mtl_host_case_tensors
from pdf_table_amd import mtl_stage as M


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "mtl_tabnet_host.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def conv():
    return M.MtlTabNetConvertor(max_seq_len=500, max_seq_len_cell=150)


def test_alphabet_is_the_reference_one(golden_dir):
    s, c = M.load_alphabets()
    with open(os.path.join(golden_dir, "mtl_tabnet_alphabet_hash.json")) as f:
        h = json.load(f)
    d = {"structure": s, "cell": c}
    assert (len(s), len(c)) == (h["structure"], h["cell"]) == (39, 277)
    assert hashlib.sha256(json.dumps(d, ensure_ascii=False, sort_keys=True).encode("utf-8")).hexdigest() == h["sha256"]


def test_special_ids_equal_the_reference_convertor(conv, golden):
    want = golden["convertor"]
    for k in ("start_idx", "end_idx", "padding_idx", "unknown_idx", "start_idx_cell", "end_idx_cell", "padding_idx_cell", "unknown_idx_cell"):
        assert getattr(conv, k) == want[k], k
    assert conv.num_classes() == want["num_classes"] == 43 and conv.num_classes_cell() == want["num_classes_cell"] == 281
    assert conv.idx_tag_cell() == want["idx_tag_cell"]
    cfg = conv.decoder_cfg()
    assert (cfg["sos"], cfg["eos"], cfg["pad"], cfg["max_len"], cfg["max_len_cell"]) == (40, 41, 42, 500, 150)
    with pytest.raises(AssertionError):
        M.MtlTabNetConvertor(start_end_same=True)


def _like(conv):
    return {"char2idx": conv.char2idx, "char2idx_cell": conv.char2idx_cell, "end_idx": conv.end_idx, "end_idx_cell": conv.end_idx_cell,
            "padding_idx_cell": conv.padding_idx_cell}


@pytest.mark.parametrize("k,name", list(enumerate(MTL_HOST_CASES)))
def test_output_format_and_post_processor_equal_the_reference(conv, golden, k, name):
    want = golden["cases"][name]
    tag, box, cell, meta = mtl_host_case_tensors(name, _like(conv), golden["seed"] + k)
    strings, scores, bboxes, cells, cell_scores = conv.output_format(tag, box, [cell], [meta])
    assert strings[0] == want["text"]
    assert abs(scores[0] - want["score"]) <= 1e-6
    assert cells[0] == want["cell"]
    post = M.MasterPostProcessor(strict=True)
    result = dict(text=strings[0], score=scores[0], bbox=bboxes[0], cell=cells[0])
    if want.get("raises") == "IndexError":
        with pytest.raises(IndexError):
            post(result)
        lenient = M.MasterPostProcessor(strict=False)(dict(text=strings[0], score=scores[0], bbox=bboxes[0], cell=cells[0]))
        assert lenient["new_bbox"].shape == (0, 4) and len(M.two_point_to_four_point(lenient["new_bbox"])) == 0
    else:
        assert np.array_equal(np.asarray(bboxes[0]), np.asarray(want["bbox_decoded"])) or \
            np.abs(np.asarray(bboxes[0]) - np.asarray(want["bbox_decoded"])).max() <= 1e-9
        assert np.allclose(cell_scores[0], want["cell_scores"], atol=1e-6)
        pred = post(result)
        assert np.abs(np.asarray(pred["bbox"]) - np.asarray(want["bbox_kept"])).max() <= 1e-9
        assert np.array_equal(pred["new_bbox"], np.asarray(want["new_bbox"], dtype=np.int32))
        assert np.array_equal(M.two_point_to_four_point(pred["new_bbox"]), np.asarray(want["polygons"]))
        result = pred
    for key in ("pred_html", "html_context", "structure_str", "structure_str_list"):
        assert result[key] == want[key], key


def test_format_ids_equals_output_format(conv, golden):
    """the engine hands over arg-max ids + probabilities instead of logits: same result"""
    tag, box, cell, meta = mtl_host_case_tensors("spans", _like(conv), golden["seed"] + 1)
    a = conv.output_format(tag, box, [cell], [meta])
    ti, tp = M._softmax_max(tag[0])
    ci, cp = M._softmax_max(cell)
    s, sc, bb, cs, css = conv.format_ids(ti, tp, box[0], ci, cp, meta)
    assert (s, cs) == (a[0][0], a[3][0]) and sc == a[1][0] and np.array_equal(bb, a[2][0]) and css == a[4][0]
    d = M.mtl_result(conv, M.MasterPostProcessor(), ti, tp, box[0], ci, cp, meta, inputs="x.png")
    assert set(d) >= {"polygons", "structure_str_list", "structure_str", "html_context", "inputs"} and d["polygons"].shape[1] == 8


def test_image_meta_follows_table_resize():
    m = M.mtl_image_meta(211, 640, 480, 158)
    assert m["scale_factor"] == (158 / 211, 480 / 640) and m["pad_shape"] == (480, 480, 3) and m["ori_shape"] == (211, 640, 3)
