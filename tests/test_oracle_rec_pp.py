"""oracle/rec_pp.py against the reference's own PPOcrRecPreProcessor (tests/golden/rec_pp.npz, generator:
tests/golden/make_golden.py rec_pp), and the host-side plan of the HIP path against the oracle's batching."""
import os

import numpy as np

from oracle import rec_pp
from rec_synth import rec_pp_crops


def test_rec_pp_oracle_equals_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, "rec_pp.npz"))
    crops = rec_pp_crops(int(gold["seed"]))
    batches = rec_pp.rec_pp_preprocess(crops)
    assert len(batches) == int(gold["n_batches"])
    assert np.array_equal(batches[0]["indices"], gold["indices"])
    for b, d in enumerate(batches):
        assert d["batch_beg_img_no"] == int(gold[f"beg{b}"])
        assert d["image"].dtype == np.float32 and np.array_equal(d["image"], gold[f"batch{b}"])     # bit-exact
    for i in (0, 4, 5, 6):
        one = rec_pp.rec_pp_preprocess([crops[i]])
        assert len(one) == 1 and np.array_equal(one[0]["image"], gold[f"single{i}"])


def test_resize_norm_img_known_answers():
    """width rules of resize_norm_img (processor_ocr_rec_pp.py:43-67): imgW = int(48 * max(ratio_max, 320/48)) in [16, 1280];
    resized_w = min(imgW, max(ceil(48 * w / h), 16)); right of resized_w the image is 0, not (0 - 0.5) / 0.5"""
    white = np.full((24, 48, 3), 255, np.uint8)                          # ratio 2 -> resized_w 96, padded to 320
    out = rec_pp.resize_norm_img(white, 2.0)
    assert out.shape == (3, 48, 320) and np.all(out[:, :, :96] == 1.0) and np.all(out[:, :, 96:] == 0.0)
    out = rec_pp.resize_norm_img(np.zeros((10, 400, 3), np.uint8), 40.0)  # ratio 40 -> 1920 -> clamped to 1280
    assert out.shape == (3, 48, 1280) and np.all(out == -1.0)
    out = rec_pp.resize_norm_img(np.zeros((100, 2, 3), np.uint8), 0.02)   # ratio 0.02 -> ceil(0.96) = 1 -> min width 16
    assert out.shape == (3, 48, 320) and np.all(out[:, :, :16] == -1.0) and np.all(out[:, :, 16:] == 0.0)


def test_rec_pp_plan_matches_oracle_batching():
    from pdf_table_amd.rec_pp_stage import rec_pp_plan
    crops = rec_pp_crops(3)
    cw = np.array([c.shape[1] for c in crops])
    ch = np.array([c.shape[0] for c in crops])
    items, batches, total = rec_pp_plan(cw, ch)
    ref = rec_pp.rec_pp_preprocess(crops)
    assert len(batches) == len(ref)
    o = 0
    for (beg, n, img_w, off), d in zip(batches, ref):
        assert beg == d["batch_beg_img_no"] and (n, 3, 48, img_w) == d["image"].shape and off == o
        o += d["image"].size
    assert total == o
    assert np.array_equal(items["line"], ref[0]["indices"])
    for k, it in enumerate(items):
        h, w = crops[it["line"]].shape[:2]
        assert it["resized_w"] == min(it["img_w"], max(int(np.ceil(48 * (w / float(h)))), 16))


def test_rec_pp_plan_static_width():
    """a static export ([1, 3, 48, 320]) runs PaddleOCR's static-shape setting of the same config fields (rec_image_shape = the graph's,
    limited_max_width = its width; ocr_recognition_task._construct_pp): every mini-batch is 320 wide, a line whose ratio exceeds 320 / 48 is
    resized to the full width (resize_norm_img, processor_ocr_rec_pp.py:43-58), a narrower one keeps its ratio and is padded"""
    from pdf_table_amd.rec_pp_stage import PPOcrRecConfig, rec_pp_plan
    cfg = PPOcrRecConfig(rec_image_shape="3, 48, 320", limited_max_width=320)
    cw, ch = np.array([300, 100, 600, 31, 2000]), np.array([24, 24, 20, 30, 25])
    items, batches, total = rec_pp_plan(cw, ch, cfg)
    assert all(b[2] == 320 for b in batches) and total == len(cw) * 3 * 48 * 320
    for it in items:
        w, h = int(cw[it["line"]]), int(ch[it["line"]])
        assert it["img_w"] == 320 and it["resized_w"] == min(320, max(int(np.ceil(48 * w / float(h))), 16))
