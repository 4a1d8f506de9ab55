"""Oracle pre/post-process restatements: golden vectors from the reference's own numpy code
(tests/golden/db_host_numpy.npz) and hand-derived known answers for the cv2/pyclipper pieces
(parity unpinned there: see oracle/db_post.py header)."""
import math
import os

import numpy as np
import pytest

from oracle import db_post, db_pre


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "db_host_numpy.npz"))


def test_det_plan_matches_reference(g):
    for h, w, nh, nw, rh, rw in g["plan"]:
        got = db_pre.det_plan_db_pp(int(h), int(w))
        assert got == (int(nh), int(nw))
        assert got[0] / float(h) == rh and got[1] / float(w) == rw


def test_det_plan_torch_known_answers():
    # processor_ocr_dbnet.py:50-60 by hand: 1024x1024 -> short side 736, other side ceil(736/32)*32 = 736
    assert db_pre.det_plan_db_torch(1024, 1024) == (736, 736)
    assert db_pre.det_plan_db_torch(500, 1000) == (736, 1472)
    assert db_pre.det_plan_db_torch(1000, 700) == (1056, 736)   # ceil(736/700*1000/32)*32 = 33*32... -> 1056


def test_filter_tag_det_res_matches_reference(g):
    got = db_post.filter_tag_det_res(g["boxes"].copy(), tuple(g["filter_shape"]))
    np.testing.assert_array_equal(got, g["filtered"])
    assert 0 < len(got) < len(g["boxes"])


def test_sort_key_matches_reference(g):
    np.testing.assert_array_equal(db_post.sort_det_result(g["sort_in"]).astype(np.float32), g["sort_out"])


def test_resize_known_answers():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    np.testing.assert_array_equal(db_pre.cv2_resize_linear_u8(img, 96, 64), img)             # identity
    const = np.full((50, 70, 3), 137, np.uint8)
    assert (db_pre.cv2_resize_linear_u8(const, 32, 32) == 137).all()                          # constants survive
    a = db_pre.cv2_resize_linear_u8(img, 48, 32)                                              # exact 2x -> area
    ref = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    np.testing.assert_array_equal(a, ref)
    # a horizontal ramp stays a (rounded) ramp under bilinear down-scaling
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (40, 1, 3))
    r = db_pre.cv2_resize_linear_u8(ramp, 80, 40)
    expect = ((np.arange(80) + 0.5) * (100 / 80) - 0.5) * 2
    assert np.abs(r[0, 2:-2, 0].astype(float) - expect[2:-2]).max() <= 1.0


def _rect_bitmap(h, w, rects):
    bm = np.zeros((h, w), bool)
    for (x0, y0, x1, y1) in rects:
        bm[y0:y1 + 1, x0:x1 + 1] = True
    return bm


def test_find_contours_known_answers():
    bm = _rect_bitmap(32, 64, [(3, 4, 10, 8), (20, 10, 40, 20)])
    cs = db_post.find_contours(bm)
    assert len(cs) == 2
    # reverse discovery order: the lower rectangle first; CHAIN_APPROX_SIMPLE keeps the 4 corners,
    # starting at the top-left pixel and running counter-clockwise on screen (down first)
    np.testing.assert_array_equal(cs[0], [[20, 10], [20, 20], [40, 20], [40, 10]])
    np.testing.assert_array_equal(cs[1], [[3, 4], [3, 8], [10, 8], [10, 4]])
    # single pixel, and a ring (outer + hole border)
    bm = np.zeros((16, 32), bool)
    bm[5, 7] = True
    cs = db_post.find_contours(bm)
    assert len(cs) == 1 and cs[0].tolist() == [[7, 5]]
    bm = _rect_bitmap(20, 32, [(4, 4, 14, 14)])
    bm[7:12, 7:12] = False
    cs = db_post.find_contours(bm)
    assert len(cs) == 2                      # RETR_LIST returns the hole border as its own contour
    # 8-connected tracing cuts the hole's corners diagonally: an octagon on the foreground ring
    assert sorted(len(c) for c in cs) == [4, 8]
    hole = cs[0]
    assert hole[:, 0].min() == 6 and hole[:, 0].max() == 12   # hole border runs on the foreground ring


def test_min_area_rect_known_answers():
    pts = np.array([[0, 0], [10, 0], [10, 4], [0, 4]])
    (cx, cy), (w, h), ang = db_post.min_area_rect(pts)
    assert (cx, cy) == (5.0, 2.0) and sorted([w, h]) == [4.0, 10.0]
    box, sside = db_post.get_mini_boxes(pts)
    np.testing.assert_allclose(np.array(box), [[0, 0], [10, 0], [10, 4], [0, 4]], atol=1e-4)
    assert sside == 4.0
    # a 45-degree bar: min-area rectangle recovers its width/height
    c, s = math.cos(0.6), math.sin(0.6)
    q = np.array([[-20, -3], [20, -3], [20, 3], [-20, 3]], float) @ np.array([[c, s], [-s, c]]) + [50, 40]
    (_, _), (w, h), _ = db_post.min_area_rect(q)
    np.testing.assert_allclose(sorted([w, h]), [6, 40], atol=1e-3)


def test_fill_poly_and_score():
    # axis-aligned quad with integer vertices: the closed rectangle is filled
    m = db_post.fill_poly_mask(10, 12, np.array([[2, 1], [8, 1], [8, 6], [2, 6]]))
    assert m.sum() == 7 * 6 and m[1:7, 2:9].all()
    # diamond: outline is Bresenham, interior by scan lines; symmetric
    m = db_post.fill_poly_mask(11, 11, np.array([[5, 0], [10, 5], [5, 10], [0, 5]]))
    assert (m == m[::-1]).all() and (m == m[:, ::-1]).all() and m[5].all() and m[0].sum() == 1
    prob = np.zeros((20, 30), np.float32)
    prob[5:10, 5:20] = 0.5
    s = db_post.box_score_fast(prob, np.array([[5, 5], [19, 5], [19, 9], [5, 9]], np.float32))
    assert s == pytest.approx(0.5)


def test_unclip_rectangle_closed_form():
    # offsetting a w x h rectangle by d = w*h*r / (2(w+h)) grows each side by d
    w, h, r = 40.0, 10.0, 1.5
    box = np.array([[10, 20], [10 + w, 20], [10 + w, 20 + h], [10, 20 + h]], np.float32)
    d = w * h * r / (2 * (w + h))
    out = db_post.unclip(box, r)
    assert out[:, 0].min() == round(10 - d) and out[:, 0].max() == round(10 + w + d)
    assert out[:, 1].min() == round(20 - d) and out[:, 1].max() == round(20 + h + d)
    mini, sside = db_post.get_mini_boxes(out)
    assert sside == pytest.approx(h + 2 * d, abs=1.0)


def test_boxes_from_bitmap_end_to_end():
    prob = np.zeros((64, 96), np.float32)
    prob[10:20, 8:60] = 0.9
    prob[40:44, 30:33] = 0.9          # too small after the sside gate
    boxes, scores = db_post.boxes_from_bitmap(prob, prob > 0.3, 192, 128, box_thresh=0.6, unclip_ratio=1.5)
    assert boxes.shape == (1, 4, 2) and scores[0] == pytest.approx(0.9, abs=1e-6)
    # 52x10 blob -> offset d = 51*9*1.5/(2*60) = 5.7 ; rescaled x2
    assert abs(int(boxes[0, 0, 0]) - 2 * (8 - 5.7)) <= 2 and abs(int(boxes[0, 2, 0]) - 2 * (59 + 5.7)) <= 2
