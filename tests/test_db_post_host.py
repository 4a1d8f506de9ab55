"""Product host post-process (C++ in libpdftable_hip.so) vs the oracle restatement: bit-exact boxes."""
import os

import numpy as np
import pytest

from oracle import db_post, db_pre
from pdf_table_amd import engine as E
from pdf_table_amd import lib as L


@pytest.fixture(scope="module", autouse=True)
def built():
    from pdf_table_amd.build import build
    build(verbose=False)


def pack_bits(bm):
    h, w = bm.shape
    assert w % 32 == 0
    b = bm.reshape(h, w // 32, 32).astype(np.uint32)
    return (b << np.arange(32, dtype=np.uint32)).sum(axis=2).astype(np.uint32)


def blobs(seed, h=96, w=160, n=14):
    rng = np.random.default_rng(seed)
    prob = rng.uniform(0, 0.25, (h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cx, cy = rng.uniform(5, w - 5), rng.uniform(5, h - 5)
        a, b = rng.uniform(2, 30), rng.uniform(1.5, 8)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        inside = (np.abs(u) < a) & (np.abs(v) < b)
        prob[inside] = rng.uniform(0.5, 1.0)
    # a ring and single pixels
    prob[10:22, 120:140] = 0.8
    prob[14:18, 126:134] = 0.1
    prob[50, 3] = 0.9
    return prob


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_candidates_bit_exact(seed):
    prob = blobs(seed)
    bm = prob > 0.3
    boxes, sside = E.db_candidates(pack_bits(bm), 1000, 3.0)
    ref = db_post.candidates_from_bitmap(bm, 1000, 3)
    assert len(ref) == len(boxes) and len(ref) >= 5
    for (pts, ss), b, s in zip(ref, boxes, sside):
        np.testing.assert_array_equal(pts.reshape(-1), b)
        assert np.float32(ss) == s


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_finalize_bit_exact(seed):
    prob = blobs(seed)
    bm = prob > 0.3
    boxes, _ = E.db_candidates(pack_bits(bm), 1000, 3.0)
    scores = np.array([db_post.box_score_fast(prob, b.reshape(4, 2)) for b in boxes], np.float32)
    out, osc = E.db_finalize(boxes, scores, prob.shape, (192, 320), 0.6, 1.5, 3.0)
    ref_boxes, ref_scores = db_post.boxes_from_bitmap(prob, bm, 320, 192, 0.6, 1.5, 1000, 3, scores_override=scores)
    np.testing.assert_array_equal(out.reshape(-1, 4, 2), ref_boxes.astype(np.int32))
    assert len(out) >= 1
    np.testing.assert_array_equal(osc, np.array(ref_scores, np.float32))


def test_empty_and_full_bitmaps():
    z = np.zeros((32, 64), bool)
    boxes, _ = E.db_candidates(pack_bits(z))
    assert len(boxes) == 0
    o = np.ones((32, 64), bool)
    boxes, sside = E.db_candidates(pack_bits(o))
    ref = db_post.candidates_from_bitmap(o)
    assert len(boxes) == len(ref) == 1
    np.testing.assert_array_equal(ref[0][0].reshape(-1), boxes[0])


def test_max_candidates_takes_last_found_first():
    bm = np.zeros((64, 64), bool)
    for i in range(6):
        bm[4 + 10 * i: 9 + 10 * i, 5:40] = True
    boxes, _ = E.db_candidates(pack_bits(bm), max_candidates=2)
    assert len(boxes) == 2 and boxes[0][1] > boxes[1][1] > 40   # bottom-most components


@pytest.mark.parametrize("seed", [5, 6])
def test_finalize_torch_flavour_bit_exact(seed):
    """db_net/ocr_detection_utils.py:167-210: score 0.3, unclip 1.5, int32 truncation before rescaling."""
    prob = blobs(seed)
    bm = prob > 0.2
    boxes, _ = E.db_candidates(pack_bits(bm), 1000, 3.0)
    scores = np.array([db_post.box_score_fast(prob, b.reshape(4, 2)) for b in boxes], np.float32)
    out, _ = E.db_finalize(boxes, scores, prob.shape, (211, 333), 0.3, 1.5, 3.0, L.PT_DET_POST_DB_TORCH)
    ref, _ = db_post.boxes_from_bitmap(prob, bm, 333, 211, 0.3, 1.5, 1000, 3, scores_override=scores, flavour="db")
    np.testing.assert_array_equal(out.reshape(-1, 4, 2), ref)
    assert len(out) >= 2


def test_batch_host_calls_equal_the_per_page_calls():
    """pt_db_candidates_batch / pt_db_finalize_batch (pages on threads inside the library, filter_tag_det_res in C++) give
    exactly what the per-page calls + the numpy filter_tag_det_res give, for any thread count"""
    import numpy as np
    from pdf_table_amd import engine as E
    from pdf_table_amd import lib as L
    from pdf_table_amd.det_stage import filter_tag_det_res
    rng = np.random.default_rng(5)
    n, H, W = 7, 160, 192
    bits = np.zeros((n, H, W), bool)
    for i in range(n):
        for _ in range(int(rng.integers(0, 40))):                      # random rectangles, some touching, some 1-2 px, some at the border
            y, x = int(rng.integers(0, H - 2)), int(rng.integers(0, W - 2))
            h, w = int(rng.integers(1, 14)), int(rng.integers(1, 60))
            bits[i, y:y + h, x:x + w] = True
    words = np.zeros((n, H, W // 32), np.uint32)
    for k in range(32):
        words |= bits[:, :, k::32].astype(np.uint32) << np.uint32(k)
    for flavour, post, filt in (("db_pp", L.PT_DET_POST_DB_PP, True), ("db", L.PT_DET_POST_DB_TORCH, False)):
        for nthr in (1, 3, 0):
            cand, counts = E.db_candidates_batch(words, 1000, 3.0, nthr)
            scores = np.zeros(cand.shape[:2], np.float32)
            want = []
            for i in range(n):
                c1, _ = E.db_candidates(words[i], 1000, 3.0)
                assert counts[i] == len(c1) and np.array_equal(cand[i, :counts[i]], c1)
                sc = rng.uniform(0.3, 1.0, len(c1)).astype(np.float32)
                scores[i, :len(c1)] = sc
                out, _ = E.db_finalize(c1, sc, (H, W), (400, 500), 0.6, 1.5, 3.0, post)
                want.append(filter_tag_det_res(out, 400, 500).reshape(-1, 8) if filt else out)
            got = E.db_finalize_batch(cand, scores, counts, (H, W), (400, 500), 0.6, 1.5, 3.0, post, filter_tag=filt, n_threads=nthr)
            assert sum(len(w) for w in want) > 20
            for g, w in zip(got, want):
                assert g.dtype == w.dtype and np.array_equal(g, w)
