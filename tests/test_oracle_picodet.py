"""Pins the PicoDet oracle (oracle/picodet.py) against outputs of the reference's own LCNet / CSPPAN / PicoHead modules
and OCRPicodetPostProcessor (tests/golden/picodet.npz)."""
import os
import sys

import numpy as np
import torch

from oracle import picodet as op
from pdf_table_amd.synth_weights import picodet_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_picodet_net_equals_reference_modules():
    gold = np.load(os.path.join(HERE, "golden", "picodet.npz"))
    sd = picodet_state_dict(int(gold["seed"]), num_classes=5)
    for tag in ("a", "b"):
        x = torch.from_numpy(gold[f"x_{tag}"])
        with torch.no_grad():
            f = op.lcnet_forward(sd, x)
            n = op.csppan_forward(sd, f)
            sc, bx = op.picohead_forward(sd, n, 5)
        assert np.allclose(f[-1].numpy(), gold[f"c5_{tag}"], atol=1e-5)
        assert np.allclose(n[0].numpy()[:, ::4], gold[f"p3_{tag}"], atol=1e-5)
        for i in range(4):
            assert sc[i].shape == gold[f"score{i}_{tag}"].shape
            assert np.allclose(sc[i].numpy(), gold[f"score{i}_{tag}"], atol=1e-6)
            assert np.allclose(bx[i].numpy(), gold[f"box{i}_{tag}"], atol=2e-5)


def test_picodet_postprocess_equals_reference():
    from lore_synth import synth_pico_heads
    gold = np.load(os.path.join(HERE, "golden", "picodet.npz"))
    labels = op.LABELS["en"]
    for tag in ("p", "q"):
        seed, th, tw, oh, ow = [int(v) for v in gold[f"post_case_{tag}"]]
        sc, bx = synth_pico_heads(seed, (th, tw))
        res = op.picodet_postprocess(sc, bx, [oh, ow], [float(th) / oh, float(tw) / ow], [th, tw], labels)
        assert len(res) == len(gold[f"post_cls_{tag}"]) > 10
        assert np.array_equal(np.array([r["category_id"] for r in res]), gold[f"post_cls_{tag}"])
        assert np.array_equal(np.array([r["bbox"] for r in res], dtype=np.float64), gold[f"post_bbox_{tag}"])
        assert np.array_equal(np.array([r["score"] for r in res], dtype=np.float64), gold[f"post_score_{tag}"])


def test_hard_nms_known_answer():
    b = np.array([[0, 0, 10, 10, 0.9], [1, 1, 11, 11, 0.8], [20, 20, 30, 30, 0.7], [0, 0, 10, 10.5, 0.95]], dtype=np.float64)
    kept = op.hard_nms(b, 0.5, top_k=100)
    assert kept[:, 4].tolist() == [0.95, 0.7]        # the 0.9 and 0.8 boxes overlap the best one by more than 0.5


def test_preprocess_shapes_and_scale():
    img = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3)).astype(np.uint8)
    x, sf = op.picodet_preprocess(img)
    assert x.shape == (3, 800, 608) and x.dtype == np.float32
    assert sf == [0.78125, 0.59375]                    # SURVEY.md section 8a stage 1
