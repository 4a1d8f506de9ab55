"""oracle/pplcnet.py and oracle/pil_resize.py against the reference's own PPLCNet module, post-processors and image
processor, and against Pillow (tests/golden/pplcnet.npz, pplcnet_post.json; generator: tests/golden/make_golden.py pplcnet)."""
import json
import os

import numpy as np
import pytest
import torch

from cls_synth import CLS_GOLDEN_TASKS, PIL_CASES, PRE_CASES, cls_inputs, u8_image
from oracle import pil_resize, pplcnet
from pdf_table_amd.synth_weights import pplcnet_state_dict


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "pplcnet.npz"))


@pytest.mark.parametrize("task", list(CLS_GOLDEN_TASKS))
def test_net_matches_reference_module(gold, task):
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    y = pplcnet.pplcnet_forward(pplcnet_state_dict(seed, cn), torch.from_numpy(cls_inputs(seed, 5, hw)), textline=textline)
    ref = gold[f"logits_{task}"]
    assert y.shape == ref.shape
    assert np.abs(y.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert np.ptp(ref, axis=0).max() > 0.05        # the five images really give different logits


def test_postprocessors_match_reference(gold, golden_dir):
    with open(os.path.join(golden_dir, "pplcnet_post.json")) as f:
        ref = json.load(f)
    for task in ("textline_orientation", "text_image_orientation"):
        assert pplcnet.topk_postprocess(gold[f"logits_{task}"], task) == ref[task]
    assert pplcnet.table_attribute_postprocess(gold["logits_table_attribute"]) == ref["table_attribute"]


def test_pillow_resize_restatement_is_bit_exact(gold):
    for i, (h, w, oh, ow) in enumerate(PIL_CASES):
        got = pil_resize.pil_resize_bilinear_u8(u8_image(100 + i, h, w), oh, ow)
        assert np.array_equal(got, gold[f"pil_out_{i}"]), (i, h, w, oh, ow)


def test_pillow_resize_live():
    """the same against the Pillow of this image, on sizes the golden does not hold"""
    Image = pytest.importorskip("PIL.Image")
    for i, (h, w, oh, ow) in enumerate([(64, 64, 80, 160), (91, 333, 80, 160), (1024, 1024, 224, 224), (224, 224, 224, 224)]):
        img = u8_image(500 + i, h, w)
        ref = np.array(Image.fromarray(img).resize((ow, oh), resample=Image.BILINEAR))
        assert np.array_equal(pil_resize.pil_resize_bilinear_u8(img, oh, ow), ref)


def test_preprocess_matches_reference_processor(gold):
    for i, (task, (h, w)) in enumerate(PRE_CASES):
        oh, ow = pplcnet.CLS_TASKS[task]["size"]
        got = pil_resize.pplcnet_preprocess(u8_image(i, h, w), oh, ow)
        assert np.abs(got - gold[f"pre_out_{i}"]).max() <= 2e-6
