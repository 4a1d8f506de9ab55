"""GPU parity tests of the PP-LCNet image classifiers through the C ABI.

Integer work (Pillow's bilinear resize) bit-exact; network logits in PT_PRECISION_BF16X3 within 1e-3 of the oracle's fp32
restatement and of the reference module's own outputs (tests/golden/pplcnet.npz); post-processing identical."""
import json
import os

import numpy as np
import pytest
import torch

from cls_synth import CLS_GOLDEN_TASKS, PRE_CASES, cls_inputs, u8_image
from oracle import pil_resize, pplcnet
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import pplcnet_state_dict
from pdf_table_amd.weights import pack_pplcnet
from test_gpu_det import _x4

pytestmark = pytest.mark.gpu
TOL_LOGIT = 1e-3          # north_star: "within 1e-3 on float logits"


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    for slot, (task, (cn, _, _, seed)) in enumerate(CLS_GOLDEN_TASKS.items()):
        e.load_weights(L.PT_MODEL_PPLCNET + slot, pack_pplcnet(pplcnet_state_dict(seed, cn)))
    yield e
    e.close()


@pytest.fixture()
def eng_x3(eng):
    eng.set_precision(L.PT_PRECISION_BF16X3)
    yield eng
    eng.set_precision(L.PT_PRECISION_BF16)


def _bf16_to_f32(t, split):
    """network input tensor [n,H,W,4|8] -> f32 [n,3,H,W]"""
    x = t.float().cpu()
    v = x[..., :3] + (x[..., 4:7] if split else 0)
    return v.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("case", range(len(PRE_CASES)))
def test_preprocess_resize_bit_exact(eng_x3, case):
    """hi + lo of the BF16X3 input reproduce the oracle's fp32 pixel values exactly where fp32 = hi + lo holds (all but
    ~1e-7 relative), i.e. the resized 8-bit image underneath is Pillow's bit for bit"""
    task, (h, w) = PRE_CASES[case]
    oh, ow = pplcnet.CLS_TASKS[task]["size"]
    img = u8_image(case, h, w)
    got = _bf16_to_f32(eng_x3.cls_preprocess([img], (oh, ow)), True)[0].numpy()
    ref = pil_resize.pplcnet_preprocess(img, oh, ow)
    # invert the normalisation: the 8-bit values must be identical
    mean = np.array(pil_resize.IMAGENET_MEAN, np.float32)[:, None, None]
    std = np.array(pil_resize.IMAGENET_STD, np.float32)[:, None, None]
    assert np.array_equal(np.rint((got * std + mean) * 255).astype(np.int32), np.rint((ref * std + mean) * 255).astype(np.int32))
    assert np.abs(got - ref).max() <= 2e-5


def test_preprocess_ragged_batch_and_identity_pass(eng_x3):
    """images of different sizes in one launch, including one that already has the output size (both passes skipped) and
    one that only needs the vertical pass"""
    imgs = [u8_image(40, 33, 300), u8_image(41, 80, 160), u8_image(42, 200, 160), u8_image(43, 80, 77), u8_image(44, 7, 9)]
    got = _bf16_to_f32(eng_x3.cls_preprocess(imgs, (80, 160)), True).numpy()
    for k, im in enumerate(imgs):
        assert np.abs(got[k] - pil_resize.pplcnet_preprocess(im, 80, 160)).max() <= 2e-5, k


@pytest.mark.parametrize("slot_task", list(enumerate(CLS_GOLDEN_TASKS)))
def test_net_x3_matches_oracle_and_reference_golden(eng_x3, golden_dir, slot_task):
    slot, task = slot_task
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    x = torch.from_numpy(cls_inputs(seed, 5, hw))
    got = eng_x3.cls_forward_net(_x4(x, split=True).cuda(), slot=slot, textline=textline).cpu().numpy()
    ref = pplcnet.pplcnet_forward(pplcnet_state_dict(seed, cn), x, textline=textline).numpy()
    gold = np.load(os.path.join(golden_dir, "pplcnet.npz"))[f"logits_{task}"]
    d_or, d_gold = np.abs(got - ref).max(), np.abs(got - gold).max()
    print(f"PP-LCNet x3 {task}: max|dlogit| vs oracle {d_or:.2e}, vs reference golden {d_gold:.2e} (scale {np.abs(gold).max():.2f})")
    assert got.shape == (5, cn)
    assert d_or <= TOL_LOGIT * max(1.0, np.abs(ref).max()) and d_gold <= TOL_LOGIT * max(1.0, np.abs(gold).max())


def test_net_bf16_drift_and_batch_tail(eng):
    """throughput mode: bounded drift; 37 images (not a multiple of the 32-row GEMM tile of the pooled head)"""
    task = "textline_orientation"
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    x = torch.from_numpy(cls_inputs(seed + 50, 37, hw)).to(torch.bfloat16).float()
    got = eng.cls_forward_net(_x4(x).cuda(), slot=0, textline=textline).cpu().numpy()
    ref = pplcnet.pplcnet_forward(pplcnet_state_dict(seed, cn), x, textline=textline).numpy()
    d = np.abs(got - ref)
    print(f"PP-LCNet bf16: max|dlogit|={d.max():.3e} mean {d.mean():.3e} (scale {np.abs(ref).max():.2f})")
    assert d.max() <= 0.05 * max(1.0, np.abs(ref).max())


def test_end_to_end_images_and_postprocess(eng_x3, golden_dir):
    """uint8 images -> resize -> net -> Topk, against the oracle chain"""
    from pdf_table_amd.cls_stage import ClsStage
    task = "text_image_orientation"
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    imgs = [u8_image(60 + k, 200 + 37 * k, 150 + 61 * k) for k in range(4)]
    for k in range(4):       # structure so that the images differ after pooling
        imgs[k][: 40 * (k + 1)] //= (k + 2)
    stage = ClsStage(eng_x3, task, slot=1)
    res = stage.images(imgs)
    x = torch.from_numpy(np.stack([pil_resize.pplcnet_preprocess(im, *hw) for im in imgs]))
    ref = pplcnet.topk_postprocess(pplcnet.pplcnet_forward(pplcnet_state_dict(seed, cn), x).numpy(), task)
    assert len(res) == 4
    for a, b in zip(res, ref):
        assert a["class_ids"] == b["class_ids"] and a["label_names"] == b["label_names"]
        assert np.abs(np.array(a["scores"]) - np.array(b["scores"])).max() <= 1e-3
    # the post-processors themselves reproduce the reference's outputs on the reference's logits
    from pdf_table_amd.cls_stage import table_attribute_postprocess, topk_postprocess
    gold = np.load(os.path.join(golden_dir, "pplcnet.npz"))
    with open(os.path.join(golden_dir, "pplcnet_post.json")) as f:
        post = json.load(f)
    assert topk_postprocess(gold["logits_textline_orientation"], "textline_orientation") == post["textline_orientation"]
    assert topk_postprocess(gold["logits_text_image_orientation"], "text_image_orientation") == post["text_image_orientation"]
    assert table_attribute_postprocess(gold["logits_table_attribute"]) == post["table_attribute"]


def test_text_lines_from_resident_pages(eng_x3):
    """pt_cls_forward_lines (warp the line crops out of the resident pages, resize, classify) == classifying the crops the
    recognition stage's oracle cuts, one at a time like OcrSystemTask.text_line_orientation"""
    from oracle import crnn as rec_oracle
    from pdf_table_amd.rec_stage import build_lines
    from pdf_table_amd.synth_pages import make_page
    pages, quads = [], []
    for i in (5, 6):
        img, gt = make_page(i, 512)
        pages.append(img)
        l = np.asarray(gt["lines"], dtype=np.float64)[:12]          # x0, y0, x1, y1 -> TL, TR, BR, BL quads
        quads.append(np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1))
    lines = build_lines(quads)
    task = "textline_orientation"
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    dev_pages = torch.from_numpy(np.stack(pages)).cuda()
    got = eng_x3.cls_forward_lines(dev_pages, lines, hw, slot=0, textline=True).cpu().numpy()
    sd = pplcnet_state_dict(seed, cn)
    ref = []
    for pi, qs in enumerate(quads):
        for q in qs:
            crop = rec_oracle.crop_image(pages[pi], rec_oracle.order_point(q))
            ref.append(pplcnet.pplcnet_forward(sd, torch.from_numpy(pil_resize.pplcnet_preprocess(crop, *hw))[None], textline=True)[0].numpy())
    ref = np.stack(ref)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL_LOGIT * max(1.0, np.abs(ref).max())


def test_task_contract():
    from pdf_table_amd.cls_image_pulc_task import ClsImagePulcTask
    t = ClsImagePulcTask(task_type="textline_orientation", synthetic_seed=21)
    one = t(u8_image(70, 40, 260))
    assert set(one) == {"class_ids", "scores", "label_names"} and one["label_names"][0] in ("0_degree", "180_degree")
    many = t([u8_image(71, 30, 200), u8_image(72, 50, 120)])
    assert isinstance(many, list) and len(many) == 2
    with pytest.raises(RuntimeError):
        ClsImagePulcTask(model="ResNet", synthetic_seed=0)
    with pytest.raises(KeyError):
        ClsImagePulcTask(task_type="no_such_task", synthetic_seed=0)
