"""GPU parity tests of the ConvNextViT recogniser through the C ABI.

Integer work (resized pixels -> gray is fp32 arithmetic on integer pixels: bit-exact; token ids) and float work (winning
logit within 1e-3 of the fp32 oracle / of the reference module's own output in PT_PRECISION_BF16X3; bf16 drift recorded);
token ids may differ from the oracle's only where the oracle's own top-2 margin is inside the tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import convnext_vit as ocv
from oracle.crnn import keepratio_resize
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import convnext_vit_state_dict
from pdf_table_amd.weights import pack_convnext_vit

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def sd():
    return convnext_vit_state_dict(seed=29)          # the seed of tests/golden/convnext_vit.npz


@pytest.fixture(scope="module")
def eng(sd):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(sd))
    yield e
    e.close()


def _gray(x):
    return x[:, 0] * 0.2989 + x[:, 1] * 0.5870 + x[:, 2] * 0.1140


def _compare(tag, ids, mx, logits, tol_logit, tol_margin):
    top2 = torch.topk(logits, 2, dim=-1)
    dmax = (mx - top2.values[..., 0]).abs().max().item()
    margin = top2.values[..., 0] - top2.values[..., 1]
    diff = ids != top2.indices[..., 0]
    print(f"convnext-vit {tag}: max|d max-logit| = {dmax:.2e} (scale {logits.abs().max().item():.1f}); "
          f"{int(diff.sum())} of {ids.numel()} ids differ; min margin {margin.min().item():.2e}")
    assert dmax <= tol_logit
    assert bool((margin[diff] <= tol_margin).all())
    return dmax


def test_x3_matches_the_reference_modules_own_output(eng, golden_dir):
    """the fixture the reference's ConvNextViT module produced (make_golden.py::gen_convnext_vit): three chunks of one line"""
    g = np.load(os.path.join(golden_dir, "convnext_vit.npz"))
    x = torch.from_numpy(g["img_u8"]).float().div(255.).permute(0, 3, 1, 2)
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        ids, mx = eng.rec_cvit_forward_net(_gray(x).contiguous().cuda())
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    ids, mx = ids.cpu(), mx.cpu()
    val, idx = torch.from_numpy(g["top2_val"]), torch.from_numpy(g["top2_idx"]).long()
    dmax = (mx - val[..., 0]).abs().max().item()
    margin = val[..., 0] - val[..., 1]
    diff = ids.long() != idx[..., 0]
    print(f"convnext-vit x3 vs reference module: max|d max-logit| = {dmax:.2e} (scale {float(g['logits_abs_max']):.1f}); "
          f"{int(diff.sum())} of {ids.numel()} ids differ")
    assert dmax <= TOL
    assert bool((margin[diff] <= 2 * TOL).all())


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_lines_match_fp32_oracle(eng, sd, mode):
    """five 804-px lines (ragged text widths, one empty) in the line layout: chunks are cut by the kernel"""
    rng = np.random.default_rng(17)
    n = 5
    img = rng.uniform(0, 1, (n, 3, 32, 804)).astype(np.float32)
    img[1, :, :, 500:] = 0
    img[2, :, :, 120:] = 0
    img[4] = 0
    xt = torch.from_numpy(img)
    chunks = torch.stack([xt[:, :, :, 252 * j:252 * j + 300] for j in range(3)], 1).reshape(3 * n, 3, 32, 300)
    with torch.no_grad():
        logits = ocv.convnext_vit_forward_fp32(sd, chunks)
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        ids, mx = eng.rec_cvit_forward_net(_gray(xt).contiguous().cuda())
        ids_c, mx_c = eng.rec_cvit_forward_net(_gray(chunks).contiguous().cuda())
        ids_1, mx_1 = eng.rec_cvit_forward_net(_gray(xt[1:2]).contiguous().cuda())
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    assert torch.equal(ids, ids_c) and torch.equal(mx, mx_c), "line layout and chunk layout are the same computation"
    assert torch.equal(ids[1:2], ids_1) and torch.equal(mx[1:2], mx_1), "a line's result does not depend on its batch"
    scale = logits.abs().max().item()
    if mode == "bf16x3":
        _compare("x3", ids.cpu().long(), mx.cpu(), logits, TOL, 2 * TOL)
    else:
        _compare("bf16", ids.cpu().long(), mx.cpu(), logits, 0.06 * scale, 0.12 * scale)


def _crops():
    rng = np.random.default_rng(23)
    sizes = [(32, 804), (64, 1608), (40, 700), (25, 90), (48, 2000), (33, 60)]
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]


def test_preprocess_bit_exact(eng):
    crops = _crops()
    gray = eng.rec_cvit_preprocess_crops(crops).cpu()
    for i, c in enumerate(crops):
        full = torch.from_numpy(keepratio_resize(c, 32, 804)).float().div(255.).permute(2, 0, 1)[None]
        want = _gray(full)[0]
        assert torch.equal(gray[i], want), f"crop {i}: {(gray[i] - want).abs().max()}"
        d = ocv.chunk_preprocess(c)                                   # and the chunks the reference's processor makes
        for j in range(3):
            assert torch.equal(gray[i][:, 252 * j:252 * j + 300], _gray(d[j:j + 1])[0])


def test_crops_end_to_end_x3(eng, sd):
    crops = _crops()
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        ids, mx = eng.rec_cvit_forward_crops(crops)
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    with torch.no_grad():
        logits = torch.cat([ocv.convnext_vit_forward_fp32(sd, ocv.chunk_preprocess(c)) for c in crops], 0)
    _compare("crops x3", ids.cpu().long(), mx.cpu(), logits, TOL, 2 * TOL)


def test_missing_weights_fail_loudly():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    try:
        with pytest.raises(L.PtError, match="ConvNextViT weights not loaded"):
            e.rec_cvit_forward_net(torch.zeros((3, 32, 300), dtype=torch.float32, device="cuda"))
    finally:
        e.close()


def test_recognition_task_serves_convnext_vit(tmp_path, sd):
    """OcrRecognitionTask(model="ConvNextViT"): from a checkpoint directory (pytorch_model.pt in the transformers-5 key names
    under the `recognizer.` prefix + vocab.txt, modeling_ocr_recognition.py:100-132) and from the seeded checkpoint: same
    texts, and the texts of the oracle's greedy decode of the device ids (vocabulary from class 2)"""
    from pdf_table_amd.ocr_recognition_task import OcrRecognitionTask
    from pdf_table_amd.rec_stage import ctc_collapse
    v5 = {}
    for k, v in sd.items():
        k = k.replace("vit.encoder.layer.", "vit.layers.").replace(".attention.attention.query.", ".attention.q_proj.")
        k = k.replace(".attention.attention.key.", ".attention.k_proj.").replace(".attention.attention.value.", ".attention.v_proj.")
        k = k.replace(".attention.output.dense.", ".attention.o_proj.").replace(".intermediate.dense.", ".mlp.fc1.")
        if ".vit.layers." in k:
            k = k.replace(".output.dense.", ".mlp.fc2.")
        v5["recognizer." + k] = v
    torch.save(v5, tmp_path / "pytorch_model.pt")
    vocab = [chr(0x4E00 + i) for i in range(L.PT_CVIT_NCLS - 2)]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    task = OcrRecognitionTask(model="ConvNextViT", task_path=str(tmp_path))
    crops = _crops()
    texts = task(crops)
    assert isinstance(texts, list) and len(texts) == len(crops) and all(isinstance(t, str) for t in texts)
    ids, _ = task._engine.rec_cvit_forward_crops(crops)
    want = ["".join(vocab[t - 2] for t in row if t >= 2) for row in ctc_collapse(ids.cpu().numpy())]
    assert texts == want and any(len(t) > 3 for t in texts)
    assert texts[0] == task(crops[0])[0]
    seeded = OcrRecognitionTask(model="ConvNextViT", synthetic_seed=29, engine=task._engine)
    assert seeded(crops) == texts


def test_pipeline_with_convnext_vit_recogniser():
    """OcrTablePipeline(recognizer="ConvNextViT"): detector boxes -> device crops -> chunked recogniser, one text per box;
    the texts equal the task's on the same boxes cropped by the oracle's crop_image"""
    from oracle import crnn as ocrnn
    from pdf_table_amd.pipeline import OcrTablePipeline
    from pdf_table_amd.synth_pages import make_page
    pipe = OcrTablePipeline(device=0, recognizer="ConvNextViT", synthetic_seed=0)
    pages = [make_page(i)[0] for i in range(2)]
    res = pipe.predict(pages)
    assert len(res) == 2
    for page, r in zip(pages, res):
        assert len(r.ocr_result) == len(r.det_result) and len(r.det_result) > 0
        k = min(6, len(r.det_result))
        crops = [ocrnn.crop_image(page, ocrnn.order_point(np.asarray(b, np.float64))) for b in r.det_result[:k]]
        assert [o["text"] for o in r.ocr_result[:k]] == pipe.text_recognizer(crops)


def test_fused_mlp_agrees_with_the_two_gemm_path(tmp_path):
    """bf16 mode: the fused MLP kernel (hidden layer never in HBM, cvit_mlp_kernel) against the two GEMMs through the hidden
    tensor (PT_CVIT_FUSED_MLP=0 in a child process: the switch is read once).  Same bf16 operands and the same bf16 rounding of
    the hidden values; the sum over the hidden units runs in another order and GELU's erf is the 1.5e-7 approximation, so
    the winning logits agree to fp32-accumulation noise -- far inside bf16's own drift -- and the ids wherever the margin is
    above that noise."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_weights import convnext_vit_state_dict
from pdf_table_amd.weights import pack_convnext_vit
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(convnext_vit_state_dict(seed=29), x3=False))
rng = np.random.default_rng(5)
g = rng.uniform(0, 1, (7, 32, 804)).astype(np.float32)
g[2, :, 300:] = 0
ids, mx = eng.rec_cvit_forward_net(torch.from_numpy(g).cuda())
np.savez(sys.argv[1], ids=ids.cpu().numpy(), mx=mx.cpu().numpy())
'''
    outs = []
    for tag, env in (("fused", {}), ("gemms", {"PT_CVIT_FUSED_MLP": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=e, timeout=300)
        outs.append(np.load(out))
    d = np.abs(outs[0]["mx"] - outs[1]["mx"]).max()
    same = (outs[0]["ids"] == outs[1]["ids"]).mean()
    print(f"convnext-vit fused vs two-GEMM MLP: max|d max-logit| = {d:.2e}, {100 * same:.2f} % of the ids equal")
    assert d <= 0.1 and same >= 0.95          # bf16-class agreement (logit scale ~ 11); the x3 tests carry the 1e-3 contract
    assert len(np.unique(outs[0]["ids"])) > 3


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_sharing_the_all_padding_chunks_is_bit_identical(eng, mode):
    """with the text widths known, chunks without text are not computed per line (one all-padding chunk per batch is, and the
    stitching gathers it): ids AND winning logits equal the computation of every chunk bit for bit -- widths on both sides of
    the 252 / 504 chunk borders, an empty line, a full line"""
    rng = np.random.default_rng(31)
    tw = [804, 0, 1, 100, 252, 253, 300, 504, 505, 640, 30, 252, 804, 17]
    g = np.zeros((len(tw), 32, 804), np.float32)
    for i, w in enumerate(tw):
        g[i, :, :w] = rng.uniform(0, 1, (32, w))
    gt = torch.from_numpy(g).cuda()
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        ids_f, mx_f = eng.rec_cvit_forward_net(gt)
        ids_s, mx_s = eng.rec_cvit_forward_net(gt, text_w=tw)
        ids_1, mx_1 = eng.rec_cvit_forward_net(gt[1:3], text_w=tw[1:3])          # a batch with no text at all + one column
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    assert torch.equal(ids_f, ids_s) and torch.equal(mx_f, mx_s)
    assert torch.equal(ids_f[1:3], ids_1) and torch.equal(mx_f[1:3], mx_1)


def test_crops_path_shares_chunks_and_matches_the_net_entry(eng):
    """pt_rec_cvit_forward_crops derives the text widths from the crop sizes (the resize kernel's own formula): same ids and
    logits as pre-processing + the net entry computing every chunk"""
    crops = _crops()
    ids_c, mx_c = eng.rec_cvit_forward_crops(crops)
    gray = eng.rec_cvit_preprocess_crops(crops)
    ids_n, mx_n = eng.rec_cvit_forward_net(gray)
    torch.cuda.synchronize()
    assert torch.equal(ids_c, ids_n) and torch.equal(mx_c, mx_n)
    g = gray.cpu()
    for i, c in enumerate(crops):          # the widths the sharing relies on: nothing but zeros right of the resized text
        ratio = c.shape[1] / float(c.shape[0])
        nw = 804 if ratio > 804 / 32 else int(32 * ratio)
        assert bool((g[i, :, nw:] == 0).all()) and (nw == 0 or bool((g[i, :, :nw] != 0).any()))
