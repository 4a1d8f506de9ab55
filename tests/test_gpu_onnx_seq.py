"""Sequence and multi-output graphs on the generic ONNX executor (pdf_table_amd/onnx_exec.py): the operator set of an
SVTR-type recogniser -- the PP-OCRv4 recogniser ``fix_model_names()`` selects for every language (model/ocr_pdf/
configuration_ocr_document.py:138-141) -- and the output convention of the PicoDet export the layout stage runs (ocr_layout_task.py:159-175).
The real files are not available offline; the graphs come from PyTorch's own exporter (tools/onnx_export.py: SvtrTiny, PicoLike), which
decomposes LayerNorm / GELU / flatten into ReduceMean / Erf / Shape-Slice-Concat chains like paddle2onnx does.  Checker: the PyTorch module in
fp32.  Arithmetic is bf16 with fp32 accumulation."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_svtr_tiny_recogniser(eng, act):
    """conv stem -> flatten(2).transpose -> 2 x [LayerNorm, fused-qkv attention, LayerNorm, MLP] -> LayerNorm -> Linear -> Softmax"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.SvtrTiny(act=act), 11)
    x = torch.randn(3, 3, 32, 64, generator=torch.Generator().manual_seed(5))
    ex = HipGraphExecutor(X.torch_export(m, x), engine=eng)
    kinds = [l.op if l.op != "act" else "act:" + l.attrs["kind"] for l in ex.layers]
    assert kinds.count("layernorm") == 5 and kinds.count("matmul") == 4 and ("act:gelu" if act == "gelu" else "act:swish") in kinds
    got = ex.run(x.numpy())[0]
    with torch.no_grad():
        want = m(x).numpy()
    assert got.shape == want.shape == (3, 8 * 16, 97)
    d = float(np.abs(got - want).max())
    same = float((got.argmax(-1) == want.argmax(-1)).mean())
    print(f"SvtrTiny[{act}]: max|d prob| = {d:.3e}, arg-max equal on {same * 100:.1f} % of {got.shape[0] * got.shape[1]} tokens ({len(ex.layers)} layers)")
    assert d <= 3e-2 and same >= 0.97
    assert np.abs(got.sum(-1) - 1.0).max() <= 2e-2
    again = ex.run(x.numpy())[0]                      # operands are cached after the first run
    assert np.array_equal(got, again)


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_svtr_tiny_recogniser_f16(act):
    """the same SVTR-type graph with precision="f16" (PT_PRECISION_F16; the executor makes its own engine in that precision): the sequence operators of
    graph_ops.hip -- LayerNorm, fused-qkv attention, GELU / swish, Softmax -- in their IEEE-half instantiation, bounds 5x tighter than the bf16 case's"""
    import onnx_export as X
    from pdf_table_amd import lib as L
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.SvtrTiny(act=act), 11)
    x = torch.randn(3, 3, 32, 64, generator=torch.Generator().manual_seed(5))
    ex = HipGraphExecutor(X.torch_export(m, x), precision="f16")
    try:
        assert ex.eng.precision == L.PT_PRECISION_F16 and ex.adt == torch.float16
        got = ex.run(x.numpy())[0]
        with torch.no_grad():
            want = m(x).numpy()
        d = float(np.abs(got - want).max())
        same = float((got.argmax(-1) == want.argmax(-1)).mean())
        print(f"SvtrTiny[{act}] f16: max|d prob| = {d:.3e}, arg-max equal on {same * 100:.1f} % of {got.shape[0] * got.shape[1]} tokens")
        assert d <= 6e-3 and same >= 0.99
        a = ex.run_device_graphed(torch.from_numpy(x.numpy()).permute(0, 2, 3, 1).contiguous().to(torch.float16).cuda(), 3)      # eager
        b = ex.run_device_graphed(torch.from_numpy(x.numpy()).permute(0, 2, 3, 1).contiguous().to(torch.float16).cuda(), 3)      # captured
        assert a[0].t.dtype == torch.float16 or a[0].t.dtype == torch.float32
        assert torch.equal(ex.values(a[0]), ex.values(b[0]))
    finally:
        ex.eng.close()


def test_dynamic_batch_export(eng):
    """an export with a symbolic batch axis (the shipped PP-OCR files): the Reshape targets come out of Shape -> Gather -> Concat arithmetic the
    executor folds on the host per input shape -- the same file runs batches of 2 and of 5, tolerance mode within 1e-3 of the fp32 module"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.SvtrTiny(act="gelu"), 17)
    ex = HipGraphExecutor(X.torch_export(m, torch.zeros(2, 3, 32, 64), dynamic_batch=True), engine=eng, precision="bf16x3")
    assert not isinstance(ex.inputs[0].shape[0], int)
    g = torch.Generator().manual_seed(3)
    for n in (2, 5):
        x = torch.randn(n, 3, 32, 64, generator=g)
        got = ex.run(x.numpy())[0]
        with torch.no_grad():
            want = m(x).numpy()
        assert got.shape == want.shape == (n, 8 * 16, 97)
        assert float(np.abs(got - want).max()) <= 1e-3


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_graph_replay_equals_the_eager_walk(eng, precision):
    """run_device_graphed(): eager on the first call for a shape, captured on the second, replayed afterwards -- the same launches, so the same bits
    as run_device() for every input; a second input shape gets its own graph"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.SvtrTiny(act="gelu"), 13)
    ex = HipGraphExecutor(X.torch_export(m, torch.zeros(1, 3, 32, 64)), engine=eng, precision=precision)
    g = torch.Generator().manual_seed(9)
    dev = torch.device("cuda", 0)
    dt = torch.float32 if precision == "bf16x3" else torch.bfloat16
    xs = [torch.randn(1, 32, 64, 3, generator=g).to(dt).to(dev) for _ in range(5)]
    want = [ex.values(ex.run_device(x, 3)[0]).clone() for x in xs]
    got = [ex.values(ex.run_device_graphed(x, 3)[0]).clone() for x in xs]          # call 1 eager, call 2 captures, calls 3.. replay
    assert len(ex._graphs) == 1 and not isinstance(next(iter(ex._graphs.values())), str)
    for w, o in zip(want, got):
        assert torch.equal(w, o)
    # a mini-batch of single-image walks in ONE graph (what OcrRecognitionTask does with a static batch-1 export)
    xb = torch.cat(xs[:3])
    for _ in range(3):                                            # eager, capture, replay
        outs = ex.run_lines_graphed(xb, 3)
        for w, (a,) in zip(want[:3], outs):
            assert torch.equal(w, ex.values(a))


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_svtr_tiny_recogniser_tolerance_mode(eng, act):
    """the SVTR-type recogniser in the executor's tolerance mode (precision="bf16x3"): LayerNorm, fused-qkv attention, GELU / swish, Softmax on
    (hi | lo) token rows, three-pass GEMMs -- probabilities within 1e-3 of the fp32 module, arg-max identical outside its own <= 2e-3 ties"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.SvtrTiny(act=act), 11)
    x = torch.randn(3, 3, 32, 64, generator=torch.Generator().manual_seed(5))
    ex = HipGraphExecutor(X.torch_export(m, x), engine=eng, precision="bf16x3")
    got = ex.run(x.numpy())[0]
    with torch.no_grad():
        want = m(x).numpy()
    d = float(np.abs(got - want).max())
    top2 = np.sort(want, -1)[..., -2:]
    tie = (top2[..., 1] - top2[..., 0]) <= 2e-3
    diff = got.argmax(-1) != want.argmax(-1)
    print(f"SvtrTiny[{act}] bf16x3: max|d prob| = {d:.3e}; arg-max differs on {int(diff.sum())} tokens ({int((diff & ~tie).sum())} off a tie)")
    assert d <= 1e-3 and not (diff & ~tie).any()
    assert np.abs(got.sum(-1) - 1.0).max() <= 1e-4


def test_picodet_shaped_outputs_tolerance_mode(eng):
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.PicoLike(), 12)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(6))
    ex = HipGraphExecutor(X.torch_export(m, x), engine=eng, precision="bf16x3")
    got = ex.run(x.numpy())
    with torch.no_grad():
        want = [w.numpy() for w in m(x)]
    worst = 0.0
    for g, w in zip(got, want):
        assert g.shape == w.shape
        worst = max(worst, float(np.abs(g - w).max()) / max(1.0, float(np.abs(w).max())))
    print(f"PicoLike bf16x3: worst max|d| / scale over six outputs = {worst:.3e}")
    assert worst <= 1e-3


def test_picodet_shaped_outputs(eng):
    """three levels, per level scores = sigmoid(conv[:, :ncls]) and distributions = conv[:, ncls:], each flatten(2).permute(0, 2, 1): six outputs,
    first half scores, second half distributions -- what OcrLayoutTask.get_onnx_output_dict splits (ocr_layout_task.py:159-175)"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    m = X.seeded(X.PicoLike(), 12)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(6))
    ex = HipGraphExecutor(X.torch_export(m, x), engine=eng)
    got = ex.run(x.numpy())
    with torch.no_grad():
        want = [w.numpy() for w in m(x)]
    assert len(got) == 6
    for g, w in zip(got, want):
        assert g.shape == w.shape
        scale = max(1.0, float(np.abs(w).max()))
        assert float(np.abs(g - w).max()) <= 4e-2 * scale, (g.shape, float(np.abs(g - w).max()), scale)
    print("PicoLike:", [tuple(g.shape) for g in got], "max|d| =", max(float(np.abs(g - w).max()) for g, w in zip(got, want)))


def test_unsupported_sequence_patterns_fail_loudly(eng):
    """a MatMul of two computed tensors that is not the fused-qkv attention names itself"""
    import onnx_export as X
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    from pdf_table_amd.onnx_import import UnsupportedOnnxGraph

    class Gram(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(3, 64, 3, 2, 1)

        def forward(self, x):
            t = self.c(x).flatten(2).transpose(1, 2)
            return t @ t.transpose(1, 2)

    x = torch.randn(1, 3, 16, 16)
    ex = HipGraphExecutor(X.torch_export(X.seeded(Gram(), 3), x), engine=eng)
    with pytest.raises(UnsupportedOnnxGraph, match="fused-qkv attention"):
        ex.run(x.numpy())


def test_layout_task_onnx_door(eng, tmp_path):
    """OcrLayoutTask(task_path=<model.onnx>): pre-processing kernel -> generic executor -> get_onnx_output_dict split -> the PicoDet host
    post-processor, against the exported PyTorch module's own outputs through the oracle's post-processor (pinned to the reference's
    OCRPicodetPostProcessor) on the oracle's pre-processing of the same page"""
    import onnx_export as X
    from oracle import picodet as opico
    from pdf_table_amd.ocr_layout_task import OcrLayoutTask
    from pdf_table_amd.synth_pages import make_page
    m = X.seeded(X.PicoLike(ncls=5, levels=4), 21)
    with torch.no_grad():
        for h in m.heads:
            h.bias[:5] -= 2.0                  # a few hundred anchors above the 0.5 score threshold, ~45 regions after the NMS
            h.weight[:5] *= 3.0
    onnx_file = tmp_path / "model.onnx"
    onnx_file.write_bytes(X.torch_export(m, torch.zeros(1, 3, 800, 608)))
    task = OcrLayoutTask(model="picodet", task_type="en", task_path=str(tmp_path), engine=eng)
    page = make_page(2, 1024)[0]
    got = task(page)[0]
    xl, sf = opico.picodet_preprocess(page)
    with torch.no_grad():
        outs = [o.numpy() for o in m(torch.from_numpy(xl)[None])]
    assert task.get_onnx_output_dict(outs)["boxes"][0].shape == (1, 100 * 76, 5)
    # (i) the graph outputs on the engine (pre-processing kernel + executor, bf16) against the module on the oracle's pre-processing
    acts = task._exec.run_device(eng.layout_preprocess(torch.from_numpy(page[None]).cuda(), 800, 608), 3)
    outs_e = [a.t[:, 0, :, :a.c].float().cpu().numpy() for a in acts]
    for o_e, o_m in zip(outs_e, outs):
        assert o_e.shape == o_m.shape and np.abs(o_e - o_m).max() <= 4e-2 * max(1.0, float(np.abs(o_m).max()))
    # (ii) the door's host half (get_onnx_output_dict split + LayoutStage.decode_outputs) against the oracle's post-processor (pinned to the
    # reference's OCRPicodetPostProcessor) on the SAME outputs: the seeded head's scores sit around the 0.5 threshold, so the comparison
    # with the fp32 module's own regions would only measure which side of it bf16 noise lands on
    want = opico.picodet_postprocess(outs_e[:4], outs_e[4:], list(page.shape[:2]), sf, [800, 608], opico.LABELS["en"])
    want_m = opico.picodet_postprocess(outs[:4], outs[4:], list(page.shape[:2]), sf, [800, 608], opico.LABELS["en"])
    print(f"layout ONNX door: {len(got)} regions on the engine (bf16), {len(want_m)} from the module in fp32")
    assert len(got) == len(want) >= 3
    for g, w in zip(got, want):
        assert g["label"] == w["label"] and g["category_id"] == w["category_id"] and abs(float(g["score"]) - float(w["score"])) <= 1e-6
        assert np.abs(np.asarray(g["bbox"], np.float64) - np.asarray(w["bbox"], np.float64)).max() <= 1e-3


def test_classifier_task_onnx_door(eng, tmp_path):
    """ClsImagePulcTask(task_path=<model.onnx>) with a classifier the importer has no dedicated graph for: Pillow-exact pre-processing kernel ->
    generic executor -> Topk; against the module in fp32 on the oracle's pre-processing"""
    import onnx_export as X
    from oracle import pil_resize as opl
    from pdf_table_amd.cls_image_pulc_task import ClsImagePulcTask
    from pdf_table_amd.synth_pages import make_page
    from test_gpu_onnx_exec import LcNetLike
    m = X.seeded(LcNetLike(classes=4), 22)
    (tmp_path / "inference.onnx").write_bytes(X.torch_export(m, torch.zeros(1, 3, 224, 224)))
    task = ClsImagePulcTask(task_type="text_image_orientation", task_path=str(tmp_path), engine=eng)
    imgs = [make_page(5, 1024)[0][:600, :800].copy(), make_page(6, 1024)[0][200:500, 100:900].copy()]
    got = task(imgs)
    for img, g in zip(imgs, got):
        x = opl.pplcnet_preprocess(img, 224, 224)
        with torch.no_grad():
            p = torch.softmax(m(torch.from_numpy(np.ascontiguousarray(x))[None]), -1)[0].numpy()
        order = np.argsort(p)[::-1]
        print("classifier ONNX door:", g, "module top-2", order[:2].tolist(), np.round(p[order[:2]], 4).tolist())
        assert len(g["class_ids"]) == 2 and abs(g["scores"][0] - float(p[g["class_ids"][0]])) <= 0.05
        assert g["class_ids"][0] == int(order[0]) or p[order[0]] - p[g["class_ids"][0]] <= 0.05


def test_recognition_task_onnx_door(eng, tmp_path):
    """OcrRecognitionTask(model="PP-OCRv4", task_path=<inference.onnx + dictionary>): PPOcrRecPreProcessor kernel -> executor (conv stem, two
    attention blocks, CTC head with Softmax) -> CTCLabelDecode, against the exported module in fp32 on the oracle's pre-processing"""
    import onnx_export as X
    from oracle import rec_pp as orp
    from pdf_table_amd.ocr_recognition_task import OcrRecognitionTask
    from pdf_table_amd.rec_postprocess import CTCLabelDecode
    from pdf_table_amd.synth_pages import make_page
    chars = [chr(0x4E00 + i) for i in range(95)]
    (tmp_path / "ppocr_keys_v1.txt").write_text("\n".join(chars) + "\n", encoding="utf-8")
    m = X.seeded(X.SvtrTiny(classes=97), 31)                      # blank + 95 characters + space
    (tmp_path / "inference.onnx").write_bytes(X.torch_export(m, torch.zeros(1, 3, 48, 320)))
    task = OcrRecognitionTask(model="PP-OCRv4", task_type="ch", task_path=str(tmp_path), engine=eng)
    page = make_page(4, 1024)[0]
    crops = [page[100:130, 50:240].copy(), page[300:340, 400:600].copy(), page[500:520, 100:200].copy()]
    got = task(crops)
    assert isinstance(got, list) and len(got) == 3 and all(isinstance(t, str) for t in got)
    ctc = CTCLabelDecode(str(tmp_path / "ppocr_keys_v1.txt"), use_space_char=True)
    for c, g in zip(crops, got):
        (b,) = orp.rec_pp_preprocess([c])
        with torch.no_grad():
            p = m(torch.from_numpy(np.ascontiguousarray(b["image"]))).numpy()
        (want, _), = ctc(p)
        import difflib
        ratio = difflib.SequenceMatcher(None, g, want).ratio()
        print(f"recogniser ONNX door: {len(g)} characters on the engine (bf16), {len(want)} from the module (fp32), similarity {ratio:.3f}")
        assert len(want) > 5 and ratio >= 0.85              # a random-init head: a few near-tie tokens flip in bf16
    # precision="fp32": the executor's tolerance mode behind the same door -- the strings of the fp32 module, token for token
    task32 = OcrRecognitionTask(model="PP-OCRv4", task_type="ch", task_path=str(tmp_path), engine=eng, precision="fp32")
    assert task32._exec.precision == "bf16x3"
    for c, g in zip(crops, task32(crops)):
        (b,) = orp.rec_pp_preprocess([c])
        with torch.no_grad():
            p = m(torch.from_numpy(np.ascontiguousarray(b["image"]))).numpy()
        (want, _), = ctc(p)
        top2 = np.sort(p[0], -1)[:, -2:]
        ties = int(((top2[:, 1] - top2[:, 0]) <= 2e-3).sum())
        print(f"recogniser ONNX door, precision='fp32' (bf16x3): {'identical' if g == want else 'DIFFERENT'} strings of {len(want)} characters ({ties} oracle ties)")
        assert g == want or ties > 0
    with pytest.raises(RuntimeError, match="no model.onnx"):
        OcrRecognitionTask(model="PP-OCRv4", task_type="ch", task_path=str(tmp_path / "missing"), engine=eng)
