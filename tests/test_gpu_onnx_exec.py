"""Generic ONNX layer-list executor on the GPU (pdf_table_amd/onnx_exec.py): graphs the engine has no
dedicated launch graph for run layer by layer through the single-operator C-ABI entry points.  Graphs come from PyTorch's
own exporter (tools/onnx_export.torch_export: BatchNorm folded, hardswish as HardSigmoid * x below opset 14, nn.Upsample as
Resize); the checker is the PyTorch module itself in fp32 (and oracle/onnx_ref on the exported bytes: the CPU graph
interpreter the importer tests use).  Arithmetic is bf16 with fp32 accumulation: tolerances are relative to the output scale."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

pytestmark = pytest.mark.gpu


class _SE(nn.Module):
    def __init__(self, c, r=4):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc1, self.fc2 = nn.Conv2d(c, c // r, 1), nn.Conv2d(c // r, c, 1)

    def forward(self, x):
        g = torch.relu(self.fc1(self.pool(x)))
        return x * nn.functional.hardsigmoid(self.fc2(g))


class _DwSep(nn.Module):
    def __init__(self, cin, cout, k, stride, se):
        super().__init__()
        self.dw = nn.Sequential(nn.Conv2d(cin, cin, k, stride, k // 2, groups=cin, bias=False), nn.BatchNorm2d(cin), nn.Hardswish())
        self.se = _SE(cin) if se else None
        self.pw = nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.Hardswish())

    def forward(self, x):
        x = self.dw(x)
        if self.se is not None:
            x = self.se(x)
        return self.pw(x)


class LcNetLike(nn.Module):
    """PP-LCNet-shaped classifier: 3x3/s2 stem, depthwise-separable blocks (3x3 and 5x5, two with SE), global pool, Linear --
    channel counts that are NOT multiples of the GEMM tile (16, 24, 40, 96)"""

    def __init__(self, classes=10):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.Hardswish())
        self.blocks = nn.Sequential(_DwSep(16, 24, 3, 1, False), _DwSep(24, 40, 3, 2, False), _DwSep(40, 40, 5, 1, True),
                                    _DwSep(40, 96, 5, 2, True))
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(96, classes)

    def forward(self, x):
        return self.fc(torch.flatten(self.pool(self.blocks(self.stem(x))), 1))


class FpnLike(nn.Module):
    """ResNet-ish trunk with a residual add, MaxPool(3,2,1), a two-level FPN (nearest x2 Upsample + Add, Concat), a 2x2
    transposed conv and a sigmoid head"""

    def __init__(self):
        super().__init__()
        def cbr(ci, co, k, s):
            return nn.Sequential(nn.Conv2d(ci, co, k, s, k // 2, bias=False), nn.BatchNorm2d(co), nn.ReLU())
        self.c1 = cbr(3, 32, 3, 2)
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.b1a, self.b1b = cbr(32, 32, 3, 1), nn.Sequential(nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32))
        self.c2 = cbr(32, 80, 3, 2)
        self.l1, self.l2 = nn.Conv2d(32, 48, 1), nn.Conv2d(80, 48, 1)
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.s1, self.s2 = nn.Conv2d(48, 24, 3, 1, 1), nn.Conv2d(48, 24, 3, 1, 1)
        self.head = nn.Sequential(nn.Conv2d(48, 16, 3, 1, 1, bias=False), nn.BatchNorm2d(16), nn.ReLU(),
                                  nn.ConvTranspose2d(16, 16, 2, 2), nn.BatchNorm2d(16), nn.ReLU(), nn.ConvTranspose2d(16, 1, 2, 2), nn.Sigmoid())

    def forward(self, x):
        x = self.pool(self.c1(x))
        x = torch.relu(self.b1b(self.b1a(x)) + x)
        y = self.c2(x)
        p2 = self.l2(y)
        p1 = self.l1(x) + self.up(p2)
        f = torch.cat([self.s1(p1), self.up(self.s2(p2))], 1)
        return self.head(f)


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.weight.data = 0.8 + 0.4 * torch.rand(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
            mod.running_mean = 0.1 * torch.randn(mod.running_mean.shape, generator=g)
            mod.running_var = 0.7 + 0.6 * torch.rand(mod.running_var.shape, generator=g)
    return m.eval()


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _check(model, x, eng, tol_rel, precision="bf16"):
    from onnx_export import torch_export
    from oracle import onnx_ref
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    from pdf_table_amd.onnx_proto import parse_model
    blob = torch_export(model, x)
    ex = HipGraphExecutor(blob, engine=eng, precision=precision)
    (got,) = ex.run(x.numpy())
    (again,) = ex.run(x.numpy())                      # operands are cached after the first run
    with torch.no_grad():
        want = model(x).numpy()
    (ref,) = onnx_ref.run(parse_model(blob), {ex.inputs[0].name: x.numpy()})
    assert got.shape == want.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got, again)
    scale = float(np.abs(want).max())
    assert np.abs(ref - want).max() <= 1e-4 * max(scale, 1.0)        # the exported graph is the module
    d = float(np.abs(got - want).max())
    print(f"{type(model).__name__} [{precision}]: max|d| = {d:.3e} on scale {scale:.2f} ({len(ex.layers)} layers)")
    assert d <= tol_rel * scale + 1e-3
    return ex


def test_lcnet_like_classifier(eng):
    """dw 3x3 / 5x5 (stride 1 / 2) + hardswish, SE (pool, two 1x1 convs, hardsigmoid, channel scale), Linear after the pool"""
    torch.manual_seed(0)
    m = _randomise(LcNetLike(), 1)
    x = torch.randn(3, 3, 64, 96)
    ex = _check(m, x, eng, 4e-2)
    kinds = {l.op for l in ex.layers}
    assert {"conv", "gap", "mul", "gemm"} <= kinds


def test_lcnet_like_classifier_tolerance_mode(eng):
    """the same graph in the executor's tolerance mode (precision="bf16x3": (hi | lo) activations, three-pass convolutions, every other operator on
    hi + lo in fp32): within 1e-3 of the fp32 module (north_star's bound on float logits) -- depthwise, SE pool / gate / scale, hardswish /
    hardsigmoid, Gemm"""
    torch.manual_seed(0)
    m = _randomise(LcNetLike(), 1)
    x = torch.randn(3, 3, 64, 96)
    _check(m, x, eng, 1e-3, precision="bf16x3")


def test_fpn_like_detector_tolerance_mode(eng):
    """MaxPool(3,2,1), residual Add, nearest x2 Resize, Concat, 2x2 transposed convs (pixel-shuffle GEMM), Sigmoid in the tolerance mode: 1e-3"""
    torch.manual_seed(0)
    m = _randomise(FpnLike(), 2)
    x = torch.randn(2, 3, 64, 96)
    _check(m, x, eng, 1e-3, precision="bf16x3")


def test_fpn_like_detector(eng):
    """MaxPool(3,2,1), residual Add, nearest x2 Resize, Concat, 2x2 transposed convs, Sigmoid; 80 / 48 / 24 channels"""
    torch.manual_seed(0)
    m = _randomise(FpnLike(), 2)
    x = torch.randn(2, 3, 64, 96)
    ex = _check(m, x, eng, 4e-2)
    kinds = {l.op for l in ex.layers}
    assert {"conv", "convT", "maxpool", "add", "resize", "concat"} <= kinds


def test_db_resnet18_export_runs_generically(eng):
    """the torch-exported DB-ResNet18 (7x7 stem -> the stem kernel) through the generic executor against the fp32 oracle"""
    from onnx_export import export_db_resnet18
    from oracle import db_net
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    sd = db_resnet18_state_dict(seed=0)
    ex = HipGraphExecutor(export_db_resnet18(sd), engine=eng)
    x = np.random.default_rng(1).standard_normal((2, 3, 96, 160)).astype(np.float32)
    (y,) = ex.run(x)
    with torch.no_grad():
        ref = db_net.db_forward_fp32(sd, torch.from_numpy(x)).numpy()
    d = float(np.abs(y - ref).max())
    print(f"generic executor, DB-ResNet18 export: max|dprob| = {d:.3e}; {len(ex._fuse)} residual adds folded into their convolutions")
    assert y.shape == ref.shape and d <= 0.1            # bf16 class (the dedicated graph measures 0.036 on this input)
    # every BasicBlock's `conv2 + identity -> ReLU` (8 of them) runs as ONE launch: Add and ReLU folded into the convolution's epilogue at load time
    assert len(ex._fuse) >= 8 and all(a is not None for _, _, a in ex._fuse.values())
    # the tolerance mode on the same graph: the reference's detector through onnxruntime in fp32 is what the 1e-3 contract is about
    ex3 = HipGraphExecutor(export_db_resnet18(sd), engine=eng, precision="bf16x3")
    (y3,) = ex3.run(x)
    d3 = float(np.abs(y3 - ref).max())
    print(f"generic executor, DB-ResNet18 export, precision bf16x3: max|dprob| = {d3:.3e}")
    assert d3 <= 1e-3


def test_unsupported_layers_fail_loudly(eng):
    from onnx_export import torch_export
    from pdf_table_amd.onnx_exec import HipGraphExecutor
    from pdf_table_amd.onnx_import import UnsupportedOnnxGraph

    class Dil(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(3, 8, 3, 1, 2, dilation=2)

        def forward(self, x):
            return self.c(x)
    ex = HipGraphExecutor(torch_export(Dil().eval(), torch.randn(1, 3, 16, 16)), engine=eng)
    with pytest.raises(UnsupportedOnnxGraph, match="dilation|padding"):
        ex.run(np.zeros((1, 3, 16, 16), np.float32))

    class Sm(nn.Module):
        def forward(self, x):
            return torch.softmax(x, 1)
    ex = HipGraphExecutor(torch_export(Sm().eval(), torch.randn(1, 3, 16, 16)), engine=eng)
    with pytest.raises(UnsupportedOnnxGraph, match="no executor|softmax|Softmax|no kernel reads"):
        ex.run(np.zeros((1, 3, 16, 16), np.float32))


def test_session_surface_falls_back_to_the_generic_executor(eng):
    """HipOnnxSession (the reference's predictor.run surface) on a graph recognise() has no launch graph for: arch "generic",
    outputs selected by name, fp16 feeds come back as fp16"""
    from onnx_export import torch_export
    from pdf_table_amd.onnx_import import HipOnnxSession
    torch.manual_seed(0)
    m = _randomise(FpnLike(), 3)
    x = torch.randn(1, 3, 32, 64)
    sess = HipOnnxSession(torch_export(m, x), engine=eng)
    assert sess.arch == "generic" and sess.get_inputs()[0].name == "x"
    (y,) = sess.run(["y"], {"x": x.numpy()})
    with torch.no_grad():
        want = m(x).numpy()
    assert y.shape == want.shape and np.abs(y - want).max() <= 4e-2 * np.abs(want).max() + 1e-3
    (y16,) = sess.run(None, {"x": x.numpy().astype(np.float16)})
    assert y16.dtype == np.float16


def test_detection_task_serves_an_unknown_onnx_detector(tmp_path, eng):
    """OcrDetectionTask(model="db_pp", task_path=<dir with model.onnx>) with a detector recognise() does not know: the graph runs
    through the generic executor between the engine's PP-OCR pre-processing and its bitmap / box kernels -- the probability
    map equals the PyTorch module on the same pre-processed pixels (bf16 class), boxes come back in the reference's format"""
    from onnx_export import torch_export
    from pdf_table_amd import lib as L
    from pdf_table_amd.ocr_detection_task import OcrDetectionTask
    from pdf_table_amd.synth_pages import make_page
    torch.manual_seed(0)
    m = _randomise(FpnLike(), 5)
    (tmp_path / "model.onnx").write_bytes(torch_export(m, torch.randn(1, 3, 64, 64)))
    task = OcrDetectionTask(model="db_pp", task_path=str(tmp_path), engine=eng, thresh=0.3)
    page = make_page(2)[0][:480, :640].copy()
    out = task(page)
    assert len(out) == 1 and out[0].ndim == 2 and out[0].shape[1] == 8
    pages = torch.from_numpy(page[None]).cuda()
    prob, bitmap, ev = task._stage.forward(pages)
    x4 = eng.det_preprocess(pages, L.PT_DET_PRE_DB_PP)
    with torch.no_grad():
        want = m(x4[..., :3].float().permute(0, 3, 1, 2).cpu())[:, 0].numpy()
    got = prob.cpu().numpy()
    assert got.shape == want.shape
    d = float(np.abs(got - want).max())
    print(f"generic detector through OcrDetectionTask: max|dprob| = {d:.3e}")
    assert d <= 4e-2


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("fp16", 6e-3)])
def test_detection_task_route_in_the_tolerance_and_half_modes(tmp_path, precision, tol):
    """The PRODUCT route of an unknown ONNX detector -- OcrDetectionTask(precision=...) on its own engine: pt_det_preprocess -> generic executor ->
    probability map -- against the fp32 PyTorch module on the ORACLE's pre-processed pixels (oracle/db_pre.py, fp32): precision="fp32" must hold the
    executor's 1e-3 contract THROUGH THE TASK (the (hi | lo) pre-process output used to lose its lo half on the way into the graph, i.e.
    the image was rounded to 8 significant bits first); precision="fp16" is the reference's own default arithmetic (PT_PRECISION_F16)."""
    from onnx_export import torch_export
    from oracle import db_pre
    from pdf_table_amd import lib as L
    from pdf_table_amd.ocr_detection_task import OcrDetectionTask
    from pdf_table_amd.synth_pages import make_page
    torch.manual_seed(0)
    m = _randomise(FpnLike(), 5)
    (tmp_path / "model.onnx").write_bytes(torch_export(m, torch.randn(1, 3, 64, 64)))
    task = OcrDetectionTask(model="db_pp", task_path=str(tmp_path), thresh=0.3, precision=precision)
    try:
        assert task._engine.precision == {"fp32": L.PT_PRECISION_BF16X3, "fp16": L.PT_PRECISION_F16}[precision]
        page = make_page(2)[0][:480, :640].copy()
        chw, _ = db_pre.preprocess_db_pp(page)
        with torch.no_grad():
            want = m(torch.from_numpy(np.ascontiguousarray(chw))[None])[:, 0].numpy()
        prob, _, _ = task._stage.forward(torch.from_numpy(page[None]).cuda())
        got = prob.cpu().numpy()
        assert got.shape == want.shape
        d = float(np.abs(got - want).max())
        print(f"generic detector through OcrDetectionTask(precision={precision!r}): max|dprob| = {d:.3e}")
        assert d <= tol
        out = task(page)
        assert len(out) == 1 and out[0].ndim == 2 and out[0].shape[1] == 8
    finally:
        task._engine.close()


class VdLike(nn.Module):
    """ResNet-vd style pieces: AvgPool(2, 2) in the shortcut, a BatchNorm that follows an Add (nothing to fold it into), ReLU6"""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Sequential(nn.Conv2d(3, 24, 3, 1, 1, bias=False), nn.BatchNorm2d(24), nn.ReLU6())
        self.c2 = nn.Sequential(nn.Conv2d(24, 40, 3, 2, 1, bias=False), nn.BatchNorm2d(40))
        self.short = nn.Sequential(nn.AvgPool2d(2, 2), nn.Conv2d(24, 40, 1, bias=False))
        self.bn = nn.BatchNorm2d(40)
        self.out = nn.Conv2d(40, 8, 1)

    def forward(self, x):
        x = self.c1(x)
        y = self.bn(self.c2(x) + self.short(x))
        return self.out(torch.relu(y))


def test_vd_like_pieces(eng):
    torch.manual_seed(0)
    m = _randomise(VdLike(), 4)
    ex = _check(m, torch.randn(2, 3, 32, 64), eng, 4e-2)
    assert {"avgpool", "bn"} <= {l.op for l in ex.layers}
