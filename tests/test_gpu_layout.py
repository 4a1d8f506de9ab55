"""GPU parity tests of the layout stage (PicoDet), through the C ABI.  Float work: PT_PRECISION_BF16X3 within 1e-3
(relative to the head's scale) of the oracle (itself pinned to the reference's LCNet / CSPPAN / PicoHead modules)."""
import os

import numpy as np
import pytest
import torch

from oracle import picodet as op
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import picodet_state_dict
from pdf_table_amd.weights import pack_picodet

pytestmark = pytest.mark.gpu
TOL_REL = 1e-3


def _x4(x, split=False):
    n, _, H, W = x.shape
    nhwc = x.permute(0, 2, 3, 1)
    if not split:
        x4 = torch.zeros(n, H, W, 4)
        x4[..., :3] = nhwc
        return x4.to(torch.bfloat16)
    hi = nhwc.to(torch.bfloat16).float()
    lo = (nhwc - hi).to(torch.bfloat16).float()
    x8 = torch.zeros(n, H, W, 8)
    x8[..., :3] = hi
    x8[..., 4:7] = lo
    return x8.to(torch.bfloat16)


@pytest.fixture(scope="module")
def pico_sd():
    return picodet_state_dict(seed=41, num_classes=5)


@pytest.fixture(scope="module")
def eng(pico_sd):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_PICODET, pack_picodet(pico_sd, 5))
    yield e
    e.close()


def _ref_logits(sd, x):
    """oracle head outputs before the sigmoid: [n, A, 5 + 32] per level"""
    with torch.no_grad():
        sc, bx = op.picodet_forward(sd, x, 5)
    return [torch.cat([torch.logit(s.double()).float(), b], 2) for s, b in zip(sc, bx)], sc, bx


@pytest.mark.parametrize("case", [(2, 37, 45, 64, 5, 2), (1, 8, 16, 128, 3, 1), (3, 19, 70, 32, 3, 2), (2, 25, 19, 256, 5, 2), (1, 13, 10, 96, 5, 0),
                                  (2, 50, 38, 512, 5, 2), (1, 1, 1, 64, 5, 2)])
def test_dwconv_tile_kernel_equals_register_kernel(eng, case, monkeypatch):
    """dwconv_tile_kernel (stride-1 depthwise convs: the input tile + halo of 32 / 64 channels staged in LDS, four workgroups per CU) against the register
    kernels (PT_DWCONV_TILE=0, read per call): every tap in the same (ky, kx) order per output, so every bit is the same; maps smaller than a tile, widths
    that are not multiples of the 16- / 32-column tile, 96 channels (three 32-channel blocks), a single pixel; and both against torch fp32."""
    import torch.nn.functional as F
    B, H, W, C, k, act = case
    g = torch.Generator().manual_seed(H * 131 + C)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16)
    w = torch.randn(k * k, C, generator=g) * 0.2
    b = torch.randn(C, generator=g) * 0.1
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("PT_DWCONV_TILE", sw)
        y = eng.op_dwconv(x.cuda(), w.cuda(), b.cuda(), k, 1, act)
        torch.cuda.synchronize()
        outs.append(y.view(torch.int16).cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.t().reshape(C, 1, k, k), b, 1, k // 2, groups=C)
    ref = F.hardswish(ref) if act == 2 else (F.relu(ref) if act == 1 else ref)
    got = torch.from_numpy(outs[0]).view(torch.bfloat16).float().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-3).all()), float(err.max())
    # the pair modes' instantiation (fp32 sums staged in LDS, four output rows per workgroup): the same comparison on (hi | lo) tensors, and against
    # torch fp32 on the UN-rounded values (the pair carries 16 mantissa bits)
    xf = torch.randn(B, H, W, C, generator=g)
    hi = xf.to(torch.bfloat16)
    xs = torch.cat([hi, (xf - hi.float()).to(torch.bfloat16)], -1).contiguous()
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("PT_DWCONV_TILE", sw)
        y = eng.op_dwconv(xs.cuda(), w.cuda(), b.cuda(), k, 1, act, split=True)
        torch.cuda.synchronize()
        outs.append(y.view(torch.int16).cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    o = torch.from_numpy(outs[0]).view(torch.bfloat16).float()
    got = (o[..., :C] + o[..., C:]).permute(0, 3, 1, 2)
    xv = (xs[..., :C].float() + xs[..., C:].float()).permute(0, 3, 1, 2)
    ref = F.conv2d(xv, w.t().reshape(C, 1, k, k), b, 1, k // 2, groups=C)
    ref = F.hardswish(ref) if act == 2 else (F.relu(ref) if act == 1 else ref)
    assert float((got - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(1, 160, 128), (2, 224, 192), (1, 320, 256)])
def test_layout_net_x3_matches_oracle(eng, pico_sd, shape):
    n, H, W = shape
    g = torch.Generator().manual_seed(500 + H)
    x = torch.randn(n, 3, H, W, generator=g)
    with torch.no_grad():
        sc, bx = op.picodet_forward(pico_sd, x, 5)
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        heads = eng.layout_forward_net(_x4(x, split=True).cuda())
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    for l in range(4):
        h = heads[l].cpu()
        assert h.shape[1] == sc[l].shape[1], (l, h.shape, sc[l].shape)
        ds = (torch.sigmoid(h[..., :5]) - sc[l]).abs().max().item()
        db = (h[..., 5:37] - bx[l]).abs().max().item()
        print(f"layout x3 {shape} level {l}: max|dscore|={ds:.2e} max|dbox|={db:.2e} (scale {bx[l].abs().max().item():.1f})")
        assert ds <= TOL_REL and db <= TOL_REL * max(1.0, bx[l].abs().max().item())


def test_layout_net_matches_reference_golden(eng, golden_dir):
    gold = np.load(os.path.join(golden_dir, "picodet.npz"))
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        for tag in ("a", "b"):
            heads = eng.layout_forward_net(_x4(torch.from_numpy(gold[f"x_{tag}"]), split=True).cuda())
            for l in range(4):
                h = heads[l].cpu()
                assert (torch.sigmoid(h[..., :5]).numpy() - gold[f"score{l}_{tag}"]).__abs__().max() <= TOL_REL
                assert np.abs(h[..., 5:37].numpy() - gold[f"box{l}_{tag}"]).max() <= TOL_REL * max(1.0, np.abs(gold[f"box{l}_{tag}"]).max())
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_layout_net_bf16_drift(eng, pico_sd):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 3, 224, 192, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        sc, bx = op.picodet_forward(pico_sd, x, 5)
    heads = eng.layout_forward_net(_x4(x).cuda())
    for l in range(4):
        h = heads[l].cpu()
        assert (h[..., 5:37] - bx[l]).abs().max().item() <= 0.1 * max(1.0, bx[l].abs().max().item())


def test_layout_preprocess_and_candidates(eng):
    """pt_layout_preprocess == the oracle's pre-process; candidates == every anchor above the threshold, with its raw values"""
    from pdf_table_amd.synth_pages import make_page
    page = make_page(5, 1024)[0]
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        x = eng.layout_preprocess(torch.from_numpy(page[None]).cuda(), 800, 608).float().cpu()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    ref, sf = op.picodet_preprocess(page)
    got = (x[0, ..., :3] + x[0, ..., 4:7]).permute(2, 0, 1).numpy()
    assert np.abs(got - ref).max() <= 2e-5
    assert sf == [800 / 1024, 608 / 1024]
    fh, fw = eng.layout_plan(800, 608)
    assert [a * b for a, b in zip(fh, fw)] == [7600, 1900, 475, 130]
    rng = np.random.default_rng(3)
    heads = [torch.from_numpy((rng.standard_normal((2, fh[l] * fw[l], 40)) * 1.5 - 2.0).astype(np.float32)).cuda() for l in range(4)]
    counts = torch.zeros(2, dtype=torch.int32, device="cuda")
    cands = torch.zeros(2, 4096, 48, device="cuda")
    from pdf_table_amd.engine import _ptr
    L.check(eng.lib.pt_layout_candidates(eng._h, *[_ptr(h) for h in heads], 2, 800, 608, 5, 0.5, 4096, _ptr(counts), _ptr(cands),
                                         eng._stream()), "pt_layout_candidates")
    counts = counts.cpu().numpy()
    for b in range(2):
        want = set()
        for l in range(4):
            s = torch.sigmoid(heads[l][b, :, :5]).max(1).values.cpu().numpy()
            want |= {(l, int(a)) for a in np.nonzero(s > 0.5)[0]}
        rec = cands[b, :counts[b]].cpu().numpy()
        got = {(int(r[:1].view(np.int32)[0]), int(r[1:2].view(np.int32)[0])) for r in rec}
        assert counts[b] == len(want) > 50 and got == want
        for r in rec[:20]:
            l, a = int(r[:1].view(np.int32)[0]), int(r[1:2].view(np.int32)[0])
            assert np.array_equal(r[2:42], heads[l][b, a].cpu().numpy())


def test_layout_stage_matches_oracle_chain(eng, pico_sd):
    """pages -> detections through the device path (BF16X3) == the oracle chain (pre-process, net, post-process), up to
    detections whose score sits within 2e-3 of the threshold / NMS boundary"""
    from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
    from pdf_table_amd.synth_pages import make_page
    pages = np.stack([make_page(i, 1024)[0] for i in (0, 1, 2, 3)])
    total = 0
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        got = LayoutStage(eng, PicodetConfig(task_type="en"))(torch.from_numpy(pages).cuda())
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    for i in range(4):
        x, sf = op.picodet_preprocess(pages[i])
        with torch.no_grad():
            sc, bx = op.picodet_forward(pico_sd, torch.from_numpy(x)[None], 5)
        ref = op.picodet_postprocess([s.numpy() for s in sc], [b.numpy() for b in bx], [1024, 1024], sf, [800, 608], op.LABELS["en"])
        total += len(ref)
        rb = np.array([r["bbox"] for r in ref]).reshape(-1, 4)
        gb = np.array([g["bbox"] for g in got[i]]).reshape(-1, 4)
        matched = 0
        for k, r in enumerate(ref):
            d = np.abs(gb - rb[k]).max(1) if len(gb) else np.array([1e9])
            j = int(np.argmin(d))
            if d[j] <= 0.05 and got[i][j]["category_id"] == r["category_id"] and abs(got[i][j]["score"] - r["score"]) <= 2e-3:
                matched += 1
        print(f"layout page {i}: {len(ref)} reference detections, {len(got[i])} device detections, {matched} matched")
        assert matched >= 0.9 * len(ref) and abs(len(got[i]) - len(ref)) <= max(2, len(ref) // 10)
    assert total > 10


def _iou(a, b):
    ix = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    iy = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    return ix * iy / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - ix * iy)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_fitted_table_head_feeds_table_structure(precision):
    """bench.py's layout -> table-structure chain: the seeded PicoDet with the fitted stride-64 head branch (tools/fit_layout_head.py) labels the tables of
    the generator's pages "table" on the ENGINE, in both arithmetic modes, close enough to the generator's rectangles (grown by 8 px, the fit's target)
    for the table-structure stage to see whole tables; OcrTablePipeline._layout_table_boxes (get_layout_by_type, score >= 0.2) is the hand-off"""
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig, layout_tables
    from pdf_table_amd.synth_pages import make_page
    x3 = precision == "bf16x3"
    e = HipEngine(0)
    try:
        e.set_precision(L.PT_PRECISION_BF16X3 if x3 else L.PT_PRECISION_BF16)
        e.load_weights(L.PT_MODEL_PICODET, pack_picodet(picodet_state_dict(seed=4, num_classes=5, table_head=True), 5, x3=x3))
        made = [make_page(i) for i in (0, 1, 2, 3, 100, 101, 300, 511)]
        pages = torch.from_numpy(np.stack([m[0] for m in made])).cuda()
        res = LayoutStage(e, PicodetConfig(task_type="en"))(pages)
        worst, n = 1.0, 0
        for m, lay in zip(made, res):
            gt = np.asarray(m[1]["tables"], dtype=np.float64).reshape(-1, 4) + np.array([-8, -8, 8, 8])
            tabs = layout_tables(lay, "table", 0.2)
            assert len(tabs) == len(gt), (len(tabs), len(gt))
            for g_, t in zip(sorted(gt.tolist(), key=lambda b: b[1]), tabs):      # layout_tables: top to bottom
                worst = min(worst, _iou(g_, [float(v) for v in t["bbox"]]))
                n += 1
        print(f"fitted table head [{precision}]: {n} tables on 8 pages, worst IoU with the generator's rectangle {worst:.3f}")
        assert worst >= 0.9
    finally:
        e.close()


@pytest.mark.parametrize("shape", [(2, 160, 128), (1, 800, 608), (3, 96, 224)])
def test_fused_depthwise_pointwise_is_bit_identical(eng, shape, monkeypatch):
    """dwpw_kernel (depthwise k x k + pointwise 1x1 of a DepthwiseSeparable / DPModule / PicoFeat block in ONE launch: the depthwise tile goes to the MFMA
    through LDS, never to HBM) against the two launches (PT_DWPW=0, read per call): the same taps in the same order, the same rounding of the depthwise
    output, the same K order in the pointwise sums -- every head value the same to the bit; sizes whose maps are not whole 4 x 32 tiles, and the bench's
    800 x 608."""
    n, H, W = shape
    g = torch.Generator().manual_seed(41 + H)
    x = _x4(torch.randn(n, 3, H, W, generator=g)).cuda()
    monkeypatch.setenv("PT_DWPW", "2")      # every stride-1 pair (the default fuses the 3 x 3 pairs only: the 5 x 5 ones measured slower fused)
    eng.profile_enable(1)
    fused = [h.clone() for h in eng.layout_forward_net(x)]
    torch.cuda.synchronize()
    labels = list(eng.profile_read_labels())
    eng.profile_enable(False)
    assert any(k.startswith("dw5 s1 + pw") for k in labels) and any(k.startswith("dw3 s1 + pw") for k in labels), labels
    monkeypatch.setenv("PT_DWPW", "0")
    two = [h.clone() for h in eng.layout_forward_net(x)]
    monkeypatch.delenv("PT_DWPW")
    dflt = eng.layout_forward_net(x)
    torch.cuda.synchronize()
    for a, b, c in zip(fused, two, dflt):
        assert torch.isfinite(a).all() and a.abs().max() > 0
        assert torch.equal(a, b) and torch.equal(c, b)
