"""GPU parity tests of the table-structure stage (Lore), through the C ABI.

Float work: PT_PRECISION_BF16X3 within 1e-3 (relative to the head's scale) of the oracle's fp32 restatement of DLASeg and
of the reference module's golden outputs; PT_PRECISION_BF16 bounded drift."""
import os

import numpy as np
import pytest
import torch

from oracle import lore_net
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import lore_dla34_state_dict
from pdf_table_amd.weights import pack_lore_dla34

pytestmark = pytest.mark.gpu

TOL_REL = 1e-3      # north_star: "within 1e-3 on float logits" -- relative to max(1, max|head|)
HEADS = ("hm", "st", "wh", "ax", "cr", "reg")


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
def lore_sd():
    return lore_dla34_state_dict(seed=21)


@pytest.fixture(scope="module")
def eng(lore_sd):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_sd))
    yield e
    e.close()


@pytest.fixture()
def eng_x3(eng):
    eng.set_precision(L.PT_PRECISION_BF16X3)
    yield eng
    eng.set_precision(L.PT_PRECISION_BF16)


def _cmp(got, ref, tag):
    worst = 0.0
    for k in HEADS:
        g = got[k].cpu().permute(0, 3, 1, 2)
        r = ref[k]
        assert g.shape == r.shape, (k, g.shape, r.shape)
        d = (g - r).abs().max().item()
        rel = d / max(1.0, r.abs().max().item())
        print(f"{tag} {k}: max|d|={d:.3e} scale={r.abs().max().item():.2f} rel={rel:.3e}")
        worst = max(worst, rel)
    return worst


@pytest.mark.parametrize("shape", [(1, 128, 160), (2, 64, 96), (1, 256, 256)])
def test_lore_net_x3_matches_fp32_oracle(eng_x3, lore_sd, shape):
    n, H, W = shape
    g = torch.Generator().manual_seed(300 + W)
    x = torch.randn(n, 3, H, W, generator=g)
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(lore_sd, x)
    got = eng_x3.tsr_forward_net(_x4(x, split=True).cuda())
    torch.cuda.synchronize()
    assert _cmp(got, ref, f"lore x3 {shape}") <= TOL_REL


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("shape", [(2, 96, 160), (1, 224, 96)])           # 1/4-resolution maps of 24 x 40 and 56 x 24 (1/32: 3 x 5 and 7 x 3): ragged 8 x 16 tiles in both directions
def test_offset_conv_inside_the_dcn_kernel_equals_its_own_launch(eng, lore_sd, mode, shape, monkeypatch):
    """dcn_fused64_kernel<..., OMF = 1> (the 27-channel offset / mask conv of a deformable conv in the kernel's prologue, lore/dcnv2.py:71-75; the default)
    in ALL sixteen layers (PT_DCN_FUSE_OM=2) against the two-launch form (PT_DCN_FUSE_OM=0: conv_igemm_kernel writes the fp32 `om` map, the DCN kernel reads it) through the whole DLA-34 + 16 DCN
    net: same tiles, same chunk -> tap -> k-step order; with eight waves the taps are summed in two groups, so the fp32 offsets may differ in the last bit --
    the heads agree to 1e-5 of their scale in the pair mode (the oracle bound on this net is 1e-3) and to bf16-drift level in bf16."""
    n, H, W = shape
    g = torch.Generator().manual_seed(900 + W)
    x = torch.randn(n, 3, H, W, generator=g)
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        xin = _x4(x, split=mode == "bf16x3").cuda()
        monkeypatch.setenv("PT_DCN_FUSE_OM", "0")
        two = {k: v.cpu().clone() for k, v in eng.tsr_forward_net(xin).items()}
        monkeypatch.setenv("PT_DCN_FUSE_OM", "2")         # 2: every layer that can (the default, 1, fuses the single-pass modes' C <= 128 layers)
        one = {k: v.cpu().clone() for k, v in eng.tsr_forward_net(xin).items()}
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    worst = 0.0
    for k in HEADS:
        rel = (one[k] - two[k]).abs().max().item() / max(1.0, two[k].abs().max().item())
        worst = max(worst, rel)
    print(f"offset conv inside the DCN kernel vs its own launch [{mode}, {shape}]: worst head difference {worst:.3e} of scale")
    assert bool(torch.isfinite(one["hm"]).all())
    assert worst <= (1e-4 if mode == "bf16x3" else 2e-2)      # measured 1.6e-5 / 5e-3: last-bit differences of the offsets, amplified by 16 stacked deformable convs
    if mode == "bf16x3":
        with torch.no_grad():
            ref = lore_net.dlaseg_forward(lore_sd, x)
        assert _cmp({k: v.cuda() for k, v in one.items()}, ref, f"lore x3 fused offset conv {shape}") <= TOL_REL


def test_lore_net_x3_matches_reference_golden(eng_x3, golden_dir):
    gold = np.load(os.path.join(golden_dir, "lore_dla34.npz"))
    for tag in ("a", "b"):
        x = torch.from_numpy(gold[f"x_{tag}"])
        got = eng_x3.tsr_forward_net(_x4(x, split=True).cuda())
        torch.cuda.synchronize()
        for k in HEADS:
            gk = got[k].cpu().permute(0, 3, 1, 2).numpy()
            gk = gk[:, ::8] if gk.shape[1] == 256 else gk
            r = gold[f"{k}_{tag}"]
            rel = np.abs(gk - r).max() / max(1.0, np.abs(r).max())
            assert rel <= TOL_REL, (tag, k, rel)


def test_lore_net_bf16_drift(eng, lore_sd):
    """throughput mode: bf16 activations through ~60 layers; drift bounded relative to each head's scale"""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 3, 128, 160, generator=g).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(lore_sd, x)
    got = eng.tsr_forward_net(_x4(x).cuda())
    torch.cuda.synchronize()
    assert _cmp(got, ref, "lore bf16") <= 0.1


# ---- decode -----------------------------------------------------------------------------------------------------
def _nhwc(t, cs):
    """NCHW f32 -> NHWC with channel stride cs (zero padded), as pt_tsr_forward_net writes the head maps"""
    n, c, h, w = t.shape
    o = torch.zeros(n, h, w, cs)
    o[..., :c] = t.permute(0, 2, 3, 1)
    return o.cuda()


@pytest.mark.parametrize("seed,H,W,rev", [(1, 80, 80, True), (2, 72, 96, True), (3, 80, 80, False), (4, 128, 128, True)])
def test_tsr_decode_matches_oracle(eng, seed, H, W, rev):
    """cells, their order, quads and logic features equal the oracle's restatement of process_detect_output (itself
    bit-exact against the reference on the golden cases).  Scores go through expf on the device: 1e-6."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lore_synth import synth_lore_heads
    from oracle import lore_decode as od
    heads = {k: torch.from_numpy(v) for k, v in synth_lore_heads(seed, H, W).items()}
    _, meta = od.lore_preprocess_geometry(4 * H, 4 * W, 4 * H, 4 * W)
    logi, ps, polys, results, raw = od.process_detect_output({k: v.clone() for k, v in heads.items()}, meta, wiz_rev=rev,
                                                             vis_thresh=0.2, return_raw=True)
    n_ref = logi.shape[1]
    dev = {k: _nhwc(heads[k], 256 if k in ("ax", "cr") else 8) for k in heads}
    counts, dets, lg = eng.tsr_decode(dev, wiz_rev=rev, vis_thresh=0.2)
    torch.cuda.synchronize()
    assert counts[0] == n_ref > 20, (counts, n_ref)
    d = dets[0, :n_ref].cpu().numpy()
    assert np.array_equal(d[:, :8], raw[:n_ref, :8]), np.abs(d[:, :8] - raw[:n_ref, :8]).max()
    assert np.allclose(d[:, 8], raw[:n_ref, 8], atol=1e-6, rtol=0)
    assert np.array_equal(lg[0, :n_ref].cpu().numpy(), logi[0].numpy())


def test_tsr_decode_batch_and_empty(eng):
    """two tables in one call == each alone; a map without peaks gives zero cells"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lore_synth import synth_lore_heads
    a = synth_lore_heads(5, 80, 80)
    b = synth_lore_heads(6, 80, 80)
    b["hm"][:] = -20.0                                          # nothing above any threshold
    both = {k: _nhwc(torch.from_numpy(np.concatenate([a[k], b[k]])), 256 if k in ("ax", "cr") else 8) for k in a}
    counts, dets, lg = eng.tsr_decode(both, wiz_rev=True, vis_thresh=0.2)
    one = {k: _nhwc(torch.from_numpy(a[k]), 256 if k in ("ax", "cr") else 8) for k in a}
    c1, d1, l1 = eng.tsr_decode(one, wiz_rev=True, vis_thresh=0.2)
    assert counts[0] == c1[0] > 20 and counts[1] == 0
    n = int(c1[0])
    assert torch.equal(dets[0, :n], d1[0, :n]) and torch.equal(lg[0, :n], l1[0, :n])


# ---- processor ----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def proc_sd():
    from pdf_table_amd.synth_weights import lore_processor_state_dict
    return lore_processor_state_dict(seed=31)


@pytest.fixture(scope="module")
def eng_proc(eng, proc_sd):
    from pdf_table_amd.weights import pack_lore_processor
    eng.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(proc_sd))
    return eng


def _proc_case(eng_proc, proc_sd, counts, use_pe, seed):
    from oracle import lore_processor as op
    g = torch.Generator().manual_seed(seed)
    n = len(counts)
    logi = torch.zeros(n, L.PT_TSR_MAX_CELLS, 256)
    dets = torch.zeros(n, L.PT_TSR_MAX_CELLS, 9)
    refs = []
    for t, c in enumerate(counts):
        logi[t, :c] = torch.randn(c, 256, generator=g)
        dets[t, :c, :8] = torch.rand(c, 8, generator=g) * 300 - 20           # some outside [0, 255]: clamp path
        if c:
            ps = dets[t, :c, :8].to(torch.int32).to(torch.float32).round().to(torch.int64).clamp(0, 255)[None]
            with torch.no_grad():
                refs.append(op.processor_forward(proc_sd, logi[t:t + 1, :c], ps if use_pe else None))
        else:
            refs.append(None)
    logic, stacked = eng_proc.tsr_process(logi.cuda(), dets.cuda(), counts, use_2dpe=use_pe)
    torch.cuda.synchronize()
    worst = 0.0
    for t, c in enumerate(counts):
        if not c:
            continue
        for got, ref in ((logic[t, :c].cpu(), refs[t][0][0]), (stacked[t, :c].cpu(), refs[t][1][0])):
            worst = max(worst, (got - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    return worst


@pytest.mark.parametrize("counts,use_pe", [([137], False), ([60, 0, 200, 1], False), ([90, 33], True), ([700], False)])
def test_tsr_processor_x3_matches_oracle(eng_proc, proc_sd, counts, use_pe):
    eng_proc.set_precision(L.PT_PRECISION_BF16X3)
    try:
        w = _proc_case(eng_proc, proc_sd, counts, use_pe, seed=sum(counts))
    finally:
        eng_proc.set_precision(L.PT_PRECISION_BF16)
    print("processor x3", counts, use_pe, w)
    assert w <= TOL_REL, w


def test_tsr_processor_bf16_drift(eng_proc, proc_sd):
    w = _proc_case(eng_proc, proc_sd, [137, 50], False, seed=5)
    print("processor bf16", w)
    assert w <= 0.1, w


# ---- pre-process and the whole stage -------------------------------------------------------------------------------
def test_tsr_preprocess_matches_oracle(eng):
    """pt_tsr_preprocess == oracle warpAffine + normalisation, bit for bit (fp32 values through the hi/lo pair)"""
    from oracle import lore_pre
    from pdf_table_amd.synth_pages import make_page
    from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
    pages = np.stack([make_page(3, 1024)[0], make_page(4, 1024)[0]])
    boxes = [np.array([[200, 100, 560, 420]]), np.array([[0, 0, 1024, 1024], [37, 500, 900, 777]])]
    cfg = LoreConfig()
    cfg.resolution = (320, 352)
    st = TsrStage(eng, cfg)
    tables, metas = st.tables((1024, 1024), boxes)
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        x = eng.tsr_preprocess(torch.from_numpy(pages).cuda(), tables, 320, 352, bgr=True).float().cpu()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    got = (x[..., :3] + x[..., 4:7]).permute(0, 3, 1, 2)
    k = 0
    for pi, bs in enumerate(boxes):
        for (x1, y1, x2, y2) in bs:
            crop = pages[pi][y1:y2, x1:x2][:, :, ::-1]
            ref, meta = lore_pre.lore_preprocess(np.ascontiguousarray(crop), 320, 352)
            assert np.array_equal(meta, metas[k])
            d = (got[k] - ref[0]).abs().max().item()
            assert d <= 2e-5, (k, d)            # hi + lo carries 16 mantissa bits of the fp32 value
            k += 1


def test_tsr_stage_matches_oracle_chain(eng_proc, lore_sd, proc_sd):
    """Whole stage (pre-process, DLA-34+DCN, decode, processor, rounding) in the fp32-class mode against the oracle
    chain on a 320 x 320 input: same cells in the same order, quads within 0.1 source px, logical locations equal except
    where the oracle's own value sits within 2e-3 of the .5 rounding boundary."""
    from oracle import lore_decode as od
    from oracle import lore_net, lore_pre, lore_processor
    from pdf_table_amd.synth_pages import make_page
    from pdf_table_amd.synth_weights import lore_dla34_state_dict
    from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
    sd = lore_dla34_state_dict(seed=21, hm_bias=(-3.6, -2.5))
    eng_proc.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(sd))
    page = make_page(3, 1024)[0]
    box = np.array([[200, 100, 560, 420]])
    cfg = LoreConfig()
    cfg.resolution = (320, 320)
    eng_proc.set_precision(L.PT_PRECISION_BF16X3)
    try:
        res = TsrStage(eng_proc, cfg)(torch.from_numpy(page[None]).cuda(), [box])[0][0]
    finally:
        eng_proc.set_precision(L.PT_PRECISION_BF16)
        eng_proc.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_sd))
    crop = np.ascontiguousarray(page[100:420, 200:560][:, :, ::-1])
    x, meta = lore_pre.lore_preprocess(crop, 320, 320)
    with torch.no_grad():
        z = lore_net.dlaseg_forward(sd, x)
        logi, ps, polys, results, raw = od.process_detect_output(z, meta, wiz_rev=True, vis_thresh=0.2, return_raw=True)
        logic, stacked = lore_processor.processor_forward(proc_sd, logi, None)
    n = logi.shape[1]
    assert n > 20
    # Peak / threshold decisions sit on fp32-class (1e-3 in logit) differences of the heat map, so cells are matched by
    # their quads, and a cell may be missing on either side only if the ORACLE's own decision was fragile: final score
    # within 3e-3 of vis_thresh, x0.4 demotion boundary, or a 3x3 neighbour within 3e-3 of the peak.
    sig = torch.sigmoid(z["hm"])[0, 0].numpy()
    inds = od.cell_decode.last_inds[:n]
    H, W = sig.shape

    def fragile(k):
        y, x0 = divmod(int(inds[k]), W)
        nb = [sig[yy, xx] for yy in range(max(y - 1, 0), min(y + 2, H)) for xx in range(max(x0 - 1, 0), min(x0 + 2, W))
              if (yy, xx) != (y, x0)]
        s = raw[k, 8]
        return abs(s - 0.2) < 3e-3 or abs(s / 0.4 - 0.2) < 3e-3 or abs(sig[y, x0] - 0.2) < 3e-3 or (sig[y, x0] - max(nb)) < 3e-3
    got_p = res["polygons"]
    used = np.zeros(len(got_p), bool)
    pairs = []
    for k in range(n):
        d = np.abs(got_p - polys[k]).max(axis=1)
        j = int(np.argmin(d))
        if d[j] <= 0.1 and not used[j]:      # 1e-3 of the wh/st head scale (~12) x 4.5 source px per map px, x2
            used[j] = True
            pairs.append((k, j))
        else:
            assert fragile(k), (k, raw[k, 8], d[j])
    assert len(pairs) >= n - 3 and (~used).sum() <= 3, (len(pairs), n, (~used).sum())
    # matched cells keep their relative order unless their scores are within 3e-3 of each other
    for (k1, j1), (k2, j2) in zip(pairs[:-1], pairs[1:]):
        assert j1 < j2 or abs(raw[k1, 8] - raw[k2, 8]) < 3e-3
    # logical locations: the processor attends over ALL cells of the table, so they are compared only when both sides
    # hold exactly the same cell set
    if len(pairs) == n == len(got_p) and all(k == j for k, j in pairs):
        ref_st = stacked[0].numpy()
        assert np.abs(res["stacked_axis"] - ref_st).max() <= 2 * TOL_REL * max(1.0, float(np.abs(ref_st).max()))
        frac = ref_st - np.floor(ref_st)
        safe = np.abs(frac - 0.5) > 5e-3
        assert np.array_equal(res["logi"][safe], od.process_logic_output(stacked)[0].numpy()[safe])


# ---- 'wireless' detector (ResNet-18 backbone) ------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 128, 192), (2, 64, 128), (1, 256, 256)])
def test_lore_wireless_net_x3_matches_oracle(eng, shape):
    from pdf_table_amd.synth_weights import lore_wireless_state_dict
    from pdf_table_amd.weights import pack_lore_wireless
    sd = lore_wireless_state_dict(seed=23)
    eng.load_weights(L.PT_MODEL_LORE_RESNET18, pack_lore_wireless(sd))
    n, H, W = shape
    g = torch.Generator().manual_seed(700 + W)
    x = torch.randn(n, 3, H, W, generator=g)
    with torch.no_grad():
        ref = lore_net.lore_wireless_forward(sd, x)
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        got = eng.tsr_forward_net(_x4(x, split=True).cuda(), wireless=True)
        torch.cuda.synchronize()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    assert _cmp(got, ref, f"lore wireless x3 {shape}") <= TOL_REL


def test_lore_wireless_net_matches_reference_golden_and_bf16(eng, golden_dir):
    from pdf_table_amd.synth_weights import lore_wireless_state_dict
    from pdf_table_amd.weights import pack_lore_wireless
    gold = np.load(os.path.join(golden_dir, "lore_wireless.npz"))
    sd = lore_wireless_state_dict(int(gold["seed"]))
    eng.load_weights(L.PT_MODEL_LORE_RESNET18, pack_lore_wireless(sd))
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        for tag in ("a", "b"):
            got = eng.tsr_forward_net(_x4(torch.from_numpy(gold[f"x_{tag}"]), split=True).cuda(), wireless=True)
            for k in HEADS:
                gk = got[k].cpu().permute(0, 3, 1, 2).numpy()
                gk = gk[:, ::8] if gk.shape[1] == 256 else gk
                r = gold[f"{k}_{tag}"]
                assert np.abs(gk - r).max() / max(1.0, np.abs(r).max()) <= TOL_REL, (tag, k)
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    x = torch.from_numpy(gold["x_a"]).to(torch.bfloat16).float()
    with torch.no_grad():
        ref = lore_net.lore_wireless_forward(sd, x)
    got = eng.tsr_forward_net(_x4(x).cuda(), wireless=True)
    assert _cmp(got, ref, "lore wireless bf16") <= 0.1


_FUSED_SCRIPT = r'''
import sys, numpy as np, torch
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_weights import lore_dla34_state_dict
from pdf_table_amd.weights import pack_lore_dla34
mode, wiz_rev, out = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
eng = HipEngine(0)
# head biases near the thresholds: hundreds of cells and corners per table on noise input
eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_dla34_state_dict(seed=2, hm_bias=(-1.2, -0.6))))
eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
g = torch.Generator().manual_seed(77)
n, H, W = 3, 256, 320
x = torch.randn(n, H, W, 3, generator=g) * 0.7
x4 = torch.zeros(n, H, W, 8 if mode == "bf16x3" else 4)
hi = x.to(torch.bfloat16).float()
x4[..., :3] = hi
if mode == "bf16x3":
    x4[..., 4:7] = (x - hi).to(torch.bfloat16).float()
xd = x4.to(torch.bfloat16).cuda()
heads = eng.tsr_forward_net(xd)
c0, d0, l0 = eng.tsr_decode(heads, wiz_rev=wiz_rev, vis_thresh=0.2, sync=True)
c1, d1, l1 = eng.tsr_forward_decode(xd, wiz_rev=wiz_rev, vis_thresh=0.2, sync=True)
np.savez(out, c0=c0, c1=c1, d0=d0.cpu().numpy(), d1=d1.cpu().numpy(), l0=l0.cpu().numpy(), l1=l1.cpu().numpy())
'''


def _run_fused(tmp_path, mode, wiz_rev, variant):
    import os
    import subprocess
    import sys
    out = str(tmp_path / f"fused_{mode}_{int(wiz_rev)}_{variant}.npz")
    e = dict(os.environ)
    if variant is not None:
        e["PT_CONV_VARIANT"] = str(variant)
    e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _FUSED_SCRIPT, mode, "1" if wiz_rev else "0", out], check=True, env=e, timeout=300)
    return np.load(out)


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("wiz_rev", [True, False])
def test_forward_decode_fused_is_bit_identical(tmp_path, mode, wiz_rev):
    """pt_tsr_forward_decode (ax / cr heads evaluated only on 3x3 patches around the positions the decode reads) gives
    exactly the outputs of pt_tsr_forward_net + pt_tsr_decode -- counts, boxes, scores and the 256 logic features -- when
    both run one conv kernel family (PT_CONV_VARIANT=0: the mosaic convs always use the register-staged kernel)"""
    r = _run_fused(tmp_path, mode, wiz_rev, 0)
    c0, c1 = r["c0"], r["c1"]
    print("fused decode: cells above vis_thresh per table", c0.tolist())
    assert np.array_equal(c0, c1) and c0.sum() > 0, (c0, c1)
    for b in range(len(c0)):         # rows beyond a table's count are unspecified in both paths
        k = int(c0[b])
        assert np.array_equal(r["d0"][b, :max(k, 1)], r["d1"][b, :max(k, 1)])
        assert np.array_equal(r["l0"][b, :k], r["l1"][b, :k]), b


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_forward_decode_fused_default_dispatch(tmp_path, mode):
    """with the default conv dispatch the dense maps' 3x3 head convs run on the DMA kernel, which sums K in another order
    than the register-staged kernel the mosaics use: the two paths agree to fp32 summation noise (after one bf16 rounding
    of the hidden layer in bf16 mode) -- scores come from the dense `hm` map in both and are identical"""
    r = _run_fused(tmp_path, mode, True, None)
    assert np.array_equal(r["c0"], r["c1"]) and r["c0"].sum() > 0
    tol_px, tol_f = (0.1, 2e-2) if mode == "bf16" else (1e-3, 1e-4)
    for b in range(len(r["c0"])):
        k = int(r["c0"][b])
        assert np.array_equal(r["d0"][b, :k, 8], r["d1"][b, :k, 8])
        dpx = float(np.abs(r["d0"][b, :k, :8] - r["d1"][b, :k, :8]).max())
        scale = float(np.abs(r["l0"][b, :k]).max())
        d = float(np.abs(r["l0"][b, :k] - r["l1"][b, :k]).max())
        print(f"fused vs dense, default dispatch, {mode}, table {b}: max|dbox| {dpx:.3e} px, max|dfeat| {d:.3e} (scale {scale:.1f})")
        assert dpx <= tol_px, (b, dpx)
        assert d <= tol_f * max(1.0, scale), (b, d, scale)


def test_thin_stem_kernel_equals_general_kernel(tmp_path):
    """bf16 mode: conv_stem7x7_thin_kernel (DLA base_layer) gives bit-identical network outputs to the general stem kernel
    (PT_STEM_THIN=0 in a child process: the switch is read once); odd tile counts in both directions"""
    import os
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_weights import lore_dla34_state_dict
from pdf_table_amd.weights import pack_lore_dla34
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_dla34_state_dict(seed=2), x3=False))
g = torch.Generator().manual_seed(5)
x = torch.zeros(2, 288, 352, 4)
x[..., :3] = torch.randn(2, 288, 352, 3, generator=g)
heads = eng.tsr_forward_net(x.to(torch.bfloat16).cuda())
np.savez(sys.argv[1], **{k: v.float().cpu().numpy() for k, v in heads.items()})
'''
    outs = []
    for tag, env in (("thin", {}), ("general", {"PT_STEM_THIN": "0"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e["PT_CONV_VARIANT"] = "0"
        e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=e, timeout=300)
        outs.append(np.load(out))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_up_sampler_block_kernel_equals_general_kernel(eng, mode):
    """dwconvt_up2_add_kernel (2x2 output block per thread, weights in LDS) gives the head maps of the general
    depthwise-transposed-conv kernel bit for bit (PT_DWCONVT2 is read at every call); odd map sizes"""
    import os
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 3, 160, 224, generator=g) * 0.7
        xd = _x4(x, split=mode == "bf16x3").cuda()
        a = {k: t.cpu().numpy() for k, t in eng.tsr_forward_net(xd).items()}
        os.environ["PT_DWCONVT2"] = "0"
        try:
            b = {k: t.cpu().numpy() for k, t in eng.tsr_forward_net(xd).items()}
        finally:
            del os.environ["PT_DWCONVT2"]
        assert set(a) == set(b) and len(a) == 6
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        assert float(np.abs(a["hm"]).max()) > 0
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


@pytest.mark.parametrize("hw", [(160, 224), (96, 64), (1024, 1024), (32, 1056)])
def test_thin_chain_equals_three_launches(eng, hw):
    """dla_thin_chain_kernel (base_layer -> level0 -> level1 in one launch, the two full-resolution maps in LDS; bf16 mode; PT_DLA_CHAIN=1) gives the
    head maps of the three stand-alone launches (PT_DLA_CHAIN=0) bit for bit (the switch is read at every call): maps smaller than a tile, maps whose
    size is not a multiple of the 8 x 30 level-1 tile (partial tiles, zero padding of the intermediate maps at every border), the bench's 1024^2.
    The default kernel (dla_thin_chain16_kernel: the two 16-channel levels on the 16 x 16 x 32 MFMA, taps paired into K = 32 blocks) sums K in another
    association.  A few roundings of stored 16-bit values flip and the network behind
    (34 layers, deformable sampling) spreads them to its own bf16 noise level -- so the claim is made against the BF16X3 evaluation of the same net:
    the default kernel's head maps are as close to it as the three launches' are."""
    import os
    H, W = hw
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(1 if H * W > 500000 else 2, 3, H, W, generator=g) * 0.7
    xd = _x4(x).cuda()
    outs = {}
    for sw in ("2", "1", "0"):
        os.environ["PT_DLA_CHAIN"] = sw
        try:
            outs[sw] = {k: t.cpu().numpy() for k, t in eng.tsr_forward_net(xd).items()}
        finally:
            del os.environ["PT_DLA_CHAIN"]
    a, b, c = outs["1"], outs["0"], outs["2"]
    assert set(a) == set(b) == set(c) and len(a) == 6
    for k in a:
        assert np.array_equal(a[k], b[k]), (hw, k, float(np.abs(a[k].astype(np.float64) - b[k]).max()))
    assert float(np.abs(a["hm"]).max()) > 0
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        ref = {k: t.cpu().numpy().astype(np.float64) for k, t in eng.tsr_forward_net(_x4(x, True).cuda()).items()}
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    for k in a:
        scale = float(np.abs(ref[k]).max())
        d1, d2 = np.abs(b[k] - ref[k]), np.abs(c[k] - ref[k])
        print(f"thin chain {hw} {k}: |three launches - x3| max {d1.max() / scale:.4f} mean {d1.mean() / scale:.5f}; |16x16x32 chain - x3| max {d2.max() / scale:.4f} "
              f"mean {d2.mean() / scale:.5f}; identical to the three launches: {np.array_equal(b[k], c[k])}")
        # (the smallest case has 1 536 values per head: the ratio of two such means scatters by tens of per cent; on 1024^2 it reads 0.98 - 1.00)
        assert d2.mean() <= 1.5 * d1.mean() + 2e-4 * scale and d2.max() <= 2.0 * d1.max() + 1e-3 * scale, (hw, k)
