"""Full-size (BASELINE.json: 1024x1024 pages) checks through size-independent properties: determinism, batch
invariance, bitmap/threshold consistency, integer post-processing vs the oracle on the engine's own map, ragged and
empty inputs."""
import numpy as np
import pytest
import torch

from oracle import db_post
from pdf_table_amd import engine as E
from pdf_table_amd import lib as L
from pdf_table_amd import rec_stage as R
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import crnn_state_dict, db_resnet18_state_dict
from pdf_table_amd.weights import pack_crnn, pack_db_resnet18

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(db_resnet18_state_dict(seed=0), x3=False))
    e.load_weights(L.PT_MODEL_CRNN, pack_crnn(crnn_state_dict(seed=1), x3=False))
    yield e
    e.close()


@pytest.fixture(scope="module")
def pages():
    return [make_page(i) for i in range(3)]


def test_det_fullsize_deterministic_and_batch_invariant(eng, pages):
    imgs = np.stack([p[0] for p in pages])
    a, bm_a = eng.det_forward(torch.from_numpy(imgs).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
    b, bm_b = eng.det_forward(torch.from_numpy(imgs).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
    assert a.shape == (3, 960, 960)
    assert torch.equal(a, b) and torch.equal(bm_a, bm_b)                       # run-to-run identical
    rev, _ = eng.det_forward(torch.from_numpy(imgs[::-1].copy()).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
    assert torch.equal(rev.flip(0), a)                                          # a page's map ignores its neighbours
    one, _ = eng.det_forward(torch.from_numpy(imgs[1:2]).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
    assert torch.equal(one[0], a[1])                                            # ... and the batch size
    p = a.cpu().numpy()
    bits = ((bm_a.cpu().numpy().view(np.uint32)[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(p.shape).astype(bool)
    np.testing.assert_array_equal(bits, p > 0.3)
    assert np.isfinite(p).all() and p.min() >= 0 and p.max() <= 1


def test_det_fullsize_integer_post_matches_oracle(eng, pages):
    img = pages[0][0]
    prob, bm = eng.det_forward(torch.from_numpy(img[None]).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
    p = prob[0].cpu().numpy()
    bmh = bm[0].cpu().numpy().view(np.uint32)
    cand, _ = E.db_candidates(bmh, 1000, 3.0)
    ref_cand = db_post.candidates_from_bitmap(p > 0.3, 1000, 3)
    assert len(cand) == len(ref_cand) > 10
    for (pts, _), c in zip(ref_cand, cand):
        np.testing.assert_array_equal(pts.reshape(-1), c)
    cb = np.concatenate([np.zeros((len(cand), 1), np.float32), cand], 1)
    sc = eng.det_box_scores(prob, torch.from_numpy(cb).cuda()).cpu().numpy()
    ref_sc = np.array([db_post.box_score_fast(p, c.reshape(4, 2)) for c in cand], np.float32)
    np.testing.assert_allclose(sc, ref_sc, rtol=0, atol=1e-6)
    out, _ = E.db_finalize(cand, sc, p.shape, img.shape[:2], 0.05, 1.5, 3.0)      # low gate: keep many boxes
    ref, _ = db_post.boxes_from_bitmap(p, p > 0.3, img.shape[1], img.shape[0], 0.05, 1.5, scores_override=sc)
    np.testing.assert_array_equal(out.reshape(-1, 4, 2), ref.astype(np.int32))
    assert len(out) > 10 and out.min() >= 0 and out.max() <= 1024


def test_det_ragged_page_sizes(eng):
    """pages of different sizes are grouped by shape by the task layer; every group keeps its own plan"""
    from pdf_table_amd.det_stage import DetConfig, DetStage
    st = DetStage(eng, DetConfig("db_pp"))
    for (h, w) in ((1024, 1024), (640, 480), (37, 53), (2048, 1536)):
        img = np.random.default_rng(h).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        nh, nw = eng.det_plan(h, w, L.PT_DET_PRE_DB_PP)
        assert nh % 32 == 0 and nw % 32 == 0 and max(nh, nw) <= 960
        boxes = st(torch.from_numpy(img).cuda())
        assert len(boxes) == 1 and boxes[0].shape[1:] == (8,)


def test_rec_order_invariance_and_empty(eng, pages):
    img, meta = pages[1]
    l = meta["lines"].astype(np.float64)
    quads = np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1)
    st = R.RecStage(eng)
    dev = torch.from_numpy(img[None]).cuda()
    ids, _ = st.ids(dev, [quads])
    perm = np.random.default_rng(0).permutation(len(quads))
    ids_p, _ = st.ids(dev, [quads[perm]])
    assert torch.equal(ids_p, ids[torch.from_numpy(perm).cuda()])               # a line's ids ignore its neighbours
    ids2, _ = st.ids(dev, [quads])
    assert torch.equal(ids, ids2)
    assert ids.min() >= 0 and ids.max() < L.PT_REC_NCLS                         # padded classes can never win
    empty, lines = st.ids(dev, [np.zeros((0, 8))])
    assert empty.shape == (0, L.PT_REC_T) and len(lines) == 0
    assert st(dev, [np.zeros((0, 8))]) == [[]]
    # degenerate quad (zero area): crop is empty -> all-padding input, still decodes without error
    deg, _ = st.ids(dev, [np.array([[10, 10, 10, 10, 10, 10, 10, 10]], np.float64)])
    assert deg.shape == (1, L.PT_REC_T)


# ---- table structure and layout at BASELINE.json's sizes: batch invariance -------------------------------------------
@pytest.fixture(scope="module")
def eng_tl():
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.synth_weights import lore_dla34_state_dict, picodet_state_dict
    from pdf_table_amd.weights import pack_lore_dla34, pack_picodet
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_dla34_state_dict(seed=2, hm_bias=(-1.2, -0.6)), x3=False))
    e.load_weights(L.PT_MODEL_PICODET, pack_picodet(picodet_state_dict(seed=4, num_classes=5), 5, x3=False))
    yield e
    e.close()


def test_tsr_fullsize_batch_invariant_and_deterministic(eng_tl):
    """1024x1024 table images (the Lore input size of BASELINE.json's configs): a table's decoded cells, boxes and the 256
    logic features do not depend on which other tables share its batch, nor on the run (bit-exact: every kernel tiles per
    image or per patch; the sparse heads' mosaics pack patches of different tables side by side)"""
    g = torch.Generator().manual_seed(31)
    n = 5
    x4 = torch.zeros(n, 1024, 1024, 4)
    x4[..., :3] = torch.randn(n, 1024, 1024, 3, generator=g) * 0.7
    x4 = x4.to(torch.bfloat16).cuda()
    c0, d0, l0 = eng_tl.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=True)
    c1, d1, l1 = eng_tl.tsr_forward_decode(x4, wiz_rev=True, vis_thresh=0.2, sync=True)
    assert c0.sum() > 0 and np.array_equal(c0, c1)
    d0, l0, d1, l1 = d0.cpu().numpy(), l0.cpu().numpy(), d1.cpu().numpy(), l1.cpu().numpy()
    for t in range(n):
        k = int(c0[t])
        assert np.array_equal(d0[t, :k], d1[t, :k]) and np.array_equal(l0[t, :k], l1[t, :k])
    for t in (0, 3):
        cs, ds, ls = eng_tl.tsr_forward_decode(x4[t:t + 1].contiguous(), wiz_rev=True, vis_thresh=0.2, sync=True)
        k = int(c0[t])
        assert int(cs[0]) == k, (t, cs, c0)
        assert np.array_equal(ds.cpu().numpy()[0, :k], d0[t, :k])
        assert np.array_equal(ls.cpu().numpy()[0, :k], l0[t, :k])
    print("tsr full size: cells per table", c0.tolist())


def test_layout_fullsize_batch_invariant(eng_tl, pages):
    """64-page batches of 1024x1024 pages through PicoDet at 800x608: a page's candidate records (the anchors above the
    score floor with their raw head values) are the same set whether the page runs alone or inside the batch"""
    imgs = np.stack([pages[i % 3][0] for i in range(64)])
    imgs[5] = imgs[5][::-1].copy()                   # one page unlike the others
    dev = torch.from_numpy(imgs).cuda()
    counts, cands = eng_tl.layout_forward(dev, thr_lo=0.3)
    counts = counts.cpu().numpy()
    cands = cands.cpu().numpy()
    assert counts.max() <= cands.shape[1]

    def records(c, k):
        r = c[:k, :2 + L.PT_LAYOUT_HEAD_CS]          # (level, anchor, 40 head values); the rest of a record is padding
        key = r[:, :2].copy().view(np.int32)
        order = np.lexsort((key[:, 1], key[:, 0]))
        return r[order]

    for p in (0, 5, 63):
        c1, r1 = eng_tl.layout_forward(dev[p:p + 1].contiguous(), thr_lo=0.3)
        k = int(c1.cpu().numpy()[0])
        assert k == int(counts[p]), (p, k, counts[p])
        assert np.array_equal(records(r1.cpu().numpy()[0], k), records(cands[p], k))
    assert np.array_equal(records(cands[0], int(counts[0])), records(cands[3], int(counts[3])))    # same page, same records
    print("layout full size: candidates per page", counts[:6].tolist())


def test_rec_one_launch_lstm_equals_two_launches(eng):
    """a step's worth of text lines (> 4096: the recognizer's LSTM then runs 192-line clusters in one launch) gives the ids
    and winning logits of the 128-line clusters in two launches, bit for bit (PT_LSTM_MI is read at every call)"""
    import os
    g = torch.Generator().manual_seed(17)
    gray = (torch.rand(4300, L.PT_REC_H, L.PT_REC_W, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    ids0, mx0 = eng.rec_forward_net(gray)
    os.environ["PT_LSTM_MI"] = "2"
    try:
        ids1, mx1 = eng.rec_forward_net(gray)
    finally:
        del os.environ["PT_LSTM_MI"]
    torch.cuda.synchronize()
    eng.check()
    assert np.array_equal(ids0.cpu().numpy(), ids1.cpu().numpy())
    assert np.array_equal(mx0.cpu().numpy(), mx1.cpu().numpy())
    assert len(np.unique(ids0.cpu().numpy())) > 20


# ---- oracle parity at BASELINE.json's sizes, BOTH arithmetic modes -----------------------------------------------------
# PT_PRECISION_BF16X3 must meet the north-star tolerance (1e-3 on float logits, relative to the logit scale; ids exact
# outside the oracle's own ties); PT_PRECISION_BF16 -- the mode bench.py's headline number runs in -- has its drift
# measured here, printed, and bounded.  One 1024x1024 page -> 960x960 DB map, one 1024x1024 Lore table, one 800x608
# PicoDet page, 64 CRNN lines: seconds of fp32 oracle each.
X3_TOL = 1e-3


def _x4(x, split, dtype=torch.bfloat16):
    n, _, H, W = x.shape
    nhwc = x.permute(0, 2, 3, 1)
    if not split:
        x4 = torch.zeros(n, H, W, 4)
        x4[..., :3] = nhwc
        return x4.to(dtype)
    hi = nhwc.to(torch.bfloat16).float()
    lo = (nhwc - hi).to(torch.bfloat16).float()
    x8 = torch.zeros(n, H, W, 8)
    x8[..., :3] = hi
    x8[..., 4:7] = lo
    return x8.to(torch.bfloat16)


@pytest.fixture(scope="module")
def eng_par():
    """one engine with every net loaded with the (hi, lo) tiles too"""
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.synth_weights import lore_dla34_state_dict, picodet_state_dict
    from pdf_table_amd.weights import pack_lore_dla34, pack_picodet
    e = HipEngine(0)
    # Lore: dcn_gain=0.02 (DCN offsets of ~0.07 px instead of ~0.3 px).  A RANDOM DLA-34 + 16 stacked DCNs is chaotic in its
    # offsets: with the default gain the fp32 oracle ITSELF moves by 1e-4 of the head scale when its input is perturbed by
    # 1e-7 relative (one fp32 ulp) and by 1.2e-3 for 1e-5 (tools/lore_conditioning.py, 512x512) -- no arithmetic can then
    # agree with it to 1e-3 at full size (measured: bf16x3 2e-3 .. 6e-3, bf16 0.4, growing with the map size).  With 0.02 the
    # same perturbations move the oracle by 1e-5 / 6e-5, the conditioning of a trained net, and the comparison is meaningful.
    # THE checkpoint set bench.py times and the end-to-end fixture runs on (synth_weights.conditioned_state_dicts): what is asserted here at BASELINE
    # sizes is what is timed -- text-signal detector, fitted CRNN classifier, fitted layout head, Lore with dcn_gain = 0.02
    from pdf_table_amd.synth_weights import conditioned_state_dicts
    sds = dict(conditioned_state_dicts())
    e.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sds["db"]))
    e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sds["crnn"]))
    e.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(sds["lore"]))
    e.load_weights(L.PT_MODEL_PICODET, pack_picodet(sds["pico"], 5))
    # PT_PRECISION_F16 (single-pass IEEE half, the reference's own GPU arithmetic): a second engine, the SAME state dicts packed as fp16 tiles
    e16 = HipEngine(0)
    e16.set_precision(L.PT_PRECISION_F16)
    e16.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sds["db"], fmt="f16"))
    e16.load_weights(L.PT_MODEL_CRNN, pack_crnn(sds["crnn"], fmt="f16"))
    e16.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(sds["lore"], fmt="f16"))
    e16.load_weights(L.PT_MODEL_PICODET, pack_picodet(sds["pico"], 5, fmt="f16"))
    sds["_f16_engine"] = e16
    yield e, sds
    e.close()
    e16.close()


def _pick(eng_par, mode):
    """(engine, state dicts) of a mode: the f16 engine has its own fp16 blobs, the three bf16-family modes share one engine"""
    e, sds = eng_par
    return (sds["_f16_engine"] if mode == "f16" else e), sds


def _mode(eng, mode):
    if mode == "f16":
        assert eng.precision == L.PT_PRECISION_F16
        return
    if eng.precision == L.PT_PRECISION_F16:
        return          # the f16 engine stays in its precision (its blobs are fp16)
    eng.set_precision({"bf16x3": L.PT_PRECISION_BF16X3, "f16x2": L.PT_PRECISION_F16X2}.get(mode, L.PT_PRECISION_BF16))


def _adt(mode):
    return torch.float16 if mode == "f16" else torch.bfloat16


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16", "bf16"])
def test_fullsize_det_oracle_parity(eng_par, pages, mode):
    """1024x1024 page -> db_pp pre-process (bit-exact with the oracle's) -> DB-ResNet18 at 960x960 vs the fp32 oracle"""
    from oracle import db_net, db_pre
    eng, sds = _pick(eng_par, mode)
    img = pages[0][0]
    chw, _ = db_pre.preprocess_db_pp(img)
    with torch.no_grad():
        ref = db_net.db_forward_fp32(sds["db"], torch.from_numpy(np.ascontiguousarray(chw))[None], return_logits=True)[0, 0]
    _mode(eng, mode)
    try:
        x = eng.det_preprocess(torch.from_numpy(img[None]).cuda(), L.PT_DET_PRE_DB_PP)
        prob, logits = eng.det_forward_net(x, want_logits=True)
        torch.cuda.synchronize()
    finally:
        _mode(eng, "bf16")
    assert tuple(logits.shape[1:]) == (960, 960)
    scale = max(1.0, ref.abs().max().item())
    dl = (logits[0].cpu() - ref).abs().max().item()
    dp = (prob[0].cpu() - torch.sigmoid(ref)).abs().max().item()
    flips = int(((prob[0].cpu() > 0.3) != (torch.sigmoid(ref) > 0.3)).sum())
    print(f"FULLSIZE det 960x960 {mode}: max|dlogit|={dl:.3e} = {dl / scale:.3e} of scale {scale:.1f}; max|dprob|={dp:.3e}; "
          f"{flips} of 921600 bitmap pixels differ")
    if mode == "f16x2":
        # the two-pass experiment: (hi, lo) activation pairs x SINGLE fp16 weights.  Every layer is exact against its
        # fp16-rounded weights (5e-6 per layer), but rounding 21 layers' weights to 11 bits moves THIS random-init net's logits by 1.2e-3 of
        # their scale (the CPU emulation tools/x2_emulation.py reads 1.4e-3 on the same page; 1.5e-4 on the text-signal checkpoint): outside
        # the 1e-3 contract, so BF16X3 stays the tolerance mode and this bound only records the drift
        assert dl <= 2e-3 * scale and dp <= 6e-3
    elif mode == "bf16x3":
        assert dl <= X3_TOL * scale and dp <= X3_TOL
        near = (torch.sigmoid(ref) - 0.3).abs() <= X3_TOL          # a bitmap pixel may differ only on the threshold itself
        assert bool((((prob[0].cpu() > 0.3) != (torch.sigmoid(ref) > 0.3)) & ~near).sum() == 0)
    elif mode == "f16":
        # single-pass IEEE half: 11 significant bits where bf16 keeps 8 -- the CPU emulation of round 2 predicted 2.9e-3 of the logit scale
        # against bf16's 2.6e-2 (DESIGN numerics)
        assert dl <= 0.008 * scale
    else:
        assert dl <= 0.06 * scale


@pytest.mark.parametrize("mode", ["bf16x3", "f16", "bf16"])
def test_fullsize_lore_oracle_parity(eng_par, pages, mode):
    """one 1024x1024 table (a warped crop of a synthetic page) through DLA-34 + 16 DCN + 6 heads vs the fp32 oracle"""
    from oracle import lore_net, lore_pre
    eng, sds = _pick(eng_par, mode)
    img, meta = pages[0]
    x1, y1, x2, y2 = (int(v) for v in meta["tables"].reshape(-1, 4)[0])
    xo, _ = lore_pre.lore_preprocess(np.ascontiguousarray(img[y1:y2, x1:x2][:, :, ::-1]), 1024, 1024)
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(sds["lore"], xo)
    _mode(eng, mode)
    try:
        got = eng.tsr_forward_net(_x4(xo, split=mode == "bf16x3", dtype=_adt(mode)).cuda())
        torch.cuda.synchronize()
    finally:
        _mode(eng, "bf16")
    worst = 0.0
    for k in ref:
        g = got[k].cpu().permute(0, 3, 1, 2)
        rel = (g - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item())
        worst = max(worst, rel)
        print(f"FULLSIZE lore 1024x1024 {mode} head {k}: rel max err {rel:.3e} (scale {ref[k].abs().max().item():.2f})")
    assert worst <= (X3_TOL if mode == "bf16x3" else 0.03 if mode == "f16" else 0.2)      # bf16: measured 0.04 .. 0.13 of the head scale (reg, the smallest head)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])      # bf16: a bound on the mean error and a loose one on the maximum keep the stressier net under a test
def test_fullsize_lore_default_gain_drift(pages, mode):
    """The SAME table through the UNCONDITIONED synthetic net (lore_dla34_state_dict(seed=2), dcn_gain = 0.1: offsets of ~0.3 px -- what bench.py timed
    until round 4; it now times the conditioned net asserted above): recorded and bounded.  The oracle itself is ill-conditioned on this random net (see eng_par: a 1e-5 relative input perturbation moves
    the fp32 oracle by 1.2e-3 of the head scale at 512 x 512), so the BF16X3 bound here is NOT the 1e-3 contract -- that one is asserted on the
    dcn_gain = 0.02 net above and, operator by operator with multi-pixel offsets, in tests/test_gpu_dcn_op.py -- but a recorded drift."""
    from oracle import lore_net, lore_pre
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.synth_weights import lore_dla34_state_dict
    from pdf_table_amd.weights import pack_lore_dla34
    sd = lore_dla34_state_dict(seed=2)
    img, meta = pages[0]
    x1, y1, x2, y2 = (int(v) for v in meta["tables"].reshape(-1, 4)[0])
    xo, _ = lore_pre.lore_preprocess(np.ascontiguousarray(img[y1:y2, x1:x2][:, :, ::-1]), 1024, 1024)
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(sd, xo)
    eng = HipEngine(0)
    try:
        eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(sd))
        _mode(eng, mode)
        got = eng.tsr_forward_net(_x4(xo, split=mode == "bf16x3").cuda())
        torch.cuda.synchronize()
        got = {k: got[k].cpu().permute(0, 3, 1, 2) for k in ref}
    finally:
        eng.close()
    worst = worst_mean = 0.0
    for k in ref:
        scale = max(1.0, ref[k].abs().max().item())
        rel = (got[k] - ref[k]).abs().max().item() / scale
        rel_mean = (got[k] - ref[k]).abs().mean().item() / scale
        worst, worst_mean = max(worst, rel), max(worst_mean, rel_mean)
        print(f"FULLSIZE lore 1024x1024 DEFAULT GAIN (dcn_gain 0.1) {mode} head {k}: rel max err {rel:.3e}, mean {rel_mean:.3e} (scale {ref[k].abs().max().item():.2f})")
    # bf16: the MEAN error is the stable statistic of this chaotic net (r06: 0.013 .. 0.047 of the head scale, the same to three digits whichever kernels
    # compute the thin levels); the maximum is the tail of one noise realisation -- 0.56 with the 32 x 32 x 16 thin chain, 0.75 with the 16 x 16 x 32 one on
    # the `reg` head, 0.45 against 0.36 on `st` the other way round
    assert worst <= (2e-2 if mode == "bf16x3" else 1.0)      # r03 measured at this gain: bf16x3 2e-3 .. 6e-3, bf16 0.4
    assert worst_mean <= (2e-3 if mode == "bf16x3" else 0.07)


@pytest.mark.parametrize("mode", ["bf16x3", "f16", "bf16"])
def test_fullsize_picodet_oracle_parity(eng_par, pages, mode):
    """one 1024x1024 page -> 800x608 PicoDet input (the oracle's pre-process) -> LCNet + CSP-PAN + PicoHead vs the oracle"""
    from oracle import picodet as op
    eng, sds = _pick(eng_par, mode)
    xl, _ = op.picodet_preprocess(pages[1][0])
    x = torch.from_numpy(xl)[None]
    with torch.no_grad():
        sc, bx = op.picodet_forward(sds["pico"], x, 5)
    _mode(eng, mode)
    try:
        heads = eng.layout_forward_net(_x4(x, split=mode == "bf16x3", dtype=_adt(mode)).cuda())
        torch.cuda.synchronize()
    finally:
        _mode(eng, "bf16")
    worst_s = worst_b = 0.0
    for l in range(4):
        h = heads[l].cpu()
        worst_s = max(worst_s, (torch.sigmoid(h[..., :5]) - sc[l]).abs().max().item())
        worst_b = max(worst_b, (h[..., 5:37] - bx[l]).abs().max().item() / max(1.0, bx[l].abs().max().item()))
    print(f"FULLSIZE picodet 800x608 {mode}: max|dscore|={worst_s:.3e}, box-logit rel max err {worst_b:.3e}")
    if mode == "bf16x3":
        assert worst_s <= X3_TOL and worst_b <= X3_TOL
    else:
        assert worst_b <= (0.015 if mode == "f16" else 0.1)


@pytest.mark.parametrize("mode", ["bf16x3", "f16", "bf16"])
def test_fullsize_crnn_64_lines_oracle_parity(eng_par, pages, mode):
    """64 text lines of a synthetic page at 32x640 through the CRNN vs the fp32 oracle: winning logit and token ids"""
    from oracle import crnn as ocrnn
    eng, sds = _pick(eng_par, mode)
    img, meta = pages[1]
    l = meta["lines"].astype(np.float64)
    quads = np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1)
    quads = np.concatenate([quads, quads + 3.0])[:64]
    xs = torch.cat([ocrnn.rec_preprocess(ocrnn.crop_image(img, ocrnn.order_point(q))) for q in quads])
    with torch.no_grad():
        logits = ocrnn.crnn_forward_fp32(sds["crnn"], xs, native_lstm=True)
    top2 = torch.topk(logits, 2, dim=-1)
    gray = xs[:, 0] * 0.2989 + xs[:, 1] * 0.5870 + xs[:, 2] * 0.1140
    _mode(eng, mode)
    try:
        if mode == "bf16x3":
            hi = gray.to(torch.bfloat16)
            g = torch.stack([hi, (gray - hi.float()).to(torch.bfloat16)], -1).contiguous()
        else:
            g = gray.to(_adt(mode)).contiguous()
        ids, mx = eng.rec_forward_net(g.cuda())
        torch.cuda.synchronize()
        eng.check()
    finally:
        _mode(eng, "bf16")
    ids, mx = ids.cpu(), mx.cpu()
    scale = logits.abs().max().item()
    dmax = (mx - top2.values[..., 0]).abs().max().item()
    margin = top2.values[..., 0] - top2.values[..., 1]
    diff = ids != top2.indices[..., 0]
    print(f"FULLSIZE crnn 64 lines {mode}: max|d winning logit|={dmax:.3e} = {dmax / scale:.3e} of scale {scale:.1f}; "
          f"{int(diff.sum())} of {ids.numel()} token ids differ (largest oracle margin among them "
          f"{margin[diff].max().item() if diff.any() else 0.0:.3e})")
    if mode == "bf16x3":
        assert dmax <= X3_TOL * scale
        assert bool((margin[diff] <= 2 * X3_TOL * scale).all())        # ids exact outside the oracle's own ties
        assert float(diff.float().mean()) < 0.01
    elif mode == "f16":
        assert dmax <= 0.008 * scale and bool((margin[diff] <= 0.016 * scale).all())
    else:
        assert dmax <= 0.06 * scale and bool((margin[diff] <= 0.12 * scale).all())


@pytest.mark.parametrize("mode", ["bf16x3", "f16", "bf16"])
def test_fullsize_convnext_vit_64_lines_oracle_parity(eng_par, pages, mode):
    """64 text lines of a synthetic 1024x1024 page through the WHOLE ConvNextViT path on the device -- perspective crop out of
    the resident page, keep-ratio resize to 32x804, three chunks (the all-padding ones shared), ConvNext + ViT, stitching,
    classifier, arg-max -- against the oracle run line by line like the reference (crop_image, chunking pre-processor, fp32
    network): winning logit and token ids"""
    from oracle import convnext_vit as ocv
    from oracle import crnn as ocrnn
    from pdf_table_amd import rec_stage as R
    from pdf_table_amd.synth_weights import convnext_vit_state_dict
    from pdf_table_amd.weights import pack_convnext_vit
    eng, _ = _pick(eng_par, mode)
    sd = convnext_vit_state_dict(seed=3)
    eng.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(sd, fmt=eng.weight_fmt))
    img, meta = pages[1]
    l = meta["lines"].astype(np.float64)
    quads = np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1)
    quads = np.concatenate([quads, quads + 3.0])[:60]
    wide = [(40, 100, 1000, 130), (40, 200, 600, 230), (40, 300, 340, 330), (500, 400, 1010, 420)]      # two and three chunks of text
    quads = np.concatenate([quads, np.array([[x0, y0, x1, y0, x1, y1, x0, y1] for x0, y0, x1, y1 in wide], np.float64)])
    with torch.no_grad():
        logits = torch.cat([ocv.convnext_vit_forward_fp32(sd, ocv.chunk_preprocess(ocrnn.crop_image(img, ocrnn.order_point(q))))
                            for q in quads])
    top2 = torch.topk(logits, 2, dim=-1)
    lines = R.build_lines([quads])
    _mode(eng, mode)
    try:
        ids, mx = eng.rec_cvit_forward(torch.from_numpy(img[None]).cuda(), lines)
        torch.cuda.synchronize()
        eng.check()
    finally:
        _mode(eng, "bf16")
    ids, mx = ids.cpu().long(), mx.cpu()
    scale = logits.abs().max().item()
    dmax = (mx - top2.values[..., 0]).abs().max().item()
    margin = top2.values[..., 0] - top2.values[..., 1]
    diff = ids != top2.indices[..., 0]
    tw = np.minimum(804, (32 * lines["crop_w"] / np.maximum(lines["crop_h"], 1)).astype(int))
    print(f"FULLSIZE convnext-vit 64 lines {mode}: max|d winning logit|={dmax:.3e} = {dmax / scale:.3e} of scale {scale:.1f}; "
          f"{int(diff.sum())} of {ids.numel()} token ids differ (largest oracle margin among them "
          f"{margin[diff].max().item() if diff.any() else 0.0:.3e}); {float(((tw > 0).astype(int) + (tw > 252) + (tw > 504)).mean()):.2f} chunks "
          f"with text per line")
    if mode == "bf16x3":
        assert dmax <= X3_TOL * scale
        assert bool((margin[diff] <= 2 * X3_TOL * scale).all())        # ids exact outside the oracle's own ties
        assert float(diff.float().mean()) < 0.01
    elif mode == "f16":
        assert dmax <= 0.008 * scale and bool((margin[diff] <= 0.016 * scale).all())
    else:
        assert dmax <= 0.06 * scale and bool((margin[diff] <= 0.12 * scale).all())
