"""GPU parity tests of the recognition stage (crop -> resize -> CRNN -> arg-max) through the C ABI.

Integer work (crop/resize pixels, token ids) bit-exact; float work (winning logit) within 1e-3 of the oracle's fp32
restatement in PT_PRECISION_BF16X3; token ids may differ from the oracle's only where the oracle's own top-2 margin
is inside that tolerance."""
import numpy as np
import pytest
import torch

from oracle import crnn as ocrnn
from pdf_table_amd import lib as L
from pdf_table_amd import rec_stage as R
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import crnn_state_dict
from pdf_table_amd.weights import pack_crnn

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def sd():
    return crnn_state_dict(seed=12)


@pytest.fixture(scope="module")
def eng(sd):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd))
    yield e
    e.close()


def _page_and_boxes(idx=3, k=12):
    img, meta = make_page(idx)
    rng = np.random.default_rng(idx)
    boxes = []
    for (x0, y0, x1, y1) in meta["lines"][:k]:
        jit = rng.uniform(-1.5, 1.5, 8)
        q = np.array([x0 - 2, y0 - 2, x1 + 2, y0 - 2, x1 + 2, y1 + 2, x0 - 2, y1 + 2], np.float64) + jit
        boxes.append(np.round(q))
    # one rotated quad and one tall (ratio < 1) quad
    boxes.append(np.array([100, 400, 380, 430, 376, 462, 96, 432], np.float64))
    boxes.append(np.array([500, 300, 520, 300, 520, 380, 500, 380], np.float64))
    return img, np.array(boxes)


def _oracle_gray(img, box):
    crop = ocrnn.crop_image(img, ocrnn.order_point(box))
    x = ocrnn.rec_preprocess(crop)                                        # [1,3,32,640] fp32
    return (x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140)[0, 0], x


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_rec_preprocess_bit_exact(eng, mode):
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        img, boxes = _page_and_boxes()
        lines = R.build_lines([boxes])
        gray = eng.rec_preprocess(torch.from_numpy(img[None]).cuda(), lines)
        torch.cuda.synchronize()
        g = gray.float().cpu()
        for i, b in enumerate(boxes):
            ref, _ = _oracle_gray(img, b)
            if mode == "bf16":
                want = ref.to(torch.bfloat16).float()
                assert torch.equal(g[i], want), f"line {i}: {(g[i] - want).abs().max()}"
            else:
                hi = ref.to(torch.bfloat16).float()
                lo = (ref - hi).to(torch.bfloat16).float()
                assert torch.equal(g[i, :, :, 0], hi) and torch.equal(g[i, :, :, 1], lo)
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_rec_net_x3_matches_fp32_oracle(eng, sd):
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        rng = np.random.default_rng(7)
        x = rng.uniform(0, 1, (5, 3, 32, 640)).astype(np.float32)
        x[1, :, :, 300:] = 0
        x[2, :, :, 90:] = 0
        x[4] = 0
        xt = torch.from_numpy(x)
        gray = xt[:, 0] * 0.2989 + xt[:, 1] * 0.5870 + xt[:, 2] * 0.1140
        with torch.no_grad():
            logits = ocrnn.crnn_forward_fp32(sd, xt)
        top2 = torch.topk(logits, 2, dim=-1)
        hi = gray.to(torch.bfloat16)
        lo = (gray - hi.float()).to(torch.bfloat16)
        ids, mx = eng.rec_forward_net(torch.stack([hi, lo], -1).contiguous().cuda())
        torch.cuda.synchronize()
        ids, mx = ids.cpu(), mx.cpu()
        dmax = (mx - top2.values[..., 0]).abs().max().item()
        margin = top2.values[..., 0] - top2.values[..., 1]
        diff = ids != top2.indices[..., 0]
        print(f"crnn x3: max|d max-logit| = {dmax:.2e}; {int(diff.sum())} of {ids.numel()} ids differ; "
              f"min margin {margin.min().item():.2e}")
        assert dmax <= TOL
        assert bool((margin[diff] <= 2 * TOL).all())
        assert float(diff.float().mean()) < 0.01
        assert R.ctc_collapse(ids.numpy()) == ocrnn.ctc_greedy_ids(ids.numpy())
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_rec_net_bf16_close(eng, sd):
    """Throughput mode: bf16-class drift on the winning logit; ids agree wherever the oracle's margin is clear."""
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1, (3, 3, 32, 640)).astype(np.float32)
    xt = torch.from_numpy(x)
    gray = (xt[:, 0] * 0.2989 + xt[:, 1] * 0.5870 + xt[:, 2] * 0.1140).to(torch.bfloat16)
    with torch.no_grad():
        logits = ocrnn.crnn_forward_fp32(sd, xt)
    top2 = torch.topk(logits, 2, dim=-1)
    ids, mx = eng.rec_forward_net(gray.contiguous().cuda())
    ids, mx = ids.cpu(), mx.cpu()
    scale = logits.abs().max().item()
    dmax = (mx - top2.values[..., 0]).abs().max().item()
    margin = top2.values[..., 0] - top2.values[..., 1]
    diff = ids != top2.indices[..., 0]
    print(f"crnn bf16: max|d max-logit| = {dmax:.2e} (scale {scale:.1f}); {int(diff.sum())} of {ids.numel()} ids differ")
    assert dmax <= 0.06 * scale
    assert bool((margin[diff] <= 0.12 * scale).all())


def test_rec_end_to_end_x3(eng, sd):
    """pages + quads -> pt_rec_forward (crop, resize, net, arg-max) vs the oracle run line by line like the reference."""
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        img, boxes = _page_and_boxes(idx=4, k=6)
        stage = R.RecStage(eng)
        ids, lines = stage.ids(torch.from_numpy(img[None]).cuda(), [boxes])
        ids = ids.cpu().numpy()
        for i, b in enumerate(boxes):
            _, x = _oracle_gray(img, b)
            with torch.no_grad():
                logits = ocrnn.crnn_forward_fp32(sd, x)[0]
            top2 = torch.topk(logits, 2, dim=-1)
            margin = (top2.values[:, 0] - top2.values[:, 1]).numpy()
            d = ids[i] != top2.indices[:, 0].numpy()
            assert (margin[d] <= 2 * TOL).all(), (i, margin[d])
        texts = stage(torch.from_numpy(img[None]).cuda(), [boxes])
        assert len(texts) == 1 and len(texts[0]) == len(boxes)
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_rec_bf16_fast_kernels_equal_tiled_kernels(tmp_path):
    """bf16 mode: the weight-stationary LSTM (lstm_cluster_kernel) and the streaming row GEMMs / fused arg-max
    (gemm_argmax_kernel) give bit-identical ids AND winning logits to the streaming LSTM + tiled 1x1 GEMM + arg-max reduce
    they replaced, and the max-pools fused into the conv epilogues to the separate pool kernels (selected with
    PT_LSTM_CLUSTER=0 PT_CLS_FUSED=0 PT_POOL_FUSED=0 in a child process: the switches are read once).
    300 lines = three 128-line clusters, the last one partial; with PT_LSTM_MI=3 two 192-line clusters (the variant a
    launch of more than 4096 lines selects), the same sums in the same order."""
    import os
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_weights import crnn_state_dict
from pdf_table_amd.weights import pack_crnn
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_CRNN, pack_crnn(crnn_state_dict(seed=1), x3=False))
rng = np.random.default_rng(21)
g = rng.uniform(0, 1, (300, 32, 640)).astype(np.float32)
for i in range(0, 300, 7):
    g[i, :, int(rng.integers(40, 600)):] = 0          # ragged right padding like real lines
ids, mx = eng.rec_forward_net(torch.from_numpy(g).to(torch.bfloat16).cuda())
np.savez(sys.argv[1], ids=ids.cpu().numpy(), mx=mx.cpu().numpy())
'''
    outs = []
    for tag, env in (("fast", {}), ("tiled", {"PT_LSTM_CLUSTER": "0", "PT_CLS_FUSED": "0", "PT_POOL_FUSED": "0"}),
                     ("fast192", {"PT_LSTM_MI": "3"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ, **env)
        e["PT_CONV_VARIANT"] = "0"     # one conv kernel family in both runs: the DMA variants sum K in another order
        e["PT_CONV01"] = "0"           # (and so does the fused conv0 + conv1 kernel: test_conv0_conv1_in_one_launch_equals_the_separate_launches)
        e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=e, timeout=300)
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["ids"], outs[1]["ids"])
    assert np.array_equal(outs[0]["mx"], outs[1]["mx"])
    assert np.array_equal(outs[0]["ids"], outs[2]["ids"])
    assert np.array_equal(outs[0]["mx"], outs[2]["mx"])
    assert len(np.unique(outs[0]["ids"])) > 20           # not a degenerate output


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_set_lstm_cluster_switches_kernels_in_process(eng, sd, fmt):
    """pt_engine_set_lstm_cluster: the streaming LSTM (what an overlapping pipeline uses) and the cluster LSTM give the
    same ids and winning logits; pt_engine_check has nothing to report after either (f16: the same pair in namespace pt_f16)"""
    rng = np.random.default_rng(33)
    g = torch.from_numpy(rng.uniform(0, 1, (70, 32, 640)).astype(np.float32))
    own = None
    if fmt == "f16":
        from pdf_table_amd.engine import HipEngine
        eng = own = HipEngine(0)
        eng.set_precision(L.PT_PRECISION_F16)
        eng.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd, fmt="f16"))
    g = g.to(eng.act_dtype).cuda()
    outs = []
    try:
        for on in (True, False, True):
            eng.set_lstm_cluster(on)
            ids, mx = eng.rec_forward_net(g)
            torch.cuda.synchronize()
            eng.check()
            outs.append((ids.cpu().numpy(), mx.cpu().numpy()))
    finally:
        eng.set_lstm_cluster(True)
        if own is not None:
            own.close()
    for ids, mx in outs[1:]:
        assert np.array_equal(ids, outs[0][0]) and np.array_equal(mx, outs[0][1])


def _x3_input(n, seed):
    rng = np.random.default_rng(seed)
    g = torch.from_numpy(rng.uniform(0, 1, (n, 32, 640)).astype(np.float32))
    g[n // 2, :, 200:] = 0
    hi = g.to(torch.bfloat16)
    return torch.stack([hi, (g - hi.float()).to(torch.bfloat16)], -1).contiguous().cuda()


def test_x3_row_gemm_equals_the_conv_kernel(eng, sd, monkeypatch):
    """BF16X3 row GEMMs of the sequence head (LSTM input projections, embeddings): gemm_rows_x3_kernel against the three-pass 1x1 conv kernel
    (PT_ROWS_X3=0, read per call) -- the same MFMA sequence per output, so ids AND winning logits are bit-identical; 37 lines (1.16 row blocks
    of the last workgroup), one of them mostly padding (ragged first projection)"""
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        x = _x3_input(37, 9)
        monkeypatch.delenv("PT_ROWS_X3", raising=False)
        ids, mx = eng.rec_forward_net(x)
        monkeypatch.setenv("PT_ROWS_X3", "0")
        ids0, mx0 = eng.rec_forward_net(x)
        torch.cuda.synchronize()
        assert np.array_equal(ids.cpu().numpy(), ids0.cpu().numpy())
        assert np.array_equal(mx.cpu().numpy(), mx0.cpu().numpy())
        assert len(np.unique(ids.cpu().numpy())) > 20
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_x3_classifier_bound_and_refine_equals_the_tiled_classifier(eng, sd, monkeypatch):
    """BF16X3 classifier: the arg-max from two single-pass sweeps + exact logits of the candidates (gemm_cand_kernel / cand_eval_kernel)
    against the tiled three-pass GEMM with per-tile partials + reduce (PT_CLS_X3_REFINE=0, read per call): the same ids wherever the
    three-term sum's top two are further apart than its own rounding, winning logits within 1e-3 of each other"""
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        x = _x3_input(37, 5)
        monkeypatch.delenv("PT_CLS_X3_REFINE", raising=False)
        ids, mx = eng.rec_forward_net(x)
        monkeypatch.setenv("PT_CLS_X3_REFINE", "0")
        ids0, mx0 = eng.rec_forward_net(x)
        torch.cuda.synchronize()
        ids, mx, ids0, mx0 = ids.cpu(), mx.cpu(), ids0.cpu(), mx0.cpu()
        assert float((mx - mx0).abs().max()) <= TOL
        diff = ids != ids0
        print(f"x3 classifier: {int(diff.sum())} of {ids.numel()} ids differ between bound-and-refine and the tiled GEMM")
        assert float(diff.float().mean()) < 1e-3
        assert len(np.unique(ids.numpy())) > 20
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


def test_x3_classifier_rows_with_more_candidates_than_slots(monkeypatch):
    """a classifier whose logits all coincide (zero weights; the CRNN's Linear has no bias): EVERY class is inside the rounding bound of the
    maximum, the per-row slots overflow and cand_full_kernel evaluates all classes -- the lowest class index wins, as torch.argmax.  Then two
    identical non-zero rows (classes 17 and 4321) over zeros: rows where they win tie between the two (slot path, 17 wins), the others fall
    back to the all-zero tie (overflow path, 0 wins) -- in both cases what the tiled three-pass classifier answers"""
    from pdf_table_amd.engine import HipEngine
    base = crnn_state_dict(seed=3)
    sd = dict(base)
    sd["cls.weight"] = torch.zeros_like(base["cls.weight"])
    e = HipEngine(0)
    try:
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd))
        e.set_precision(L.PT_PRECISION_BF16X3)
        x = _x3_input(3, 9)
        ids, mx = e.rec_forward_net(x)
        torch.cuda.synchronize()
        assert int(ids.abs().max()) == 0 and float(mx.abs().max()) == 0.0
        w = torch.zeros_like(base["cls.weight"])
        w[17] = base["cls.weight"][5]
        w[4321] = base["cls.weight"][5]
        sd["cls.weight"] = w
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd))
        monkeypatch.delenv("PT_CLS_X3_REFINE", raising=False)
        ids, mx = e.rec_forward_net(x)
        monkeypatch.setenv("PT_CLS_X3_REFINE", "0")
        ids0, mx0 = e.rec_forward_net(x)
        torch.cuda.synchronize()
        ids, ids0 = ids.cpu(), ids0.cpu()
        assert set(np.unique(ids.numpy()).tolist()) <= {0, 17}
        assert bool((ids == ids0).all()) and float((mx.cpu() - mx0.cpu()).abs().max()) <= TOL
    finally:
        e.close()


def test_x3_cluster_lstm_equals_the_streaming_lstm(eng, sd, monkeypatch):
    """lstm_cluster8_x3_kernel (eight members x 32 hidden units, (hi, lo) W_hh in LDS) against the streaming hi/lo kernel
    (PT_LSTM_CLUSTER_X3=0): the same recurrence summed per k-step instead of per pass -- ids equal, winning logits within 1e-3; eight and
    four waves per member at both cluster sizes each (PT_LSTM_X3_WAVES, PT_LSTM_MI), a batch that does not fill its last cluster, nothing for
    pt_engine_check to report"""
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        x = _x3_input(300, 11)
        monkeypatch.setenv("PT_LSTM_CLUSTER_X3", "0")
        ids0, mx0 = eng.rec_forward_net(x)
        torch.cuda.synchronize()
        ids0, mx0 = ids0.cpu(), mx0.cpu()
        monkeypatch.delenv("PT_LSTM_CLUSTER_X3")
        for waves, mi in (("8", "1"), ("8", "2"), ("4", "2"), ("4", "3")):      # eight / four waves per member, tiles per wave
            monkeypatch.setenv("PT_LSTM_X3_WAVES", waves)
            monkeypatch.setenv("PT_LSTM_MI", mi)
            ids, mx = eng.rec_forward_net(x)
            torch.cuda.synchronize()
            eng.check()
            assert float((mx.cpu() - mx0).abs().max()) <= TOL
            assert float((ids.cpu() != ids0).float().mean()) < 1e-3
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)


# ---- PP-OCR recognition pre-processor (PPOcrRecPreProcessor, ocr_rec_pp/processor_ocr_rec_pp.py:24-135) ---------------
def test_rec_pp_preprocessor_crops_bit_exact(eng, golden_dir):
    """already-cropped inputs (the reference call shape), all crops in one call and one crop per call: every mini-batch
    array equals the oracle's AND the reference's own output (tests/golden/rec_pp.npz) bit for bit"""
    import os
    from oracle import rec_pp
    from pdf_table_amd.rec_pp_stage import PPOcrRecPreProcessor
    from rec_synth import rec_pp_crops
    gold = np.load(os.path.join(golden_dir, "rec_pp.npz"))
    crops = rec_pp_crops(int(gold["seed"]))
    pre = PPOcrRecPreProcessor(engine=eng)
    got = pre(list(crops))
    ref = rec_pp.rec_pp_preprocess(crops)
    assert len(got) == len(ref) == int(gold["n_batches"])
    for b, (g, r) in enumerate(zip(got, ref)):
        assert g["batch_beg_img_no"] == r["batch_beg_img_no"] and np.array_equal(g["indices"], r["indices"])
        a = g["image"].cpu().numpy()
        assert a.dtype == np.float32 and a.shape == r["image"].shape
        assert np.array_equal(a, r["image"]) and np.array_equal(a, gold[f"batch{b}"])
    for i in (0, 4, 5, 6):
        one = pre(crops[i])
        assert len(one) == 1 and np.array_equal(one[0]["image"].cpu().numpy(), gold[f"single{i}"])
    gray = pre(crops[0][:, :, 0])                                  # a 2-D input is replicated to three channels (cv2.COLOR_GRAY2RGB)
    want = rec_pp.rec_pp_preprocess([np.repeat(crops[0][:, :, :1], 3, 2)])
    assert np.array_equal(gray[0]["image"].cpu().numpy(), want[0]["image"])
    assert pre([]) == []
    with pytest.raises(TypeError):
        pre(3.14)


def test_rec_pp_preprocessor_page_lines_bit_exact(eng):
    """lines cut from a resident page on the device (order_point + crop_image, then the PP pre-processor) equal the oracle
    run crop by crop: ~60 lines -> 10 width-sorted mini-batches"""
    from oracle import rec_pp
    from pdf_table_amd.rec_pp_stage import PPOcrRecPreProcessor
    img, boxes = _page_and_boxes(idx=6, k=61)
    pre = PPOcrRecPreProcessor(engine=eng)
    got = pre.lines(torch.from_numpy(img[None]).cuda(), [boxes])
    crops = [ocrnn.crop_image(img, ocrnn.order_point(b)) for b in boxes]
    ref = rec_pp.rec_pp_preprocess(crops)
    assert len(got) == len(ref) and len(ref) >= 3
    widths = set()
    for g, r in zip(got, ref):
        assert np.array_equal(g["indices"], r["indices"]) and g["batch_beg_img_no"] == r["batch_beg_img_no"]
        assert np.array_equal(g["image"].cpu().numpy(), r["image"])
        widths.add(r["image"].shape[3])
    assert len(widths) >= 2                                       # mini-batches of different padded widths


@pytest.mark.parametrize("mode", ["bf16", "f16", "bf16x3"])
def test_ragged_conv_stack_is_bit_identical_to_the_full_one(sd, mode):
    """pt_rec_forward knows every line's crop size, so the conv stack does no work right of the text (skipped columns are
    filled with an all-padding line's activations): token ids AND winning logits equal the full computation bit for bit
    (PT_REC_RAGGED=0, read when the engine is created), for short, long, over-long, tall and degenerate lines"""
    import os
    from pdf_table_amd.engine import HipEngine
    img = make_page(7)[0]
    rng = np.random.default_rng(11)
    boxes = []
    for k in range(150):
        w = int(rng.choice([6, 17, 40, 90, 150, 260, 420, 700, 990]))
        h = int(rng.choice([9, 14, 22, 31, 48]))
        x0, y0 = int(rng.integers(0, 1024 - min(w, 1000))), int(rng.integers(0, 1024 - h))
        boxes.append([x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h])
    boxes.append([10, 10, 10, 10, 10, 10, 10, 10])                  # degenerate: empty crop, all padding
    boxes.append([5, 100, 25, 100, 25, 400, 5, 400])                # tall: 20 x 300 -> two columns of text
    boxes = np.array(boxes, np.float64)
    outs = []
    for ragged in ("1", "0"):
        os.environ["PT_REC_RAGGED"] = ragged
        try:
            e = HipEngine(0)
        finally:
            del os.environ["PT_REC_RAGGED"]
        e.set_precision({"bf16x3": L.PT_PRECISION_BF16X3, "f16": L.PT_PRECISION_F16}.get(mode, L.PT_PRECISION_BF16))
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd, fmt=e.weight_fmt))          # f16: the pt_f16 instantiation of the same kernels on fp16 tiles
        lines = R.build_lines([boxes])
        pages = torch.from_numpy(img[None]).cuda()
        ids, mx = e.rec_forward(pages, lines)
        ids2, mx2 = e.rec_forward(pages, lines[::-1].copy())       # second call: cached all-padding line, other line order
        torch.cuda.synchronize()
        e.check()
        assert torch.equal(ids2.flip(0), ids) and torch.equal(mx2.flip(0), mx)
        outs.append((ids.cpu().numpy(), mx.cpu().numpy()))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert len(np.unique(outs[0][0])) > 20


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_conv0_conv1_in_one_launch_equals_the_separate_launches(sd, mode, monkeypatch):
    """crnn_conv01_kernel (conv0 + pool + conv1 + pool in one launch; the 64-channel 16 x 320 map is recomputed per tile in LDS and never stored) is the
    kernel that runs in the single-pass modes, and its token ids and winning logits equal conv0+pool -> the v4 conv kernel -> MaxPool2d(2) as three launches
    (PT_CONV01=0 PT_POOL_FUSED=0, read per call: the same K order -- 16-channel slices outer, taps inner) bit for bit; so does the ragged run (column limits
    per line) of the fused kernel.  (Against PT_CONV01=0 alone the logits differ by bf16 noise: that path's conv1 is the register-staged kernel, whose K loop
    runs taps outer over 32-channel chunks.)"""
    import os
    from pdf_table_amd.engine import HipEngine
    img = make_page(5)[0]
    rng = np.random.default_rng(23)
    boxes = []
    for k in range(90):
        w = int(rng.choice([6, 17, 40, 90, 150, 260, 420, 700, 990]))
        h = int(rng.choice([9, 14, 22, 31, 48]))
        x0, y0 = int(rng.integers(0, 1024 - min(w, 1000))), int(rng.integers(0, 1024 - h))
        boxes.append([x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h])
    boxes.append([10, 10, 10, 10, 10, 10, 10, 10])
    boxes = np.array(boxes, np.float64)
    pages = torch.from_numpy(img[None]).cuda()
    lines = R.build_lines([boxes])

    def run(e):
        e.profile_enable(True)
        ids, mx = e.rec_forward(pages, lines)
        torch.cuda.synchronize()
        labels = list(e.profile_read_labels())
        e.profile_enable(False)
        e.check()
        return (ids.cpu().numpy(), mx.cpu().numpy()), labels

    outs = {}
    for ragged in ("0", "1"):
        os.environ["PT_REC_RAGGED"] = ragged
        try:
            e = HipEngine(0)
        finally:
            del os.environ["PT_REC_RAGGED"]
        e.set_precision({"f16": L.PT_PRECISION_F16}.get(mode, L.PT_PRECISION_BF16))
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd, fmt=e.weight_fmt))
        outs["fused" + ragged], labels = run(e)
        assert any(l.startswith("conv0+pool+conv1+pool") for l in labels) and not any(l.startswith("crnn conv0+pool") for l in labels), labels
        if ragged == "0":
            monkeypatch.setenv("PT_CONV01", "0")
            outs["two"], labels = run(e)
            assert any(l.startswith("crnn conv0+pool") for l in labels) and any(l.startswith("conv3x3 s1 64->128") for l in labels), labels
            monkeypatch.setenv("PT_POOL_FUSED", "0")
            outs["three"], labels = run(e)
            assert any(l.startswith("crnn conv0+pool") for l in labels) and any(l.startswith(("conv3x3 v4", "conv3x3 v5")) and "64->128" in l for l in labels), labels
            monkeypatch.delenv("PT_CONV01")
            monkeypatch.delenv("PT_POOL_FUSED")
        e.close()
    for k in ("three", "fused1"):
        assert np.array_equal(outs["fused0"][0], outs[k][0]) and np.array_equal(outs["fused0"][1], outs[k][1]), k
    # the register-staged conv1 of the two-launch path: another summation order, same network
    assert (outs["fused0"][0] != outs["two"][0]).mean() < 0.01 and np.abs(outs["fused0"][1] - outs["two"][1]).max() < 0.25
    assert len(np.unique(outs["fused0"][0])) > 20


@pytest.mark.parametrize("mode", ["bf16", "f16", "bf16x3"])
def test_classifier_dma_kernel_equals_the_register_staged_one(eng, sd, mode, monkeypatch):
    """cls_argmax_dma_kernel (class tiles by LDS-DMA into two buffers, both 32-class halves multiplied together, fragment reads pinned two steps ahead) gives
    the ids and maxima of gemm_argmax_kernel<32, 0, 8> (PT_CLS_DMA=0, read per call) bit for bit; in the hi/lo mode it is the first sweep of the
    bound-and-refine arg-max.  600 rows x 160 steps: several workgroups, the last one partial."""
    from pdf_table_amd.engine import HipEngine
    rng = np.random.default_rng(31)
    g = rng.uniform(0, 1, (77, 32, 640)).astype(np.float32)
    for i in range(0, 77, 5):
        g[i, :, int(rng.integers(40, 600)):] = 0
    e = HipEngine(0)
    try:
        e.set_precision({"bf16x3": L.PT_PRECISION_BF16X3, "f16": L.PT_PRECISION_F16}.get(mode, L.PT_PRECISION_BF16))
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(sd, fmt=e.weight_fmt))
        gt = torch.from_numpy(g)
        if mode == "bf16x3":
            hi = gt.to(torch.bfloat16)
            x = torch.stack([hi, (gt - hi.float()).to(torch.bfloat16)], -1).contiguous().cuda()
        else:
            x = gt.to(torch.float16 if mode == "f16" else torch.bfloat16).cuda()
        outs = {}
        for sw in ("1", "0"):
            monkeypatch.setenv("PT_CLS_DMA", sw)
            ids, mx = e.rec_forward_net(x)
            torch.cuda.synchronize()
            e.check()
            outs[sw] = (ids.cpu().numpy(), mx.cpu().numpy())
    finally:
        e.close()
    assert np.array_equal(outs["1"][0], outs["0"][0]) and np.array_equal(outs["1"][1], outs["0"][1])
    assert len(np.unique(outs["1"][0])) > 20
