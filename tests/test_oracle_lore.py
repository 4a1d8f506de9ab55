"""Pins the Lore oracle (oracle/lore_net.py): DCN against the reference's own vendored DCNv2 C++ (oracle/_ref), the
DLA-34 + DCN graph against outputs of the reference ``DLASeg`` module (tests/golden/lore_dla34.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lore_net
from pdf_table_amd.synth_weights import lore_dla34_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libdcnv2_ref.so")


def test_dcn_zero_offset_is_conv2d():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 9, 11, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    b = torch.randn(7, generator=g)
    off = torch.zeros(2, 18, 9, 11)
    m = torch.ones(2, 9, 9, 11)
    y = lore_net.deform_conv2d(x, off, m, w, b)
    assert torch.allclose(y, F.conv2d(x, w, b, 1, 1), atol=1e-5)


def test_dcn_integer_offset_is_shift():
    """every tap displaced by (+1, -2) == the plain conv of the image shifted by the same amount (compared away from
    the border, where the two zero-fill rules differ)"""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 10, 12, generator=g)
    w = torch.randn(4, 3, 3, 3, generator=g)
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    y = lore_net.deform_conv2d(x, off, torch.ones(1, 9, 10, 12), w, None)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]
    assert torch.allclose(y[:, :, 2:-2, 3:-3], F.conv2d(xs, w, None, 1, 1)[:, :, 2:-2, 3:-3], atol=1e-5)


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (needs /root/reference; __graft_entry__.build())")
@pytest.mark.parametrize("shape,scale", [((2, 6, 13, 17), 1.5), ((1, 16, 32, 32), 4.0), ((1, 3, 5, 5), 8.0)])
def test_dcn_columns_equal_reference_cpp(shape, scale):
    """columns of the oracle == modulated_deformable_im2col_cpu compiled from the reference tree, bit for bit
    (offsets large enough to leave the image on every side)."""
    rng = np.random.default_rng(7)
    B, C, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * scale).astype(np.float32)
    off[:, :, 0, 0] = np.round(off[:, :, 0, 0])                     # exact-integer positions too
    m = rng.uniform(0, 1, (B, 9, H, W)).astype(np.float32)
    ref = lore_net.ref_dcn_im2col(x, off, m)                         # [B, C*9, H, W]
    mine = lore_net.deform_conv2d(torch.from_numpy(x), torch.from_numpy(off), torch.from_numpy(m),
                                  torch.zeros(1, C, 3, 3), None, return_cols=True).numpy().reshape(B, C * 9, H, W)
    assert np.array_equal(ref, mine)


def test_dlaseg_equals_reference_module():
    gold = np.load(os.path.join(HERE, "golden", "lore_dla34.npz"))
    sd = lore_dla34_state_dict(int(gold["seed"]))
    for tag in ("a", "b"):
        with torch.no_grad():
            z = lore_net.dlaseg_forward(sd, torch.from_numpy(gold[f"x_{tag}"]))
        for k in lore_net.HEADS:
            v = z[k].numpy()
            v = v[:, ::8] if v.shape[1] == 256 else v
            assert np.allclose(v, gold[f"{k}_{tag}"], atol=2e-5, rtol=1e-5), (tag, k, np.abs(v - gold[f"{k}_{tag}"]).max())


# ---- decode: oracle/lore_decode.py against the reference's own process_detect_output -------------------------------
def _decode_case(gold, tag):
    import sys
    sys.path.insert(0, HERE)
    from lore_synth import synth_lore_heads
    seed, H, W, src_h, src_w, rev, ul = [int(v) for v in gold[f"case_{tag}"]]
    heads = {k: torch.from_numpy(v) for k, v in synth_lore_heads(seed, H, W).items()}
    from oracle import lore_decode as od
    _, meta = od.lore_preprocess_geometry(src_h, src_w, 4 * H, 4 * W, upper_left=bool(ul))
    assert np.array_equal(meta, gold[f"meta_{tag}"])
    return heads, gold[f"meta_{tag}"], bool(rev), bool(ul)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_decode_equals_reference(tag):
    from oracle import lore_decode as od
    gold = np.load(os.path.join(HERE, "golden", "lore_decode.npz"))
    heads, meta, rev, ul = _decode_case(gold, tag)
    logi, ps, polys, results = od.process_detect_output(heads, meta, wiz_rev=rev, vis_thresh=0.2, upper_left=ul)
    assert logi.shape[1] == gold[f"logi_{tag}"].shape[1] > 20
    assert np.array_equal(logi.numpy(), gold[f"logi_{tag}"])
    assert np.array_equal(ps.numpy(), gold[f"ps_{tag}"])
    n = logi.shape[1]
    assert np.array_equal(results[:n], gold[f"results_{tag}"][:n])          # valid cells: bit-exact, same order
    # beyond the valid cells the reference's rows are zero-score peaks in torch.topk tie order: same multiset of rows
    ours = results[np.lexsort(results.T[::-1])]
    ref = gold[f"results_{tag}"]
    assert np.array_equal(ours, ref[np.lexsort(ref.T[::-1])])


def test_process_logic_output_equals_reference():
    from oracle import lore_decode as od
    gold = np.load(os.path.join(HERE, "golden", "lore_decode.npz"))
    assert np.array_equal(od.process_logic_output(torch.from_numpy(gold["logic_in"])).numpy(), gold["logic_out"])


def test_point_in_polygon_known_answers():
    from oracle.lore_decode import point_strictly_in_polygon as pin
    sq = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], dtype=np.float64)
    assert pin(2, 2, sq) and pin(0.001, 3.999, sq)
    assert not pin(0, 2, sq) and not pin(4, 4, sq) and not pin(2, 0, sq)      # boundary is not "within"
    assert not pin(5, 2, sq) and not pin(-1, -1, sq)
    para = np.array([[0, 0], [6, 1], [8, 5], [2, 4]], dtype=np.float64)
    assert pin(4, 2.5, para) and not pin(1, 3, para) and not pin(7, 1.2, para)


def test_affine_geometry_known_answers():
    from oracle import lore_decode as od
    # 300 x 420 source into a 1024 x 1024 input: uniform scale 1024 / 420 about the centres
    trans, meta = od.lore_preprocess_geometry(300, 420)
    assert meta.tolist() == [210, 150, 420, 1024, 1024, 256, 256]
    k = 1024 / 420
    assert np.allclose(trans, [[k, 0, 512 - 210 * k], [0, k, 512 - 150 * k]], atol=1e-9)
    back = od.transform_preds(np.array([[128.0, 128.0], [0.0, 0.0]], np.float32), meta[:2], meta[2], (256, 256))
    assert np.allclose(back, [[210, 150], [210 - 128 * 420 / 256, 150 - 128 * 420 / 256]], atol=1e-9)


@pytest.mark.parametrize("tag", ["wtw", "ptn"])
def test_processor_equals_reference_module(tag):
    from oracle import lore_processor as op
    from pdf_table_amd.synth_weights import lore_processor_state_dict
    gold = np.load(os.path.join(HERE, "golden", "lore_processor.npz"))
    L = int(gold[f"layers_{tag}"])
    sd = lore_processor_state_dict(seed=31, layers=L, stacking_layers=L)
    dets = torch.from_numpy(gold[f"dets_{tag}"]) if f"dets_{tag}" in gold else None
    with torch.no_grad():
        logic, stacked = op.processor_forward(sd, torch.from_numpy(gold[f"feat_{tag}"]), dets, L, L)
    assert np.allclose(logic.numpy(), gold[f"logic_{tag}"], atol=2e-5)
    assert np.allclose(stacked.numpy(), gold[f"stacked_{tag}"], atol=2e-5)


def test_warp_affine_known_answers():
    from oracle import lore_pre
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 30, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(lore_pre.warp_affine_u8(img, ident, 30, 20), img)
    shift = np.array([[1, 0, 4], [0, 1, -3]], np.float64)            # dst(x, y) = src(x - 4, y + 3), zero outside
    out = lore_pre.warp_affine_u8(img, shift, 30, 20)
    exp = np.zeros_like(img)
    exp[:17, 4:] = img[3:, :26]
    assert np.array_equal(out, exp)
    # dst = src / 2 about the origin: dst(x, y) samples src(2x, 2y) exactly
    half = np.array([[0.5, 0, 0], [0, 0.5, 0]], np.float64)
    assert np.array_equal(lore_pre.warp_affine_u8(img, half, 15, 10), img[::2, ::2])
    # half-pixel shift: mean of horizontal neighbours, rounded half up ((a+b)*16384 + 16384) >> 15
    hshift = np.array([[1, 0, 0.5], [0, 1, 0]], np.float64)
    out = lore_pre.warp_affine_u8(img, hshift, 30, 20).astype(np.int64)
    exp = (img[:, 1:].astype(np.int64) + img[:, :-1] + 1) >> 1
    assert np.array_equal(out[:, 1:], exp)


def test_wireless_detector_equals_reference_module():
    from pdf_table_amd.synth_weights import lore_wireless_state_dict
    gold = np.load(os.path.join(HERE, "golden", "lore_wireless.npz"))
    sd = lore_wireless_state_dict(int(gold["seed"]))
    for tag in ("a", "b"):
        with torch.no_grad():
            z = lore_net.lore_wireless_forward(sd, torch.from_numpy(gold[f"x_{tag}"]))
        for k in lore_net.HEADS:
            v = z[k].numpy()
            v = v[:, ::8] if v.shape[1] == 256 else v
            assert np.allclose(v, gold[f"{k}_{tag}"], atol=2e-5, rtol=1e-5), (tag, k, np.abs(v - gold[f"{k}_{tag}"]).max())
