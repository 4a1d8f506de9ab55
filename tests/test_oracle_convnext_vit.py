"""Pin oracle/convnext_vit.py against golden vectors produced by the reference's own ConvNextViT module
(tests/golden/make_golden.py::gen_convnext_vit, reference model/convnext_vit/modeling_convnext_vit.py:20-45)."""
import os

import numpy as np
import torch

from oracle import convnext_vit as ocv
from pdf_table_amd.synth_weights import convnext_vit_state_dict


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "convnext_vit.npz"))


def test_fp32_oracle_matches_reference_module(golden_dir):
    g = _golden(golden_dir)
    sd = ocv.canonical_state_dict(convnext_vit_state_dict(seed=int(g["seed"])))
    x = torch.from_numpy(g["img_u8"]).float().div(255.).permute(0, 3, 1, 2)
    with torch.no_grad():
        gray = x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140
        feats = ocv.cnn_forward_fp32(sd, gray)                     # NHWC [3,1,75,512]
        seq = ocv.vit_features_fp32(sd, feats)
        logits = ocv.convnext_vit_forward_fp32(sd, x)
    np.testing.assert_allclose(feats[:, 0].permute(0, 2, 1)[:, ::8, ::5].numpy(), g["feats_sub"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(seq[:, ::5, ::8].numpy(), g["seq_sub"], rtol=1e-4, atol=1e-4)
    scale = float(g["logits_abs_max"])
    assert np.abs(logits[:, :, ::97].numpy() - g["logits_sub"]).max() <= 1e-5 * scale
    top2 = torch.topk(logits, 2, dim=-1)
    assert np.abs(top2.values.numpy() - g["top2_val"]).max() <= 1e-5 * scale
    # ids equal wherever the reference's own margin is above fp32 noise
    margin = g["top2_val"][..., 0] - g["top2_val"][..., 1]
    same = top2.indices[..., 0].numpy() == g["top2_idx"][..., 0]
    assert same[margin > 1e-4 * scale].all()


def test_both_key_name_generations_are_accepted():
    sd = convnext_vit_state_dict(seed=3)
    v5 = {}
    for k, v in sd.items():
        k = k.replace("vit.encoder.layer.", "vit.layers.").replace(".attention.attention.query.", ".attention.q_proj.")
        k = k.replace(".attention.attention.key.", ".attention.k_proj.").replace(".attention.attention.value.", ".attention.v_proj.")
        k = k.replace(".attention.output.dense.", ".attention.o_proj.").replace(".intermediate.dense.", ".mlp.fc1.")
        if ".vit.layers." in k:
            k = k.replace(".output.dense.", ".mlp.fc2.")
        v5["recognizer." + k] = v
    back = ocv.canonical_state_dict(v5)
    assert set(back) == set(sd)
    assert all(torch.equal(back[k], sd[k]) for k in sd)


def test_chunk_preprocess_layout_and_greedy_text():
    rng = np.random.default_rng(0)
    crop = rng.integers(0, 256, (40, 700, 3), dtype=np.uint8)
    d = ocv.chunk_preprocess(crop)
    assert tuple(d.shape) == (3, 3, 32, 300)
    from oracle.crnn import keepratio_resize
    full = keepratio_resize(crop, 32, 804).astype(np.float32) / 255.
    for i in range(3):
        np.testing.assert_array_equal(d[i].permute(1, 2, 0).numpy(), full[:, 252 * i:252 * i + 300])
    assert (full[:, 560:] == 0).all()            # int(32 * 700 / 40) = 560 text columns, zero padding to 804
    logits = torch.full((1, 6, 8), -1.0)
    for t, c in enumerate([0, 3, 3, 0, 3, 2]):
        logits[0, t, c] = 1.0
    assert ocv.greedy_text(logits, ["a", "b", "c"]) == ["bba"]


def _read_blob(raw):
    """the PTW1 container of pdf_table_amd/weights.py: name -> ndarray"""
    import struct
    assert raw[:4] == b"PTW1"
    n = struct.unpack("<I", raw[4:8])[0]
    out = {}
    for i in range(n):
        name, dt, nd, *rest = struct.unpack("<96sII6IQQ", raw[8 + i * 144: 8 + (i + 1) * 144])
        dims, off, nb = rest[:6][:nd], rest[6], rest[7]
        dtype = {0: np.uint16, 1: np.float32, 2: np.int32}[dt]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(raw, dtype=dtype, count=nb // np.dtype(dtype).itemsize, offset=off).reshape(dims)
    return out


def test_packer_layouts_the_kernels_rely_on():
    """pack_convnext_vit: layer scale folded into pwconv2, 1/8 folded into the query rows, the (2,1) down-sampler as K = 2C,
    and the fused-MLP layouts -- W1 rows as they are, W2 columns per 32-unit chunk in the MFMA accumulator order."""
    from pdf_table_amd.weights import pack_convnext_vit
    sd = convnext_vit_state_dict(seed=3)
    t = _read_blob(pack_convnext_vit(sd, x3=False))

    def bf(a):
        return torch.from_numpy(a.astype(np.int32) << 16).view(torch.float32) if False else \
            torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32))

    q = "cnn_model.encoder.stages.1.layers.2."
    w1, w2, g = sd[q + "pwconv1.weight"], sd[q + "pwconv2.weight"], sd[q + "layer_scale_parameter"]
    assert torch.equal(bf(t["s1.l2.mlp.w1"]), w1.to(torch.bfloat16).float())
    w2s = (w2.double() * g.double()[:, None]).float().to(torch.bfloat16).float()             # [192, 768]
    got = bf(t["s1.l2.mlp.w2p"])                                                            # [24 chunks][192][32]
    assert tuple(got.shape) == (24, 192, 32)
    for s2 in range(2):
        for half in range(2):
            for j in range(8):
                hidden = (j & 3) + 8 * (2 * s2 + (j >> 2)) + 4 * half
                assert torch.equal(got[5, :, s2 * 16 + half * 8 + j], w2s[:, 5 * 32 + hidden])
    np.testing.assert_allclose(t["s1.l2.pw2.b"], (sd[q + "pwconv2.bias"].double() * g.double()).float().numpy(), rtol=0, atol=0)
    assert "s3.l0.mlp.w1" not in t and "s2.l7.mlp.w1" in t and "vit.l11.mlp.w2p" in t       # C = 512 keeps the two-GEMM path
    # q rows carry 1/sqrt(64); biases too
    a = "vitstr.vit.encoder.layer.4.attention.attention."
    np.testing.assert_array_equal(t["vit.l4.qkv.b"][:192], (sd[a + "query.bias"] * 0.125).numpy())
    np.testing.assert_array_equal(t["vit.l4.qkv.b"][192:384], sd[a + "key.bias"].numpy())
    # tap-major depthwise weights, position table without the class-token slot, classifier padding
    np.testing.assert_array_equal(t["s0.l0.dw.w"], sd["cnn_model.encoder.stages.0.layers.0.dwconv.weight"].reshape(96, 49).t().numpy())
    np.testing.assert_array_equal(t["vit.pos"], sd["vitstr.vit.embeddings.position_embeddings"][0, 1:].numpy())
    assert t["cls.b"].shape == (7680,) and (t["cls.b"][7644:] == np.float32(-3.0e38)).all()
    # down-sampler: K index = row * Cin + c
    wd = sd["cnn_model.encoder.stages.2.downsampling_layer.1.weight"]                         # [256, 192, 2, 1]
    tiles = bf(t["s2.down.w"])                                                              # [N/64][K/32][1][64][32]
    flat = tiles.permute(0, 3, 1, 2, 4).reshape(256, 384)
    assert torch.equal(flat[:, :192], wd[:, :, 0, 0].to(torch.bfloat16).float()) and torch.equal(flat[:, 192:], wd[:, :, 1, 0].to(torch.bfloat16).float())
