"""Pin oracle/db_net.py against golden vectors produced by the reference's DBModel itself
(tests/golden/make_golden.py::gen_db_resnet18, reference model/db_net/dbnet.py:715-728)."""
import os

import numpy as np
import torch

from oracle import db_net
from pdf_table_amd.synth_weights import db_resnet18_state_dict


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "db_resnet18.npz"))


def test_fp32_oracle_matches_reference_module(golden_dir):
    g = _load(golden_dir)
    sd = db_resnet18_state_dict(seed=int(g["seed"]))
    for tag in ("a", "b"):
        x = torch.from_numpy(g[f"x_{tag}"])
        with torch.no_grad():
            c2, _, _, c5 = db_net.db_backbone_fp32(sd, x)
            prob = db_net.db_forward_fp32(sd, x)
        # same ops in the same order -> identical up to conv algorithm selection noise
        np.testing.assert_allclose(c2.numpy(), g[f"c2_{tag}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c5.numpy(), g[f"c5_{tag}"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(prob.numpy(), g[f"prob_{tag}"], rtol=0, atol=1e-5)


def test_bf16_contract_stays_close_to_fp32(golden_dir):
    """The engine's bf16 contract is a different arithmetic; record how far it sits from fp32."""
    g = _load(golden_dir)
    sd = db_resnet18_state_dict(seed=int(g["seed"]))
    x = torch.from_numpy(g["x_b"])
    with torch.no_grad():
        p16 = db_net.db_forward_bf16(sd, x)
    err = np.abs(p16.numpy() - g["prob_b"]).max()
    assert err < 0.08, err  # bf16 activations: ~1e-2 class deviation on a prob map (measured 0.0x), not 1e-3


def test_fold_bn_equals_conv_then_bn():
    sd = db_resnet18_state_dict(seed=3)
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    import torch.nn.functional as F
    ref = db_net._bn(sd, "backbone.bn1", F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3))
    s = sd["backbone.bn1.weight"].double() / torch.sqrt(sd["backbone.bn1.running_var"].double() + 1e-5)
    w = (sd["backbone.conv1.weight"].double() * s.view(-1, 1, 1, 1)).float()
    b = (sd["backbone.bn1.bias"].double() - sd["backbone.bn1.running_mean"].double() * s).float()
    got = F.conv2d(x, w, b, 2, 3)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


# ---- DB-ProxylessNAS (`DBNasModel`, db_net/dbnet.py:693-712) ------------------------------------------------------
def test_nas_oracle_matches_reference_module(golden_dir):
    """oracle/db_nas.py against tensors the reference's own DBNasModel produced (tests/golden/make_golden.py db_nas)."""
    from oracle import db_nas
    from pdf_table_amd.synth_weights import db_nas_state_dict
    g = np.load(os.path.join(golden_dir, "db_nas.npz"))
    sd = db_nas_state_dict(seed=int(g["seed"]))
    for tag in ("a", "b"):
        x = torch.from_numpy(g[f"x_{tag}"])
        with torch.no_grad():
            c2, _, _, c5 = db_nas.dbnas_backbone_fp32(sd, x)
        prob = db_nas.dbnas_forward_fp32(sd, x).numpy()
        assert np.abs(c2.numpy() - g[f"c2_{tag}"]).max() <= 1e-5
        assert np.abs(c5.numpy() - g[f"c5_{tag}"]).max() <= 1e-5
        assert np.abs(prob - g[f"prob_{tag}"]).max() <= 1e-6


def test_nas_synthetic_checkpoint_has_the_reference_inventory(golden_dir):
    """key names, order and shapes of the synthetic checkpoint == DBNasModel().state_dict() of the reference"""
    from pdf_table_amd.synth_weights import db_nas_state_dict
    g = np.load(os.path.join(golden_dir, "db_nas.npz"))
    sd = db_nas_state_dict(seed=1)
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == [str(s) for s in g["shapes"]]


def test_nas_packer_folding_is_exact():
    """pack_db_nas's algebra on one `rep` block: sum of (depthwise k x k + BN) branches == one centred 5x5 depthwise
    conv + bias (what the blob holds), and the DwPwConvTranspose tail table reproduces the module arithmetic."""
    import struct
    import torch.nn.functional as F
    from oracle import db_nas
    from pdf_table_amd.synth_weights import db_nas_state_dict
    from pdf_table_amd.weights import pack_db_nas
    sd = db_nas_state_dict(seed=3)
    blob = pack_db_nas(sd, x3=False)
    n = struct.unpack("<I", blob[4:8])[0]
    tens = {}
    for i in range(n):
        name, dt, nd, *rest = struct.unpack("<96sII6IQQ", blob[8 + i * 144: 8 + (i + 1) * 144])
        dims, off, nb = rest[:6], rest[6], rest[7]
        tens[name.rstrip(b"\0").decode()] = (dt, dims[:nd], off, nb)

    def f32(name):
        _, dims, off, nb = tens[name]
        return torch.from_numpy(np.frombuffer(blob, dtype=np.float32, count=nb // 4, offset=off).reshape(dims).copy())

    p = "backbone.blocks.1.mobile_inverted_conv"        # '135_RepConv2': 1x1 + 3x3 + 5x5 branches, 64 channels
    g = torch.Generator().manual_seed(0)
    y = torch.randn(1, 64, 12, 10, generator=g)
    ref = None
    for ri, k in enumerate((1, 3, 5)):
        t = db_nas._bn(sd, f"{p}.rep_conv.{ri}.bn", F.conv2d(y, sd[f"{p}.rep_conv.{ri}.conv.weight"], None, 1, k // 2, 1, 64))
        ref = t if ref is None else ref + t
    w = f32("b1.dw.wf32").reshape(5, 5, 64).permute(2, 0, 1).unsqueeze(1)
    got = F.conv2d(y, w, f32("b1.dw.b"), 1, 2, 1, 64)
    assert (got - ref).abs().max().item() <= 2e-5
    # tail: y16 (after ReLU) -> logits
    y16 = torch.rand(1, 16, 3, 4, generator=g)
    z = F.relu(db_nas._bn(sd, "decoder.binarize.4", db_nas._dwpw_t(sd, "decoder.binarize.3", y16)))
    ref = db_nas._dwpw_t(sd, "decoder.binarize.6", z)[0, 0]
    tw = f32("dec.tail")
    W1, B1, P1, pb1 = tw[0:64].view(4, 16), tw[64:80], tw[80:336].view(16, 16), tw[336:352]
    W2, B2, P2, pb2 = tw[352:416].view(4, 16), tw[416:432], tw[432:448], tw[448]
    out = torch.zeros(12, 16)
    for yy in range(3):
        for xx in range(4):
            v = y16[0, :, yy, xx]
            for q in range(4):
                u = F.relu(P1 @ F.relu(v * W1[q] + B1) + pb1)
                for r in range(4):
                    out[4 * yy + 2 * (q >> 1) + (r >> 1), 4 * xx + 2 * (q & 1) + (r & 1)] = P2 @ F.relu(u * W2[r] + B2) + pb2
    assert (out - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
