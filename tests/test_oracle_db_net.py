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
