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
