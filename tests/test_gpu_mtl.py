"""GPU parity of the MtlTabNet backbone (second half) through the C ABI against the fp32 oracle, which is
pinned to the reference's own TableResNetExtra (tests/test_oracle_mtl_tabnet.py).  The decoders are not on the engine yet."""
import os

import numpy as np
import pytest
import torch

from oracle import mtl_tabnet as omt
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import mtl_tabnet_backbone_state_dict
from pdf_table_amd.weights import pack_mtl_backbone

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_MTL_BACKBONE, pack_mtl_backbone(mtl_tabnet_backbone_state_dict(seed=41)))
    yield e
    e.close()


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_backbone_matches_oracle_and_reference_golden(eng, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "mtl_tabnet_backbone.npz"))
    sd = mtl_tabnet_backbone_state_dict(seed=int(g["seed"]))
    rng = np.random.default_rng(9)
    x = np.concatenate([g["x"], rng.standard_normal((1, 3, 96, 128)).astype(np.float32)])      # the golden's input + one more image
    with torch.no_grad():
        want = omt.backbone_forward_fp32(sd, torch.from_numpy(x))[2]
    eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
    try:
        got = eng.mtl_backbone_forward(torch.from_numpy(x).cuda()).cpu()
        one = eng.mtl_backbone_forward(torch.from_numpy(x[1:2]).cuda()).cpu()
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    assert tuple(got.shape) == (2, 512, 12, 16)
    assert torch.equal(got[1:2], one), "an image's features do not depend on its batch"
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    gerr = np.abs(got[0:1, ::4].numpy() - g["f3"]).max()
    print(f"mtl backbone {mode}: max|d f3| = {err:.3e} = {err / scale:.2e} of scale {scale:.2f}; vs the reference module's own output {gerr / scale:.2e}")
    if mode == "bf16x3":
        assert err <= 1e-3 * scale and gerr <= 1e-3 * scale
    else:
        assert err <= 0.1 * scale


def test_missing_weights_fail_loudly():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    try:
        with pytest.raises(L.PtError, match="MtlTabNet backbone weights not loaded"):
            e.mtl_backbone_forward(torch.zeros((1, 3, 32, 32), device="cuda"))
    finally:
        e.close()
