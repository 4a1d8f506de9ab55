"""Pin oracle/mtl_tabnet.py (backbone of MtlTabNet) against golden vectors produced by the reference's own
TableResNetExtra module (tests/golden/make_golden.py::gen_mtl_tabnet_backbone)."""
import os

import numpy as np
import torch

from oracle import mtl_tabnet as omt
from pdf_table_amd.synth_weights import mtl_tabnet_backbone_state_dict


def test_backbone_oracle_matches_reference_module(golden_dir):
    g = np.load(os.path.join(golden_dir, "mtl_tabnet_backbone.npz"))
    sd = mtl_tabnet_backbone_state_dict(seed=int(g["seed"]))
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == int(g["n_params"])
    with torch.no_grad():
        f = omt.backbone_forward_fp32(sd, torch.from_numpy(g["x"]))
    assert [list(t.shape) for t in f] == g["shapes"].tolist() == [[1, 256, 48, 64], [1, 256, 24, 32], [1, 512, 12, 16]]
    for got, want, tag in ((f[0][:, ::8, ::3, ::3], g["f1_sub"], "f1"), (f[1][:, ::4, ::2, ::2], g["f2_sub"], "f2"), (f[2][:, ::4], g["f3"], "f3")):
        scale = np.abs(want).max()
        err = np.abs(got.numpy() - want).max()
        assert err <= 1e-4 * scale, (tag, err, scale)
    assert float(f[2].abs().mean()) > 1e-3            # not a dead network


def test_decoder_oracle_matches_reference_module(golden_dir):
    """greedy test-time decode (structure tokens, boxes, cell content) of oracle/mtl_tabnet.py against the reference's own
    MtlTabNetDecoder.forward(train_mode=False) on seeded weights (make_golden.py::gen_mtl_tabnet_decoder)"""
    from pdf_table_amd.synth_weights import mtl_tabnet_decoder_state_dict
    g = np.load(os.path.join(golden_dir, "mtl_tabnet_decoder.npz"))
    cfg = dict(N=3, sos=40, eos=41, pad=42, max_len=12, sos_cell=57, eos_cell=58, pad_cell=59, max_len_cell=6, idx_tag_cell=[3, 5])
    sd = mtl_tabnet_decoder_state_dict(seed=int(g["seed"]), num_classes=43, num_classes_cell=60)
    with torch.no_grad():
        feature = omt.positional_encoding(torch.from_numpy(g["fmap"]))
        np.testing.assert_allclose(feature[:, ::3, ::16].numpy(), g["feature_sub"], rtol=0, atol=1e-6)
        tag, box, cells = omt.greedy_decode(sd, feature, cfg)
    assert tuple(tag.shape) == g["tag_logits"].shape and tuple(box.shape) == g["boxes"].shape
    scale = np.abs(g["tag_logits"]).max()
    assert np.abs(tag.numpy() - g["tag_logits"]).max() <= 1e-4 * scale
    assert (tag.argmax(-1).numpy() == g["tag_logits"].argmax(-1)).all()
    assert np.abs(box.numpy() - g["boxes"]).max() <= 1e-5
    assert len(cells) == int(g["n_cells"])
    for i, cl in enumerate(cells):
        want = g[f"cell_{i}"]
        assert tuple(cl.shape) == want.shape
        assert np.abs(cl.numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    assert any(c.dim() == 3 for c in cells)            # the cell-content decoder really ran for one sample
