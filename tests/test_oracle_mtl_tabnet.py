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
