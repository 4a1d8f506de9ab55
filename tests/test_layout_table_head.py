"""The fitted stride-64 head branch of the synthetic PicoDet checkpoint (tools/fit_layout_head.py -> pdf_table_amd/data/picodet_synth_table_head.npz,
picodet_state_dict(table_head=True)): bench.py's layout -> table-structure chain.  CPU: the overlay itself and, through the oracle, that the chain's
hand-off (class "table", score >= 0.2) returns the generator's table on a fitted page."""
import numpy as np
import pytest
import torch

from oracle import picodet as op
from pdf_table_amd.layout_stage import LAYOUT_LABELS, layout_tables
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import picodet_state_dict


def test_overlay_replaces_only_the_stride64_branch():
    base = picodet_state_dict(seed=4, num_classes=5)
    fit = picodet_state_dict(seed=4, num_classes=5, table_head=True)
    assert list(base) == list(fit)
    table = LAYOUT_LABELS["en"].index("table")
    changed = {k for k in base if not torch.equal(base[k], fit[k])}
    assert changed and all(k.startswith("head.") for k in changed)
    for k in changed:
        assert "3_" in k or k.startswith("head.head_cls"), k         # cls_conv_dw3_i / cls_conv_pw3_i / head_cls*
    for lvl in range(3):      # other levels: only the table logit is switched off
        w0, w1 = base[f"head.head_cls{lvl}.weight"], fit[f"head.head_cls{lvl}.weight"]
        keep = [c for c in range(w0.shape[0]) if c != table]
        assert torch.equal(w0[keep], w1[keep]) and float(w1[table].abs().max()) == 0.0
        assert float(fit[f"head.head_cls{lvl}.bias"][table]) == -12.0
    with pytest.raises(ValueError):
        picodet_state_dict(seed=5, num_classes=5, table_head=True)


def test_oracle_finds_the_generators_table_on_a_fitted_page():
    sd = {k: v.float() for k, v in picodet_state_dict(seed=4, num_classes=5, table_head=True).items()}
    img, meta = make_page(0)
    x, sf = op.picodet_preprocess(img)
    with torch.no_grad():
        sc, bx = op.picodet_forward(sd, torch.from_numpy(x)[None], 5)
    res = op.picodet_postprocess([s.numpy() for s in sc], [b.numpy() for b in bx], img.shape[:2], sf, (800, 608), LAYOUT_LABELS["en"])
    tabs = layout_tables(res, "table", 0.2)
    gt = np.asarray(meta["tables"], dtype=np.float64).reshape(-1, 4) + np.array([-8, -8, 8, 8])       # the fit's target: the ruled grid grown by 8 px
    assert len(tabs) == len(gt) == 1
    assert np.abs(np.asarray(tabs[0]["bbox"], dtype=np.float64) - gt[0]).max() <= 16.0
    assert float(tabs[0]["score"]) > 0.6          # a quality score that peaks below 0.95 at the anchor nearest the centre (tools/fit_layout_head.py)
