"""PP-OCR CTCLabelDecode: oracle and product both pinned to outputs of the reference class itself
(tests/golden/ctc_decode.{npz,json}, make_golden.py::gen_ctc)."""
import json
import os

import numpy as np

from oracle import crnn as ocrnn
from pdf_table_amd.rec_postprocess import CTCLabelDecode


def _golden(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "ctc_decode.json")))
    probs = np.load(os.path.join(golden_dir, "ctc_decode.npz"))["probs"]
    return meta, probs


def test_oracle_and_product_match_reference(golden_dir):
    meta, probs = _golden(golden_dir)
    chars = meta["chars"] + [" "]
    assert ["blank"] + chars == meta["character"]
    want = list(zip(meta["texts"], meta["confs"]))
    got_o = ocrnn.ctc_label_decode(probs, chars)
    dec = CTCLabelDecode(characters=chars)
    got_p = dec(probs)
    got_ids = dec.decode_ids(probs.argmax(2), probs.max(2))      # the engine's call shape
    for (t, c), o, p, q in zip(want, got_o, got_p, got_ids):
        assert o[0] == t and p[0] == t and q[0] == t
        assert o[1] == c and p[1] == c and q[1] == c
    assert any(len(t) > 0 for t, _ in want)


def test_edge_cases():
    dec = CTCLabelDecode(characters=list("ab"))
    ids = np.array([[0, 0, 0, 0], [1, 1, 0, 1], [2, 1, 1, 2]])
    pr = np.full(ids.shape, 0.5, np.float32)
    res = dec.decode_ids(ids, pr)
    assert res[0] == ("", 0.0)                  # empty -> conf_list [0]
    assert res[1][0] == "aa" and res[2][0] == "bab"
    assert CTCLabelDecode.pred_reverse("ab12بةx") == "xةبab12"
