"""TableMaster (reference table/mtl_tabnet/master_decoder.py:532-645, table_master_config.py, master_convertor.py:787-1070): the
MtlTabNet machinery without the cell-content decoder.  CPU half: the oracle's ``table_master_decode`` against the reference's own ``TableMasterDecoder``
(tests/golden/table_master_decoder.npz, ``make_golden.py::gen_table_master``), the host half (``TableMasterConvertor`` + ``MasterPostProcessor``) against
what the reference's classes made of the same logits, and the weight blob of a decoder without cell tensors.  GPU half: tests/test_gpu_mtl.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mtl_tabnet as omt
from pdf_table_amd.mtl_stage import MasterPostProcessor, TableMasterConvertor
from pdf_table_amd.synth_weights import table_master_decoder_state_dict


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, "table_master_decoder.npz"))
    sos, eos, pad, ncls = (int(v) for v in g["ids"])
    cfg = dict(N=3, sos=sos, eos=eos, pad=pad, max_len=int(g["max_len"]), idx_tag_cell=[0, 0])
    return g, cfg, ncls, json.loads(str(g["host_json"]))


def test_oracle_equals_the_reference_decoder(gold):
    g, cfg, ncls, _ = gold
    sd = table_master_decoder_state_dict(seed=int(g["seed"]), num_classes=ncls)
    assert not any(k.startswith(("cell_", "embedding_cell")) for k in sd)
    with torch.no_grad():
        out, box = omt.table_master_decode(sd, omt.positional_encoding(torch.from_numpy(g["fmap"])), cfg)
    assert out.shape == g["tag_logits"].shape == (3, cfg["max_len"] + 1, ncls)
    assert np.abs(out.numpy() - g["tag_logits"]).max() <= 2e-4 * np.abs(g["tag_logits"]).max()
    assert np.abs(box.numpy() - g["boxes"]).max() <= 1e-5
    assert (out.argmax(-1).numpy() == g["tag_logits"].argmax(-1)).all()
    assert (g["tag_logits"].argmax(-1) == cfg["pad"]).any()      # the fixture exercises the non-causal <PAD> mask


def test_host_half_equals_the_reference_classes(gold):
    g, cfg, ncls, host = gold
    conv = TableMasterConvertor(max_seq_len=500)
    assert (conv.start_idx, conv.end_idx, conv.padding_idx, conv.num_classes(), conv.num_classes_cell()) == (cfg["sos"], cfg["eos"], cfg["pad"], ncls, 0)
    assert set(conv.decoder_cfg()) == {"N", "sos", "eos", "pad", "max_len", "idx_tag_cell"}
    meta = {"scale_factor": (0.75, 0.75), "pad_shape": (480, 480, 3), "ori_shape": (400, 640, 3), "img_shape": (300, 480, 3)}
    seen = set()
    for b, want in enumerate(host):
        strings, scores, bboxes = conv.output_format(g["tag_logits"][b:b + 1], g["boxes"][b:b + 1], [meta])
        assert strings[0] == want["text"] and abs(scores[0] - want["score"]) <= 1e-6
        assert np.array_equal(np.asarray(bboxes[0]), np.asarray(want["bbox_decoded"]))
        result = dict(text=strings[0], score=scores[0], bbox=bboxes[0])
        if want.get("raises") == "IndexError":
            with pytest.raises(IndexError):
                MasterPostProcessor(strict=True)(result)
            seen.add("raises")
            continue
        pred = MasterPostProcessor(strict=True)(result)
        assert pred["html_context"] == want["html_context"] and pred["structure_str"] == want["structure_str"]
        assert pred["structure_str_list"] == want["structure_str_list"] and np.array_equal(pred["new_bbox"], np.asarray(want["new_bbox"]))
        seen.add("ok")
    assert seen == {"ok", "raises"}


def test_blob_of_a_decoder_without_cell_tensors(gold):
    from pdf_table_amd.weights import pack_mtl_decoder
    from test_oracle_convnext_vit import _read_blob
    g, cfg, ncls, _ = gold
    sd = table_master_decoder_state_dict(seed=int(g["seed"]), num_classes=ncls)
    t = _read_blob(pack_mtl_decoder(sd, cfg))
    assert not any(k.startswith(("cell", "emb_cell")) for k in t) and "cls.qkv.w" in t and "bbox.ff2.w3" in t
    meta = t["meta"].ravel()
    assert meta[0] == ncls and meta[1] == 0 and list(meta[6:10]) == [0, 0, 0, 0]
    # the key / value projection keeps the five-slot channel layout; the absent layer's slot is zero weights
    kv_b = t["kv.b"].ravel()
    assert kv_b.shape[0] == 5 * 1024 and not kv_b[4 * 1024:].any() and kv_b[:4 * 1024].any()
