"""CPU-side checks of round 5's host logic (no GPU): the fp16 weight blobs of PT_PRECISION_F16, the generated C-ABI dispatcher, the conditioned checkpoint
set shared by bench.py and the end-to-end fixture, and the fixtures added this round."""
import os
import re
import struct

import numpy as np
import torch

from pdf_table_amd import weights as Wt

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tensors(blob):
    assert blob[:4] == b"PTW1"
    n = struct.unpack("<I", blob[4:8])[0]
    out = {}
    for i in range(n):
        rec = struct.unpack("<96sII6IQQ", blob[8 + i * 144: 8 + (i + 1) * 144])
        name = rec[0].split(b"\0")[0].decode()
        dims = [d for d in rec[3:9][:rec[2]]]
        out[name] = (rec[1], dims, rec[9], rec[10])
    return out


def test_f16_blob_is_marked_single_pass_and_rounds_like_torch_half():
    from pdf_table_amd.synth_weights import crnn_state_dict
    sd = crnn_state_dict(seed=1)
    b16, bbf = Wt.pack_crnn(sd, fmt="f16"), Wt.pack_crnn(sd, x3=False)
    t16, tbf = _tensors(b16), _tensors(bbf)
    assert "__act_f16__" in t16 and "__act_f16__" not in tbf
    assert not any(k.endswith((".w3", ".wh")) for k in t16), "an fp16 blob carries no pair-mode tiles"
    assert set(t16) - {"__act_f16__"} == set(tbf)
    # one conv tile: the fp16 rounding of the folded weight, tiled like the bf16 one
    w, _ = Wt.fold_conv_bn(sd, "conv1.0", "conv1.1")
    _, dims, off, nb = t16["conv1.w"]
    got = np.frombuffer(b16, np.uint16, nb // 2, off).reshape(dims)
    assert np.array_equal(got, Wt.tile_conv_weight(w, "f16"))
    n, cin = w.shape[:2]
    back = torch.from_numpy(got.view(np.int16).copy()).view(torch.float16).float().reshape(n // 64, cin // 32, 9, 64, 32)
    ref = w.reshape(n // 64, 64, cin // 32, 32, 3, 3).permute(0, 2, 4, 5, 1, 3).reshape(n // 64, cin // 32, 9, 64, 32).to(torch.float16).float()
    assert torch.equal(back, ref)
    # the 16-bit rounding held in fp32 for the first conv follows the format too
    o = t16["conv0.wbf"]
    v = np.frombuffer(b16, np.float32, o[3] // 4, o[2])
    assert np.array_equal(v, torch.from_numpy(v.copy()).to(torch.float16).float().numpy())


def test_every_packer_takes_fmt():
    import inspect
    for name in Wt.__all__:
        if name.startswith("pack_"):
            assert "fmt" in inspect.signature(getattr(Wt, name)).parameters, name


def test_dispatcher_covers_the_header_and_picks_the_namespace_by_precision():
    from pdf_table_amd import build as B
    src = B.dispatch_source()
    hdr = open(os.path.join(REPO, "include", "pdftable_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    once = {"pt_last_error", "pt_abi_version", "pt_db_candidates", "pt_db_finalize", "pt_db_candidates_batch", "pt_db_finalize_batch", "pt_hard_nms"}
    body = src[src.index('extern "C"'):]
    defined = set(re.findall(r"\b(pt_[a-z0-9_]+)\(", body)) & declared
    assert defined == declared - once
    # an entry point with the engine first dispatches on its precision; a plan function runs the bf16 namespace's copy
    assert re.search(r"int pt_det_forward\(pt_engine\* e, [^)]*\) \{ return \(e && e->precision == PT_PRECISION_F16\) \? pt_f16::api::pt_det_forward\(e, ", body)
    assert re.search(r"int pt_det_plan\([^)]*\) \{ return pt_bf16::api::pt_det_plan\(", body)
    assert "void pt_engine_destroy(pt_engine* e) { (e && e->precision == PT_PRECISION_F16) ? pt_f16::api::pt_engine_destroy(e) : pt_bf16::api::pt_engine_destroy(e); }" in body


def test_conditioned_checkpoint_set_and_fixtures():
    from pdf_table_amd.synth_weights import CRNN_NUM_CLASSES, conditioned_state_dicts, crnn_state_dict
    sds = conditioned_state_dicts()
    assert set(sds) == {"db", "crnn", "pico", "lore", "proc"}
    w = sds["crnn"]["cls.weight"]
    assert tuple(w.shape) == (CRNN_NUM_CLASSES, 512)
    used = (w.abs().sum(1) > 0).nonzero().flatten()
    z = np.load(os.path.join(REPO, "pdf_table_amd", "data", "crnn_synth_classifier.npz"))
    assert used.tolist() == sorted(int(i) for i in z["ids"]) and 0 in used.tolist()      # the fitted rows, CTC blank among them
    for k, v in crnn_state_dict(seed=1).items():                                             # everything but the classifier is the seeded net
        if k != "cls.weight":
            assert torch.equal(v, sds["crnn"][k]), k
    g = np.load(os.path.join(REPO, "tests", "golden", "e2e_page.npz"))
    for pi in range(2):
        assert len(g[f"p{pi}_rec_text"]) == len(g[f"p{pi}_det_boxes"])
        for ti in range(int(g[f"p{pi}_n_tables"])):
            html = [str(x) for x in g[f"p{pi}_t{ti}_html"]]
            assert html[0].startswith("<table") and html[-1] == "</table>" and sum(h == "<tr>" for h in html) >= 3
    m = np.load(os.path.join(REPO, "tests", "golden", "mtl_tabnet_lengths.npz"))
    assert m["tag0"].shape == (501, 43) and m["cells0"].shape[1:] == (151, 60)
