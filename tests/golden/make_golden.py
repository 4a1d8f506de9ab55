#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REFERENCE's own Python modules.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference package cannot be imported as a whole (``import pdftable`` needs pdfminer, cv2,
onnxruntime ... -- SURVEY.md section 8c), so single files are imported by path underneath empty
stand-in parent packages (only ``__path__`` is set; no reference ``__init__`` runs).  Only DATA is
written here: seeded inputs and the tensors the reference modules produce for them.  Weights are
NOT stored; they are regenerated from a seed by ``pdf_table_amd.synth_weights`` and loaded into
the reference modules with ``strict=True`` (which also pins every key name and shape).
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
sys.path.insert(0, REPO)


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def ref_import(modname):
    """import pdftable.<...> by file without running any package __init__."""
    parts = modname.split(".")
    for i in range(1, len(parts)):
        pk = ".".join(parts[:i])
        if pk not in sys.modules:
            _pkg(pk, os.path.join(REF_SRC, *parts[:i]))
    return importlib.import_module(modname)


def gen_db_resnet18():
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    dbnet = ref_import("pdftable.model.db_net.dbnet")
    torch.manual_seed(0)
    model = dbnet.DBModel().eval()
    sd = db_resnet18_state_dict(seed=11)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(101)
    out = {}
    for tag, (h, w) in {"a": (64, 96), "b": (128, 128)}.items():
        x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            xt = torch.from_numpy(x)
            c2, c3, c4, c5 = model.backbone(xt)
            y = model(xt)
        out[f"x_{tag}"] = x
        out[f"prob_{tag}"] = y.numpy()
        out[f"c2_{tag}"] = c2.numpy()
        out[f"c5_{tag}"] = c5.numpy()
    out["seed"] = np.array(11)
    np.savez_compressed(os.path.join(HERE, "db_resnet18.npz"), **out)
    print("db_resnet18.npz", {k: v.shape for k, v in out.items()})


def gen_crnn():
    from pdf_table_amd.synth_weights import crnn_state_dict
    crnn = ref_import("pdftable.model.crnn.modeling_crnn")
    model = crnn.CRNN().eval()
    sd = crnn_state_dict(seed=12)
    model.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(102)
    x = rng.uniform(0, 1, (2, 3, 32, 96)).astype(np.float32)
    with torch.no_grad():
        y = model(torch.from_numpy(x))
    # logits are [2, 24, 7644]; keep every 16th class column + argmax to stay small
    ynp = y.numpy()
    np.savez_compressed(os.path.join(HERE, "crnn.npz"), x=x, seed=np.array(12),
                        logits_sub=ynp[:, :, ::16], argmax=ynp.argmax(-1).astype(np.int32),
                        maxval=ynp.max(-1))
    print("crnn.npz logits", ynp.shape)


def gen_registry_hash():
    cfg = ref_import("pdftable.model.ocr_pdf.ocr_table_model_config")
    blob = json.dumps(cfg.TABLE_MODEL_DICT, sort_keys=True, ensure_ascii=False).encode("utf-8")
    leaves = []

    def walk(d, path):
        if isinstance(d, dict):
            for k, v in d.items():
                walk(v, path + (str(k),))
        else:
            leaves.append(("/".join(path), d))
    walk(cfg.TABLE_MODEL_DICT, ())
    info = {"sha256": hashlib.sha256(blob).hexdigest(), "n_leaves": len(leaves),
            "providers": sorted(cfg.TABLE_MODEL_DICT.keys())}
    with open(os.path.join(HERE, "registry_hash.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print("registry", info)


def gen_ctc():
    """CTCLabelDecode known answers (ocr_rec_pp/rec_postprocess.py:126-191)."""
    rp = ref_import("pdftable.model.ocr_rec_pp.rec_postprocess")
    chars = list("abcdefghij")
    dict_path = os.path.join(HERE, "_tmp_dict.txt")
    with open(dict_path, "w") as f:
        f.write("\n".join(chars) + "\n")
    dec = rp.CTCLabelDecode(character_dict_path=dict_path, use_space_char=True)
    rng = np.random.default_rng(103)
    ncls = len(dec.character)
    probs = rng.uniform(0, 1, (6, 40, ncls)).astype(np.float32)
    # make repeats and blanks frequent
    for b in range(6):
        for t in range(40):
            if rng.uniform() < 0.35:
                probs[b, t, 0] = 2.0
            elif t > 0 and rng.uniform() < 0.4:
                probs[b, t] = probs[b, t - 1]
    res = dec(probs)
    os.remove(dict_path)
    with open(os.path.join(HERE, "ctc_decode.json"), "w") as f:
        json.dump({"chars": chars, "character": dec.character,
                   "texts": [r[0] for r in res], "confs": [float(r[1]) for r in res]}, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "ctc_decode.npz"), probs=probs)
    print("ctc", [r[0] for r in res])


if __name__ == "__main__":
    which = sys.argv[1:] or ["db", "crnn", "registry", "ctc"]
    if "db" in which:
        gen_db_resnet18()
    if "crnn" in which:
        gen_crnn()
    if "registry" in which:
        gen_registry_hash()
    if "ctc" in which:
        gen_ctc()
