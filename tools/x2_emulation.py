"""CPU emulation of candidate arithmetic for the tolerance mode: DB-ResNet18 on one 1024x1024 page at 960x960 with
activations / weights rounded per layer as the engine would store them, fp32 accumulate; distance of the logits from the fp32 oracle.

    python tools/x2_emulation.py

a = what an activation keeps after every layer, w = what a folded weight keeps.  'pair16' = (hi, lo) fp16 pair (hi = fp16(x), lo = fp16(x - hi)),
'pairbf' = (hi, lo) bf16 pair.  MFMA passes per product: single x single = 1, pair x single = 2, pair x pair = 3 (lo*lo dropped)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import db_net, db_pre                                   # noqa: E402
from pdf_table_amd.synth_pages import make_page                    # noqa: E402
from pdf_table_amd.synth_weights import db_resnet18_state_dict     # noqa: E402

f16 = lambda t: t.to(torch.float16).to(torch.float32)
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
ROUND = {"fp32": lambda t: t, "fp16": f16, "bf16": bf, "pair16": lambda t: f16(t) + f16(t - f16(t)), "pairbf": lambda t: bf(t) + bf(t - bf(t))}


def run(sd, x, a, w):
    orig = db_net.fold_bn

    def fold(*args, **kw):
        db_net.bf16_round = ROUND[w]
        try:
            return orig(*args, **kw)
        finally:
            db_net.bf16_round = ROUND[a]
    db_net.fold_bn = fold
    db_net.bf16_round = ROUND[a]
    try:
        with torch.no_grad():
            return db_net.db_forward_bf16(sd, x, return_logits=True)[0, 0]
    finally:
        db_net.fold_bn, db_net.bf16_round = orig, bf


if __name__ == "__main__":
    torch.set_num_threads(8)
    sd = db_resnet18_state_dict(seed=0, text_signal=True) if "text_signal" in db_resnet18_state_dict.__code__.co_varnames else db_resnet18_state_dict(seed=0)
    img = make_page(0, 1024)[0]
    chw, _ = db_pre.preprocess_db_pp(img)
    x = torch.from_numpy(np.ascontiguousarray(chw))[None]
    with torch.no_grad():
        ref = db_net.db_forward_fp32(sd, x, return_logits=True)[0, 0]
    scale = max(1.0, ref.abs().max().item())
    for a, w, passes in [("bf16", "bf16", 1), ("fp16", "fp16", 1), ("pair16", "fp16", 2), ("fp32", "fp16", 2), ("pairbf", "bf16", 2), ("pair16", "pair16", 3),
                         ("pairbf", "pairbf", 3), ("fp16", "pair16", 2)]:
        y = run(sd, x, a, w)
        d = (y - ref).abs().max().item()
        dp = (torch.sigmoid(y) - torch.sigmoid(ref)).abs().max().item()
        print(f"a={a:7s} w={w:7s} passes={passes}: max|dlogit| = {d:.3e} = {d / scale:.2e} of scale {scale:.1f}; max|dprob| = {dp:.2e}", flush=True)
