#!/usr/bin/env python
"""ConvNextViT recogniser throughput (lines/s) on synthetic lines: python tools/cvit_bench.py [--lines 2048] [--steps 5] [--x3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd import lib as L  # noqa: E402
from pdf_table_amd.engine import HipEngine  # noqa: E402
from pdf_table_amd.synth_weights import convnext_vit_state_dict  # noqa: E402
from pdf_table_amd.weights import pack_convnext_vit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lines", type=int, default=2048)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--x3", action="store_true")
ap.add_argument("--ragged", action="store_true", help="text widths like the synthetic pages' lines (median 148 px): all-padding chunks are shared")
a = ap.parse_args()
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(convnext_vit_state_dict(1)))
if a.x3:
    eng.set_precision(L.PT_PRECISION_BF16X3)
g = torch.rand((a.lines, 32, 804), device="cuda")
tw = None
if a.ragged:
    import numpy as np
    rng = np.random.default_rng(0)
    tw = np.minimum(804, (32 * np.exp(rng.normal(1.53, 1.0, a.lines))).astype(np.int64)).clip(8).tolist()
    for i, w in enumerate(tw):
        g[i, :, w:] = 0
    print("chunks with text per line:", sum((w > 0) + (w > 252) + (w > 504) for w in tw) / a.lines)
eng.rec_cvit_forward_net(g, text_w=tw)
torch.cuda.synchronize()
t = time.time()
for _ in range(a.steps):
    eng.rec_cvit_forward_net(g, text_w=tw)
torch.cuda.synchronize()
dt = (time.time() - t) / a.steps
frac = 1.0 if tw is None else sum((w > 0) + (w > 252) + (w > 504) for w in tw) / (3.0 * a.lines)
gflop_line = 0.59 + 11.7 * frac          # classifier per line + CNN / ViT per chunk that is computed
print(f"convnext-vit {'bf16x3' if a.x3 else 'bf16'}: {a.lines} lines in {dt * 1e3:.1f} ms = {a.lines / dt:.0f} lines/s "
      f"(~{a.lines / dt * gflop_line / 1e3:.0f} TFLOP/s of GEMM work)")
