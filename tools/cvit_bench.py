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
a = ap.parse_args()
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(convnext_vit_state_dict(1)))
if a.x3:
    eng.set_precision(L.PT_PRECISION_BF16X3)
g = torch.rand((a.lines, 32, 804), device="cuda")
eng.rec_cvit_forward_net(g)
torch.cuda.synchronize()
t = time.time()
for _ in range(a.steps):
    eng.rec_cvit_forward_net(g)
torch.cuda.synchronize()
dt = (time.time() - t) / a.steps
gflop_line = 12.3
print(f"convnext-vit {'bf16x3' if a.x3 else 'bf16'}: {a.lines} lines in {dt * 1e3:.1f} ms = {a.lines / dt:.0f} lines/s "
      f"(~{a.lines / dt * gflop_line / 1e3:.0f} TFLOP/s of GEMM work)")
