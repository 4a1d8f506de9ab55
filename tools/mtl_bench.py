#!/usr/bin/env python
"""MtlTabNet backbone throughput on synthetic 480x480 tables: python tools/mtl_bench.py [--tables 16] [--x3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd import lib as L  # noqa: E402
from pdf_table_amd.engine import HipEngine  # noqa: E402
from pdf_table_amd.synth_weights import mtl_tabnet_backbone_state_dict  # noqa: E402
from pdf_table_amd.weights import pack_mtl_backbone  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=16)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--x3", action="store_true")
a = ap.parse_args()
eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_MTL_BACKBONE, pack_mtl_backbone(mtl_tabnet_backbone_state_dict(1), x3=a.x3))
if a.x3:
    eng.set_precision(L.PT_PRECISION_BF16X3)
x = torch.randn((a.tables, 3, 480, 480), device="cuda")
eng.mtl_backbone_forward(x)
torch.cuda.synchronize()
t = time.time()
for _ in range(a.steps):
    eng.mtl_backbone_forward(x)
torch.cuda.synchronize()
dt = (time.time() - t) / a.steps
gflop = 2 * 9 * (480 * 480 * (3 * 64 + 64 * 128) + 240 * 240 * (128 * 256 + 256 * 256 * 2) + 120 * 120 * 256 * 256 * 5
                 + 60 * 60 * (256 * 512 + 512 * 512 * 17)) / 1e9
print(f"mtl backbone {'bf16x3' if a.x3 else 'bf16'}: {a.tables} tables of 480x480 in {dt * 1e3:.1f} ms = {a.tables / dt:.0f} tables/s "
      f"(~{gflop:.0f} GFLOP of 3x3 convs per table: {a.tables / dt * gflop / 1e3:.0f} TFLOP/s incl. the host-side layout conversion)")
