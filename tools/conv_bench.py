#!/usr/bin/env python
"""Micro-benchmark of one conv shape through pt_op_conv2d (for rocprofv3 --pmc runs and A/B of kernel variants).
    python tools/conv_bench.py B H W Cin N [ks] [stride] [iters]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.weights import tile_conv_weight

B, H, W, Cin, N = [int(a) for a in sys.argv[1:6]]
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 3
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
eng = HipEngine(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
w = torch.randn(N, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
if os.environ.get("PT_BENCH_ZERO"):      # zero operands: the same instruction stream at the lowest switching power (DVFS probe)
    x.zero_()
    w.zero_()
wt = torch.from_numpy(tile_conv_weight(w).view(np.int16)).cuda()
b = torch.zeros(N).cuda()
res = None
if os.environ.get("PT_BENCH_RES"):       # + residual of the output's shape (the second conv of a BasicBlock)
    pad_ = ks // 2
    res = torch.randn(B, (H + 2 * pad_ - ks) // stride + 1, (W + 2 * pad_ - ks) // stride + 1, N, generator=g).to(torch.bfloat16).cuda()
out = eng.op_conv2d(x, wt, b, ks, stride, relu=True, res=res, res_mode=1 if res is not None else 0)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(iters):
    eng.op_conv2d(x, wt, b, ks, stride, relu=True, out=out, res=res, res_mode=1 if res is not None else 0)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / iters
pad = ks // 2
Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
fl = 2.0 * B * Ho * Wo * N * Cin * ks * ks
print(f"conv {ks}x{ks} s{stride} {Cin}->{N} @{H}x{W} B={B}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s  (variant={os.environ.get('PT_CONV_VARIANT','0')})")
