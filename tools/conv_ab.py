#!/usr/bin/env python
"""A/B of two conv kernel selections inside ONE process (interleaved rounds, median + min), bit-compare of their outputs.
    python tools/conv_ab.py ENV=a,b B H W Cin N [ks] [stride] [rounds] [iters]     e.g.  PT_CONV_PIPE=0,1 64 64 64 256 256
The environment variable must be one the launcher reads per call (PT_CONV_PIPE, PT_CONV_WS64, PT_CONV_XP ...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.weights import tile_conv_weight

var, vals = sys.argv[1].split("=")
vals = vals.split(",")
B, H, W, Cin, N = [int(a) for a in sys.argv[2:7]]
ks = int(sys.argv[7]) if len(sys.argv) > 7 else 3
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
rounds = int(sys.argv[9]) if len(sys.argv) > 9 else 7
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 10
eng = HipEngine(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
w = torch.randn(N, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5
wt = torch.from_numpy(tile_conv_weight(w).view(np.int16)).cuda()
b = (torch.randn(N, generator=g) * 0.1).cuda()
res = None
if os.environ.get("PT_BENCH_RES"):
    pad_ = ks // 2
    res = torch.randn(B, (H + 2 * pad_ - ks) // stride + 1, (W + 2 * pad_ - ks) // stride + 1, N, generator=g).to(torch.bfloat16).cuda()
kw = dict(relu=True, res=res, res_mode=1 if res is not None else 0)
outs, times = {}, {v: [] for v in vals}
for v in vals:
    os.environ[var] = v
    outs[v] = eng.op_conv2d(x, wt, b, ks, stride, **kw).clone()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
scratch = torch.empty_like(outs[vals[0]])
for r in range(rounds):
    for v in vals:
        os.environ[var] = v
        ev0.record()
        for _ in range(iters):
            eng.op_conv2d(x, wt, b, ks, stride, out=scratch, **kw)
        ev1.record()
        torch.cuda.synchronize()
        times[v].append(ev0.elapsed_time(ev1) / iters)
pad = ks // 2
Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
fl = 2.0 * B * Ho * Wo * N * Cin * ks * ks
same = all(torch.equal(outs[vals[0]].view(torch.int16), outs[v].view(torch.int16)) for v in vals[1:])
line = f"conv {ks}x{ks} s{stride} {Cin}->{N} @{H}x{W} B={B}{' +res' if res is not None else ''}: "
for v in vals:
    t = sorted(times[v])
    line += f"{var}={v}: median {t[len(t)//2]*1e3:.1f} us ({fl/t[len(t)//2]/1e9:.0f} TF) min {t[0]*1e3:.1f} us ({fl/t[0]/1e9:.0f} TF)  "
print(line + ("outputs bit-identical" if same else "OUTPUTS DIFFER"))
