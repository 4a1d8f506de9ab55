"""tools/dcn_op_bench.py: the three deformable-convolution kernels on ONE layer shape (64 -> 64 @256 x 256, 16 tables) by offset range -- offsets ~ N(0, sigma px):
what a trained Lore checkpoint produces (sub-pixel to a few pixels) against the bench's random-init fields (2 .. 9 px on average, tails to 40 px)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.weights import tile_conv_weight

eng = HipEngine(0)
dev = torch.device("cuda", 0)
B, C, N, H, W = 16, 64, 64, 256, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, W, C, generator=g).abs().to(torch.bfloat16).to(dev)
w = (torch.randn(N, C, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()
wt = torch.from_numpy(tile_conv_weight(w.permute(0, 2, 3, 1).reshape(N, 9 * C, 1, 1).contiguous()).view(np.int16)).to(dev)
b = torch.zeros(N, device=dev)
for sigma in (0.5, 2.0, 4.0, 8.0, 16.0):
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * sigma
    om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
    om = om.to(dev)
    row = []
    for mode, name in ((0, "valu"), (1, "mfma"), (2, "mfma2")):
        eng.set_dcn_mfma(mode)
        for _ in range(2):
            eng.op_dcn(x, om, wt, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng.op_dcn(x, om, wt, b)
        e1.record()
        torch.cuda.synchronize()
        row.append(f"{name} {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us")
    gb = B * H * W * 36 * 128 / 1e9
    print(f"offsets ~ N(0, {sigma:4.1f} px): " + "   ".join(row) + f"   ({gb:.2f} GB gathered)")
eng.set_dcn_mfma(0)
