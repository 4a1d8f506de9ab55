"""GPU diagnostic: Lore DLA-34+DCN head maps, BF16X3 vs the fp32 oracle, by input size and content (randn / page crop)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lore_net, lore_pre
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import lore_dla34_state_dict
from pdf_table_amd.weights import pack_lore_dla34

def x4(x, split):
    n, _, H, W = x.shape
    nhwc = x.permute(0, 2, 3, 1)
    if not split:
        o = torch.zeros(n, H, W, 4); o[..., :3] = nhwc; return o.to(torch.bfloat16)
    hi = nhwc.to(torch.bfloat16).float(); lo = (nhwc - hi).to(torch.bfloat16).float()
    o = torch.zeros(n, H, W, 8); o[..., :3] = hi; o[..., 4:7] = lo
    return o.to(torch.bfloat16)

torch.set_num_threads(32)
eng = HipEngine(0)
sd = lore_dla34_state_dict(seed=2)
eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(sd))
img, meta = make_page(0)
x1, y1, x2, y2 = (int(v) for v in meta["tables"].reshape(-1, 4)[0])
crop = np.ascontiguousarray(img[y1:y2, x1:x2][:, :, ::-1])
cases = []
for S in (256, 512, 1024):
    g = torch.Generator().manual_seed(S)
    cases.append((f"randn {S}", torch.randn(1, 3, S, S, generator=g)))
    cases.append((f"crop->{S}", lore_pre.lore_preprocess(crop, S, S)[0]))
for name, x in cases:
    t = time.time()
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(sd, x)
    for mode in ("bf16x3", "bf16"):
        eng.set_precision(L.PT_PRECISION_BF16X3 if mode == "bf16x3" else L.PT_PRECISION_BF16)
        got = eng.tsr_forward_net(x4(x, mode == "bf16x3").cuda())
        torch.cuda.synchronize()
        rels = {k: float((got[k].cpu().permute(0, 3, 1, 2) - ref[k]).abs().max() / max(1.0, ref[k].abs().max())) for k in ref}
        print(name, mode, {k: f"{v:.2e}" for k, v in rels.items()}, f"oracle {time.time()-t:.1f}s", flush=True)
