"""CPU experiment: how much the fp32 ORACLE of the synthetic Lore DLA-34+DCN net moves under 1e-7 / 1e-5 relative input noise,
for the default DCN offset gain (0.1) and the well-conditioned one (0.02) the full-size parity test uses.  python tools/lore_conditioning.py [size]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lore_net, lore_pre
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import lore_dla34_state_dict
torch.set_num_threads(8)
img, meta = make_page(0)
x1,y1,x2,y2 = (int(v) for v in meta["tables"].reshape(-1,4)[0])
crop = np.ascontiguousarray(img[y1:y2, x1:x2][:, :, ::-1])
S = int(sys.argv[1]) if len(sys.argv)>1 else 512
x,_ = lore_pre.lore_preprocess(crop, S, S)
for gain in (0.1, 0.02):
    sd = lore_dla34_state_dict(seed=2) if gain==0.1 else lore_dla34_state_dict(seed=2, dcn_gain=gain)
    with torch.no_grad():
        ref = lore_net.dlaseg_forward(sd, x)
        g = torch.Generator().manual_seed(1)
        for eps in (1e-7, 1e-5):
            xp = x * (1 + eps * torch.randn(x.shape, generator=g))
            out = lore_net.dlaseg_forward(sd, xp)
            print(S, "gain", gain, "eps", eps, {k: f"{float((out[k]-ref[k]).abs().max()/max(1.0, ref[k].abs().max())):.2e}" for k in ref})
