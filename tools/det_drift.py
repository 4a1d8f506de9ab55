"""bf16 / BF16X3 drift of the DB detector on the two noise pages of test_det_pipeline_boxes, for the four combinations of
PT_CONV_WS64 and PT_DB_FUSE_BIN0 (profiles/r03/experiments.txt):  python tools/det_drift.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import db_net, db_pre
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_weights import db_resnet18_state_dict
from pdf_table_amd.weights import pack_db_resnet18
eng = HipEngine(0)
sd = db_resnet18_state_dict(seed=11)
eng.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd))
rng = np.random.default_rng(21)
pages = rng.integers(0, 256, (2, 160, 224, 3), dtype=np.uint8)
refs = []
for b in range(2):
    chw, _ = db_pre.preprocess_db_pp(pages[b])
    with torch.no_grad():
        refs.append(db_net.db_forward_fp32(sd, torch.from_numpy(np.ascontiguousarray(chw))[None])[0, 0].numpy())
for ws in ("0", "2"):
    for fb in ("0", "1"):
        os.environ["PT_CONV_WS64"] = ws; os.environ["PT_DB_FUSE_BIN0"] = fb
        for mode, name in ((L.PT_PRECISION_BF16, "bf16"), (L.PT_PRECISION_BF16X3, "x3")):
            eng.set_precision(mode)
            prob, bm = eng.det_forward(torch.from_numpy(pages).cuda(), L.PT_DET_PRE_DB_PP, 0.3)
            torch.cuda.synchronize()
            ph = prob.cpu().numpy()
            d = [np.abs(ph[b] - refs[b]) for b in range(2)]
            print(f"ws64={ws} fuse_bin0={fb} {name}: max {max(x.max() for x in d):.5f} mean {np.mean([x.mean() for x in d]):.6f}", flush=True)
