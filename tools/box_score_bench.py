#!/usr/bin/env python
"""Stand-alone time of the DB box-score kernel on the candidates of 64 synthetic pages (tools/box_score_bench.py)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd import engine as E, lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import db_resnet18_state_dict
from pdf_table_amd.weights import pack_db_resnet18

eng = HipEngine(0)
eng.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(db_resnet18_state_dict(seed=0, text_signal=True), x3=False))
pages = torch.from_numpy(np.stack([make_page(i)[0] for i in range(64)])).cuda()
prob, bm = eng.det_forward(pages, L.PT_DET_PRE_DB_PP, 0.3)
torch.cuda.synchronize()
cand, counts = E.db_candidates_batch(bm.cpu().numpy(), 1000, 3.0, 32)
valid = np.arange(cand.shape[1])[None, :] < counts[:, None]
ab = np.zeros((int(counts.sum()), 9), np.float32)
ab[:, 0] = np.repeat(np.arange(64, dtype=np.float32), counts)
ab[:, 1:] = cand[valid]
q = ab[:, 1:].reshape(-1, 4, 2)
area = (q[..., 0].max(1) - q[..., 0].min(1) + 1) * (q[..., 1].max(1) - q[..., 1].min(1) + 1)
print(f"{len(ab)} candidates on 64 pages; bounding rectangles: median {np.median(area):.0f} px, p90 {np.percentile(area, 90):.0f}, p99 {np.percentile(area, 99):.0f}, "
      f"max {area.max():.0f}, sum {area.sum() / 1e6:.1f} Mpx")
def timed(sel, label):
    d = torch.from_numpy(np.ascontiguousarray(ab[sel])).cuda()
    eng.det_box_scores(prob, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.det_box_scores(prob, d)
    e1.record()
    torch.cuda.synchronize()
    print(f"box_score_kernel alone, {label} ({int(np.count_nonzero(sel))} boxes, {area[sel].sum() / 1e6:.1f} Mpx): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")


timed(np.ones(len(ab), bool), "all candidates")
timed(area <= 4096, "rectangles <= 4 096 px")
timed((area > 4096) & (area <= 65536), "4 096 < px <= 65 536")
timed(area > 65536, "> 65 536 px")
order = np.argsort(-area)
timed(np.isin(np.arange(len(ab)), order[:1]), "the largest one")
