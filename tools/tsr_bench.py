#!/usr/bin/env python
"""Timing of the Lore stage pieces on synthetic 1024x1024 pages: python tools/tsr_bench.py [n_pages] [iters]
(PT_PROF_VERBOSE=1 prints per-kernel-label HIP-event times)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdf_table_amd import lib as L
from pdf_table_amd.engine import HipEngine
from pdf_table_amd.synth_pages import make_page
from pdf_table_amd.synth_weights import lore_dla34_state_dict, lore_processor_state_dict
from pdf_table_amd.weights import pack_lore_dla34, pack_lore_processor
from pdf_table_amd.tsr_stage import LoreConfig, TsrStage

npg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = HipEngine(0)
hb = tuple(float(v) for v in os.environ.get('HM_BIAS', '-6.0,-5.0').split(','))
eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lore_dla34_state_dict(21, hm_bias=hb), x3=False))
eng.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(lore_processor_state_dict(31), x3=False))
pages, boxes = [], []
for i in range(npg):
    pg, meta = make_page(i, 1024)
    pages.append(pg)
    boxes.append(np.array(meta["tables"]).reshape(-1, 4))
pages = torch.from_numpy(np.stack(pages)).cuda()
st = TsrStage(eng, LoreConfig(), micro_batch=int(os.environ.get("TSR_MB", "8")))
ntab = sum(len(b) for b in boxes)
res = st(pages, boxes)
torch.cuda.synchronize()
print("tables", ntab, "cells per table", [len(r["polygons"]) for p in res for r in p][:16])
eng.profile_enable(True)
t0 = time.time()
for _ in range(iters):
    res = st(pages, boxes)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
pr = eng.profile_read()
print(f"{dt*1e3:.1f} ms per {npg} pages / {ntab} tables = {dt*1e3/ntab:.2f} ms per table; kernel classes ms/iter:",
      {k: round(v["ms"] / iters, 2) for k, v in pr.items()},
      "TFLOP/s:", {k: round(v["flop"] / v["ms"] / 1e9) for k, v in pr.items() if v["ms"] > 0 and v["flop"] > 0})
