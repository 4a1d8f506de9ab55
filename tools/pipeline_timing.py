"""tools/pipeline_timing.py: wall time of OcrTablePipeline.predict (synchronous, stage after stage, one 32-page batch) with
the recogniser on the main stream and on a second stream, and of OcrTablePipeline.predict_stream over 64-page batches
resident on the device (the product API next to bench.py's own loop; GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pdf_table_amd.pipeline import OcrTablePipeline
from pdf_table_amd.synth_pages import make_page

made = [make_page(i) for i in range(8)]
pages = [made[i % 8][0] for i in range(32)]
tb = [np.asarray(made[i % 8][1]["tables"]).reshape(-1, 4) for i in range(32)]
for overlap in (False, True, False, True):
    p = OcrTablePipeline(device=0, synthetic_seed=0, layout=True, table_structure=True, overlap_rec=overlap)
    # random-init detection finds almost no text: give the recogniser the generator's lines through a stub of the box stage
    quads = []
    for i in range(32):
        l = made[i % 8][1]["lines"].astype(np.float64)
        quads.append(np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1))
    stage = p.text_detector._stage
    orig = stage.boxes
    stage.boxes = lambda prob, bm, shape, ev, _o=orig, _q=quads: (_o(prob, bm, shape, ev), _q)[1][:prob.shape[0]]      # pages repeat with period 8: a chunk's quads are the first n
    p.predict(pages, table_boxes=tb)
    ts = []
    for _ in range(3):
        t0 = time.time()
        p.predict(pages, table_boxes=tb)
        ts.append(time.time() - t0)
    print(f"overlap_rec={overlap}: {min(ts) * 1e3:.0f} ms per 32-page predict() (best of 3), {32 / min(ts):.0f} pages/s")
    p.engine.close()

# predict_stream: ten 64-page batches already on the device, table regions given (bench.py's workload shape)
# detect_model="db_pp": the PP-OCR pre/post flavour around DB-ResNet18 (1024^2 page -> 960^2 net input), bench.py's detection
# workload; the "db" flavour of the runs above feeds the net 1024^2
import sys
p = OcrTablePipeline(device=0, synthetic_seed=0, layout=True, table_structure=True, overlap_rec="--overlap-rec" in sys.argv)
print("predict_stream with overlap_rec =", p.overlap_rec)
quads64 = (quads + quads)
stage = p.text_detector._stage
from pdf_table_amd.det_stage import DetConfig
stage.cfg = DetConfig(flavour="db_pp", thresh=stage.cfg.thresh, box_thresh=stage.cfg.box_thresh, unclip_ratio=stage.cfg.unclip_ratio,
                      use_dilation=stage.cfg.use_dilation)
orig = stage.boxes
stage.boxes = lambda prob, bm, shape, ev, _o=orig, _q=quads64: (_o(prob, bm, shape, ev), _q)[1][:prob.shape[0]]
batch = torch.from_numpy(np.stack(pages + pages)).cuda()
tb64 = tb + tb
for _ in p.predict_stream([batch] * 3, table_boxes=[tb64] * 3):
    pass
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.time()
    stamps = []
    for r in p.predict_stream([batch] * 20, table_boxes=[tb64] * 20):
        stamps.append(time.time())
    torch.cuda.synchronize()
    dt = time.time() - t0
    steady = (len(stamps) - 1) * 64 / (stamps[-1] - stamps[0])      # between the first and the last result: no fill / drain
    print(f"predict_stream: 1280 pages in {dt * 1e3:.0f} ms = {1280 / dt:.0f} pages/s incl. the two-batch fill, {steady:.0f} pages/s "
          "between results (64-page batches); host s: " + ", ".join(f"{k} {v:.3f}" for k, v in p.metric["host_seconds"].items()))
for ch in ("0", "32", "16", "0", "32"):      # PT_PREDICT_CHUNK: 0 = the serial path (stage after stage over the whole batch)
    os.environ["PT_PREDICT_CHUNK"] = ch
    p.predict(pages + pages, table_boxes=tb64)
    t0 = time.time()
    for _ in range(4):
        p.predict(batch_pages := [pg for pg in (pages + pages)], table_boxes=tb64)
    dt = time.time() - t0
    print(f"predict (same pipeline, 64 host pages per call, PT_PREDICT_CHUNK={ch}): {256 / dt:.0f} pages/s")
p.engine.close()
