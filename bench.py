#!/usr/bin/env python
"""bench.py -- pages/s of the hot path on N MI355X (one process per GPU, RCCL only for the weight broadcast).

    python bench.py [--gpus 1] [--steps 10] [--warmup 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic 1024x1024 pages per rank (inputs already
resident in HBM).  Workload today (named in ``config.workload``): BASELINE.json configs[1], the batched DB
text-detection stage -- resize + normalise, DB-ResNet18, prob -> bitmap, contour candidates, box scoring,
unclip, final boxes.  Pages are sharded across ranks (weak scaling: fixed pages per rank); there is no
collective in the timed region.

Stages (--stages, default layout,det,rec,tsr): PicoDet layout detection, DB text detection, CRNN recognition of the
page's text lines, Lore table structure of the page's tables.

Extra objects on the JSON line: ``roofline`` for the dominant kernel class (3x3 MFMA convolutions, HIP-event
timed inside the timed region) and ``cpu_baseline`` (the oracle restatement of the same stage on the host
cores, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PAGE = 1024
PAGES_PER_STEP = int(os.environ.get("PT_BENCH_PAGES", "64"))   # per rank
DISTINCT = 8                     # distinct synthetic pages, tiled up to the batch
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
DB_GFLOP_960 = 111.71            # DB-ResNet18 at the reference-preprocessed 960x960


def cpu_baseline_tsr(lsd, psd, page, box):
    """One table through the oracle chain (fp32 torch convs in the reference's op order, DCN restatement, the decode
    with its Python vertex-snapping loop, processor) on the host cores -> seconds per table and a note."""
    from oracle import lore_decode as od
    from oracle import lore_net, lore_pre, lore_processor
    x1, y1, x2, y2 = (int(v) for v in box)
    crop = np.ascontiguousarray(page[y1:y2, x1:x2][:, :, ::-1])
    t0 = time.time()
    x, meta = lore_pre.lore_preprocess(crop, 1024, 1024)
    t1 = time.time()
    with torch.no_grad():
        z = lore_net.dlaseg_forward(lsd, x)
    t2 = time.time()
    with torch.no_grad():
        logi, ps, polys, results = od.process_detect_output(z, meta, wiz_rev=True, vis_thresh=0.2)
    t3 = time.time()
    with torch.no_grad():
        lore_processor.processor_forward(psd, logi, None)
    t4 = time.time()
    return t4 - t0, (f"; table structure {t4 - t0:.2f} s/table measured on 1 table with {logi.shape[1]} cells (warp {t1 - t0:.2f}, "
                     f"DLA-34+DCN fp32 {t2 - t1:.2f}, decode {t3 - t2:.2f}, processor {t4 - t3:.2f})")


def cpu_baseline(sd, pages_np, cfg, csd=None, quads=None, max_lines=12, tsr=None, layout=None, lines_per_page=None):
    """The oracle (port of the reference CPU path: fp32 torch ops in the reference's op order + numpy/python
    pre/post) on the host cores, batch 1 per call as the reference runs it."""
    from oracle import db_nas, db_net, db_post, db_pre
    det_fwd = db_nas.dbnas_forward_fp32 if "backbone.first_conv.0.weight" in sd else db_net.db_forward_fp32
    # 32 threads: on the 256-thread GPU host, batch-1 convolutions and the per-step LSTM matmuls get SLOWER beyond a
    # few dozen threads (47 s for two pages at 256 threads vs ~3 s/page at 8); "cores" reports what was really used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    budget_t0 = time.time()
    n = len(pages_np)
    with torch.no_grad():   # one un-timed warm-up forward (thread pool start-up, oneDNN primitive cache)
        det_fwd(sd, torch.zeros(1, 3, 960, 960))
    t_pre = t_net = t_post = 0.0
    nboxes = 0
    done = 0
    for img in pages_np:
        if done >= 1 and time.time() - budget_t0 > 15.0:      # bounded sample
            break
        done += 1
        t0 = time.time()
        chw, shape_list = db_pre.preprocess_db_pp(img)
        t1 = time.time()
        with torch.no_grad():
            prob = det_fwd(sd, torch.from_numpy(np.ascontiguousarray(chw))[None])[0, 0].numpy()
        t2 = time.time()
        boxes = db_post.db_postprocess(prob, shape_list, img.shape, cfg.thresh, cfg.box_thresh, cfg.unclip_ratio,
                                       cfg.use_dilation, cfg.max_candidates)
        t3 = time.time()
        t_pre += t1 - t0
        t_net += t2 - t1
        t_post += t3 - t2
        nboxes += len(boxes)
    n = done
    dt = t_pre + t_net + t_post
    rec_note = ""
    if csd is not None:
        # recognition: one call per text line like the reference (ocr_system_task.py:309-320); timed on a bounded
        # number of lines and scaled to the page's line count
        from oracle import crnn as ocrnn
        t_rec, nl = 0.0, 0
        # scaled to the WORKLOAD's mean line count per page (the sampled pages may have fewer or more)
        lines_total = lines_per_page * n if lines_per_page else sum(len(q) for q in quads[:n])
        rec_t0 = time.time()
        for img, qs in zip(pages_np[:n], quads[:n]):
            for q in qs[:max_lines]:
                if nl >= 2 and time.time() - rec_t0 > 12.0:     # bounded sample
                    break
                t0 = time.time()
                crop = ocrnn.crop_image(img, ocrnn.order_point(q))
                x = ocrnn.rec_preprocess(crop)
                with torch.no_grad():
                    ocrnn.ctc_greedy_ids(ocrnn.crnn_forward_fp32(csd, x).numpy())
                t_rec += time.time() - t0
                nl += 1
        per_line = t_rec / max(1, nl)
        dt += per_line * lines_total
        rec_note = (f"; recognition {per_line:.3f} s/line measured on {nl} lines (crop + CRNN fp32 with a Python-loop LSTM "
                    f"+ CTC), scaled to {lines_total / n:.0f} lines/page")
    if layout is not None:
        from oracle import picodet as opico
        t0 = time.time()
        for img in pages_np[:n]:
            xl, sf = opico.picodet_preprocess(img)
            with torch.no_grad():
                sc, bx = opico.picodet_forward(layout, torch.from_numpy(xl)[None], 5)
            opico.picodet_postprocess([s_.numpy() for s_ in sc], [b_.numpy() for b_ in bx], list(img.shape[:2]), sf, [800, 608],
                                      opico.LABELS["en"])
        t_lay = (time.time() - t0) / n
        dt += t_lay * n
        rec_note += f"; layout (resize + LCNet/CSP-PAN/PicoHead fp32 + numpy NMS) {t_lay:.2f} s/page"
    if tsr is not None:
        lsd, psd, boxes, tables_per_page = tsr
        per_table, note = cpu_baseline_tsr(lsd, psd, pages_np[0], boxes[0][0])
        dt += per_table * tables_per_page * n
        rec_note += note + f", scaled to {tables_per_page:.2f} tables/page"
    return {"value": n / dt, "unit": "pages/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 1024x1024 pages, batch 1 per call as the reference runs it, "
                      f"torch.set_num_threads({cores}); per page: det pre (numpy) {t_pre / n:.2f} s, DB-ResNet18 fp32 "
                      f"(torch CPU) {t_net / n:.2f} s, det post (pure-Python restatement of cv2/pyclipper) {t_post / n:.2f} s"
                      + rec_note,
            "net_only_pages_per_s": n / t_net}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--det-backbone", default="resnet18", choices=["resnet18", "proxylessnas"],
                    help="DB detector network: DBModel (BASELINE.json's configuration) or DBNasModel (diagnostic variant)")
    ap.add_argument("--no-post", action="store_true", help="device half only (diagnostic; not a valid headline)")
    ap.add_argument("--aux-stream", type=int, default=0,
                    help="1: the small-kernel work without 3x3 convolutions (PicoDet layout, the Lore processor and their "
                         "D2H copies) runs on a second stream beside the conv-heavy nets")
    ap.add_argument("--overlap-rec", action="store_true",
                    help="diagnostic: the recogniser runs on a second stream, concurrently with the other stages "
                         "(per-kernel HIP-event durations then include contention, so the roofline object reads low)")
    ap.add_argument("--stages", default=os.environ.get("PT_BENCH_STAGES", "layout,det,rec,tsr"),
                    help="comma list of stages in the timed step: layout (PicoDet), det (configs[1]), rec, tsr (Lore)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:      # N ranks share the host: keep each rank's CPU-side torch / BLAS work inside its share of the cores
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    dist = None
    # PT_BENCH_FORCE_DIST=1: initialise RCCL and take the broadcast / barrier / all-reduce path even with one rank
    # (exercises the N > 1 code on a single-GPU box; launch with torchrun --nproc-per-node 1 or set MASTER_PORT)
    use_dist = world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pdf_table_amd import lib as L
    from pdf_table_amd.det_stage import DetConfig, DetStage
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.synth_pages import make_page
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    from pdf_table_amd.weights import pack_crnn, pack_db_resnet18
    stages = [x for x in args.stages.split(",") if x]
    assert set(stages) <= {"layout", "det", "rec", "tsr", "cls"} and stages

    eng = HipEngine(local_rank)
    eng_rec, rec_stream = None, None
    aux = torch.cuda.Stream(device=dev) if args.aux_stream else None

    def on_aux():
        return torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()
    # weights: packed once on rank 0, broadcast over RCCL/xGMI, loaded from device memory everywhere
    nas = args.det_backbone == "proxylessnas"
    if nas:
        from pdf_table_amd.synth_weights import db_nas_state_dict as det_state_dict
        from pdf_table_amd.weights import pack_db_nas as pack_det
    else:
        det_state_dict, pack_det = db_resnet18_state_dict, pack_db_resnet18
    det_kind = L.PT_MODEL_DB_NAS if nas else L.PT_MODEL_DB_RESNET18
    sd = det_state_dict(seed=0) if rank == 0 or world == 1 else None
    if use_dist:
        from pdf_table_amd.dist_utils import broadcast_blob
        blob = broadcast_blob(pack_det(sd, x3=False) if rank == 0 else None, dev)
        eng.load_weights_device(det_kind, blob)
    else:
        eng.load_weights(det_kind, pack_det(sd, x3=False))

    rec = None
    if "rec" in stages:
        from pdf_table_amd.rec_stage import RecStage, build_lines
        from pdf_table_amd.synth_weights import crnn_state_dict
        csd = crnn_state_dict(seed=1) if rank == 0 or world == 1 else None
        if use_dist:
            from pdf_table_amd.dist_utils import broadcast_blob
            eng.load_weights_device(L.PT_MODEL_CRNN, broadcast_blob(pack_crnn(csd, x3=False) if rank == 0 else None, dev))
        else:
            eng.load_weights(L.PT_MODEL_CRNN, pack_crnn(csd, x3=False))
        rec = RecStage(eng)
        if args.overlap_rec:       # same engine (every stage has its own activation arena and scratch), second stream
            rec_stream = torch.cuda.Stream(device=dev)

    layout = None
    if "layout" in stages:
        from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
        from pdf_table_amd.synth_weights import picodet_state_dict
        from pdf_table_amd.weights import pack_picodet
        ysd = picodet_state_dict(seed=4, num_classes=5) if rank == 0 or world == 1 else None
        if use_dist:
            from pdf_table_amd.dist_utils import broadcast_blob
            eng.load_weights_device(L.PT_MODEL_PICODET, broadcast_blob(pack_picodet(ysd, 5, x3=False) if rank == 0 else None, dev))
        else:
            eng.load_weights(L.PT_MODEL_PICODET, pack_picodet(ysd, 5, x3=False))
        layout = LayoutStage(eng, PicodetConfig(task_type="en"))

    tsr = None
    if "tsr" in stages:
        from pdf_table_amd.synth_weights import lore_dla34_state_dict, lore_processor_state_dict
        from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
        from pdf_table_amd.weights import pack_lore_dla34, pack_lore_processor
        lsd = lore_dla34_state_dict(seed=2) if rank == 0 or world == 1 else None
        psd = lore_processor_state_dict(seed=3) if rank == 0 or world == 1 else None
        if use_dist:
            from pdf_table_amd.dist_utils import broadcast_blob
            eng.load_weights_device(L.PT_MODEL_LORE_DLA34, broadcast_blob(pack_lore_dla34(lsd, x3=False) if rank == 0 else None, dev))
            eng.load_weights_device(L.PT_MODEL_LORE_PROCESSOR,
                                    broadcast_blob(pack_lore_processor(psd, x3=False) if rank == 0 else None, dev))
        else:
            eng.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(lsd, x3=False))
            eng.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(psd, x3=False))
        tsr = TsrStage(eng, LoreConfig(task_type="wtw"), micro_batch=int(os.environ.get("PT_TSR_MICROBATCH", "80")))

    cls_line = cls_page = None
    if "cls" in stages:      # SURVEY 8f-1 (not part of BASELINE.json's metric; opt-in): PP-LCNet text-line + page orientation
        from pdf_table_amd.cls_stage import ClsStage
        from pdf_table_amd.synth_weights import pplcnet_state_dict
        from pdf_table_amd.weights import pack_pplcnet
        for slot, (seed, ncls) in enumerate(((5, 2), (6, 4))):
            csd_ = pplcnet_state_dict(seed, ncls) if rank == 0 or world == 1 else None
            if use_dist:
                from pdf_table_amd.dist_utils import broadcast_blob
                eng.load_weights_device(L.PT_MODEL_PPLCNET + slot, broadcast_blob(pack_pplcnet(csd_, x3=False) if rank == 0 else None, dev))
            else:
                eng.load_weights(L.PT_MODEL_PPLCNET + slot, pack_pplcnet(csd_, x3=False))
        cls_line, cls_page = ClsStage(eng, "textline_orientation", 0), ClsStage(eng, "text_image_orientation", 1)

    # pages: rank r owns pages [r*P, (r+1)*P) of the global batch (static contiguous shard, dist_utils.shard_range)
    from pdf_table_amd.dist_utils import shard_range
    lo, hi = shard_range(world * PAGES_PER_STEP, rank, world)
    assert hi - lo == PAGES_PER_STEP
    made = [make_page(rank * DISTINCT + i, PAGE) for i in range(DISTINCT)]
    base = [m[0] for m in made]
    pages_np = np.stack([base[i % DISTINCT] for i in range(PAGES_PER_STEP)])
    # Recognition input: the DB weights are random-init, so its boxes are not text.  The recogniser is fed the
    # generator's own text-line rectangles (60-140 per page) as 4-point quads instead.
    gt_quads = []
    for i in range(PAGES_PER_STEP):
        l = made[i % DISTINCT][1]["lines"].astype(np.float64)
        gt_quads.append(np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1))
    lines_per_page = float(np.mean([len(q) for q in gt_quads]))
    # Table regions: the layout net has random-init weights (its boxes are not tables), so the table-structure stage is fed
    # the generator's own table rectangles (1-2 per page) where the reference feeds it the layout boxes with label "table"
    # (ocr_system_task.py:192-198), grown by 8 px like a detector's box
    table_boxes = []
    for i in range(PAGES_PER_STEP):
        t = made[i % DISTINCT][1]["tables"].astype(np.int64).reshape(-1, 4)
        table_boxes.append(np.stack([np.maximum(t[:, 0] - 8, 0), np.maximum(t[:, 1] - 8, 0), np.minimum(t[:, 2] + 8, PAGE),
                                     np.minimum(t[:, 3] + 8, PAGE)], 1))
    tables_per_page = float(np.mean([len(t) for t in table_boxes]))
    pages = torch.from_numpy(pages_np).to(dev)
    cfg = DetConfig(flavour="db_pp", thresh=0.3, box_thresh=0.6, unclip_ratio=1.5)
    stage = DetStage(eng, cfg)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    nboxes = 0
    ntok = 0
    ncells = 0
    ncls_lines = 0
    nlayout = 0

    trace = {} if os.environ.get("PT_BENCH_TRACE") else None

    def tick(name, t0):
        if trace is not None:
            trace[name] = trace.get(name, 0.0) + time.perf_counter() - t0

    def run(steps, count=False):
        """software pipeline: all device work of step k is enqueued before the host halves run (the detection
        post-process of step k-1 first), so the GPU queue never drains while the host works"""
        nonlocal nboxes, ntok, ncells, nlayout, ncls_lines
        prev = None          # detection maps of the previous step (host post-process pending)
        tprev = None         # table-structure state of the previous step (cell counts, processor, host shaping pending)
        tproc = None         # ... of two steps ago (processor queued, rows on their way to pinned memory)
        for k in range(steps):
            t0 = time.perf_counter()
            with on_aux():
                lay = layout.forward(pages) if layout is not None else None  # resize, LCNet/CSP-PAN/PicoHead, candidates (async)
            cur = stage.forward(pages, slot=k & 1) if "det" in stages else None
            rec_ids = None
            if rec is not None:
                with torch.cuda.stream(rec_stream) if rec_stream is not None else contextlib.nullcontext():
                    rec_ids, _ = rec.ids(pages, gt_quads)      # host quad geometry + one pt_rec_forward (async)
            tpend = None
            if tsr is not None:
                tsr_tables, tsr_metas = tsr.tables((PAGE, PAGE), table_boxes)    # host: one affine map per table
                tpend = (tsr.start(pages, tsr_tables), tsr_metas)                # warp, DLA-34+DCN, decode (async)
                if aux is not None:      # the processor of these tables will run on the auxiliary stream, behind this event
                    ev = torch.cuda.Event()
                    ev.record()
                    for (_, _, c_, d_, l_) in tpend[0]:
                        for t_ in (c_, d_, l_):
                            t_.record_stream(aux)
                    tpend = tpend + (ev,)
            cls_out = None
            if cls_line is not None:
                from pdf_table_amd.rec_stage import build_lines
                cls_out = (eng.cls_forward_lines(pages, build_lines(gt_quads), (80, 160), 0, True),
                           eng.cls_forward_pages(pages, (224, 224), 1, False))
            tick("enqueue", t0)
            t0 = time.perf_counter()
            if prev is not None and not args.no_post:          # host half of the previous step, under this step's GPU work
                res = stage.boxes(prev[0], prev[1], (PAGE, PAGE), prev[2])
                if count:
                    nboxes += sum(len(r) for r in res)
            tick("det_post", t0)
            t0 = time.perf_counter()
            if lay is not None:
                lres = layout.finish(lay[0], lay[1], (PAGE, PAGE))      # D2H of the candidates, decode + per-class hard NMS
                if count:
                    nlayout += sum(len(r) for r in lres)
            tick("layout_post", t0)
            t0 = time.perf_counter()
            if rec_ids is not None:
                from pdf_table_amd.rec_stage import ctc_collapse
                with torch.cuda.stream(rec_stream) if rec_stream is not None else contextlib.nullcontext():
                    toks = ctc_collapse(rec_ids.cpu().numpy())     # D2H of int32 [lines, 160] + host collapse
                (eng_rec or eng).check()
                if count:
                    ntok += sum(len(t) for t in toks)
            tick("ctc", t0)
            if cls_out is not None:
                t0 = time.perf_counter()
                lid, lsc = cls_line.top1(cls_out[0])       # D2H of [lines, 2] logits, soft-max + top-1 (vectorised host)
                o = 0
                for q in gt_quads:                         # the reference's per-page upright / upside-down vote
                    cls_line.vote_top1(lid[o:o + len(q)], lsc[o:o + len(q)])
                    o += len(q)
                cls_page.top1(cls_out[1])
                if count:
                    ncls_lines += len(lid)
                tick("cls_post", t0)
            t0 = time.perf_counter()
            if tproc is not None:      # tables of two steps ago: their rows reached pinned memory during the last step
                tres = tsr.collect(tproc[0], tproc[1])
                if count:
                    ncells += sum(len(t["polygons"]) for t in tres)
            tproc = None
            if tprev is not None:      # tables of the previous step: counts are ready, the processor and its D2H are queued
                with on_aux():
                    if aux is not None:
                        aux.wait_event(tprev[2])
                    tproc = (tsr.process(tprev[0]), tprev[1])  # behind this step's work; nothing here blocks on this step
            tick("tsr_finish", t0)
            prev = cur
            tprev = tpend
        if prev is not None and not args.no_post:
            res = stage.boxes(prev[0], prev[1], (PAGE, PAGE), prev[2])
            if count:
                nboxes += sum(len(r) for r in res)
        for fin in ((lambda: tsr.collect(tproc[0], tproc[1])) if tproc is not None else None,
                    (lambda: tsr.finish(tprev[0], tprev[1])) if tprev is not None else None):
            if fin is not None:
                tres = fin()
                if count:
                    ncells += sum(len(t["polygons"]) for t in tres)

    for s_ in (rec_stream, aux):
        if s_ is not None:
            s_.wait_stream(torch.cuda.current_stream(dev))          # the resident pages were uploaded on the default stream
    run(args.warmup)
    barrier()
    # HIP events around the launches of the roofline's kernel class only (mode 2 + class 0 = the 3x3 convs): an event pair per
    # launch of every class (PT_BENCH_PROF=1, fills all_kernel_classes_ms) costs 1-3 % of the step in idle GPU time
    prof_mode = int(os.environ.get("PT_BENCH_PROF", "2"))
    eng.profile_enable(prof_mode)
    if eng_rec is not None:
        eng_rec.profile_enable(prof_mode)
    t0 = time.perf_counter()
    run(args.steps, count=True)
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    if eng_rec is not None:
        for k_, v_ in eng_rec.profile_read().items():
            for f_ in v_:
                prof[k_][f_] += v_[f_]
        eng_rec.profile_enable(False)
    if trace is not None and rank == 0:
        print("[bench trace] host seconds over warm-up + timed steps:", {k: round(v, 3) for k, v in trace.items()}, file=sys.stderr)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_pages = world * PAGES_PER_STEP * args.steps
        value = total_pages / dt
        c3 = prof["conv3x3"]
        achieved = (c3["flop"] / (c3["ms"] * 1e-3)) / 1e12 if c3["ms"] > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_summary.py)
            with open(os.path.join(REPO, "profiles", "pmc_latest.json")) as f:
                pm = json.load(f)
            tot = nl = 0.0
            for kname, rec_ in pm.items():       # every 3x3 conv kernel variant, launch-weighted
                if ("conv_igemm_kernel<3," in kname or "conv3x3_dma" in kname) and "hbm_bytes_per_launch" in rec_:
                    n_ = rec_["FETCH_SIZE"]["dispatches"]
                    tot += rec_["hbm_bytes_per_launch"] * n_
                    nl += n_
            traffic = tot / nl if nl else None
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "kernel": "conv_igemm_kernel<3,*> / conv3x3_dma16_kernel (3x3 implicit-GEMM convolutions of all stages)",
                "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_note": "bytes/launch over the 3x3 conv kernels, PMC passes of profiles/pmc_latest.json "
                                                    "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                "launches": c3["launches"], "avg_launch_ms": c3["ms"] / max(1, c3["launches"]),
                "algorithmic_flop_per_launch": c3["flop"] / max(1, c3["launches"]),
                "events": "3x3 class only" if prof_mode == 2 else "every launch"}
        if prof_mode == 1:       # PT_BENCH_PROF=1: events around every launch
            roof["all_kernel_classes_ms"] = {k: v["ms"] for k, v in prof.items()}
            roof["net_tflops_on_111.71_gflop_per_page"] = DB_GFLOP_960e9_per_page(total_pages, prof)
        out = {"metric": "pages/s", "value": value, "unit": "pages/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ("PicoDet layout detection (resize to 800x608, LCNet + CSP-PAN + PicoHead, hard NMS) + "
                                       if "layout" in stages else "")
                                      + ("BASELINE.json configs[1] batched DB text detection (db_pp pre/post around "
                                       + ("DB-ProxylessNAS [--det-backbone variant, not the BASELINE.json network]" if nas
                                          else "DB-ResNet18") + ", 1024x1024 synthetic pages -> 960x960 net input, boxes out)"
                                       if "det" in stages else "")
                                      + (" + CRNN text-line recognition of the page's text lines (crop, resize, CRNN, "
                                         "arg-max, CTC collapse)" if "rec" in stages else "")
                                      + (" + Lore table-structure recognition of the page's tables (wtw: warp to 1024x1024, "
                                         "DLA-34+DCN, heat-map/corner decode with vertex snapping, 2 x 4-layer processor, "
                                         "quads + logical locations)" if "tsr" in stages else "")
                                      + (" + PP-LCNet text-line orientation of every text line and page orientation of every page "
                                         "[opt-in stage, not part of BASELINE.json's metric]" if "cls" in stages else "")
                                      + (" [DEVICE HALF ONLY]" if args.no_post else "")
                                      + (" [recogniser on a second stream: --overlap-rec diagnostic]" if args.overlap_rec else "")
                                      + (" [layout and the Lore processor on an auxiliary stream]" if args.aux_stream else "")
                                      + "; weights are random-init, so the stages are chained by the page generator's ground truth "
                                        "(table regions for TSR, text-line quads for recognition) instead of each other's outputs",
                          "pages_per_step_per_gpu": PAGES_PER_STEP, "page": [PAGE, PAGE],
                          "stages": stages, "parallelism": f"page-shard x{world}",
                          "text_lines_per_page": lines_per_page if "rec" in stages else 0,
                          "tokens_per_page": ntok / max(1, PAGES_PER_STEP * args.steps),
                          "boxes_per_page": nboxes / max(1, PAGES_PER_STEP * args.steps),
                          "tables_per_page": tables_per_page if "tsr" in stages else 0,
                          "layout_regions_per_page": nlayout / max(1, PAGES_PER_STEP * args.steps),
                          "table_cells_per_page": ncells / max(1, PAGES_PER_STEP * args.steps),
                          "classified_lines_per_page": ncls_lines / max(1, PAGES_PER_STEP * args.steps),
                          "weights": "seeded random init (reference state_dict layout)"},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, pages_np[:2], cfg, csd if rec is not None else None,
                                               gt_quads[:2] if rec is not None else None,
                                               tsr=(lsd, psd, table_boxes, tables_per_page) if tsr is not None else None,
                                               layout=ysd if layout is not None else None,
                                               lines_per_page=lines_per_page if rec is not None else None)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def DB_GFLOP_960e9_per_page(total_pages, prof):
    ms = sum(v["ms"] for k, v in prof.items() if k in ("conv3x3", "conv1x1", "stem"))
    return (total_pages * DB_GFLOP_960 * 1e9 / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0


if __name__ == "__main__":
    main()
