#!/usr/bin/env python
"""bench.py -- pages/s of the hot path on N MI355X (one process per GPU, RCCL only for the weight broadcast).

    python bench.py [--gpus N] [--steps 10] [--warmup 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment re-executes itself under ``torch.distributed.run`` with N
ranks (and fails loudly when fewer than N devices are visible); under a launcher it checks WORLD_SIZE == N.

A "step" is one pass of the hot path -- BASELINE.json configs[2]: PicoDet layout + DB text detection + CRNN recognition
+ Lore table structure -- over one batch of synthetic 1024x1024 pages per rank (inputs already resident in HBM).
Pages are sharded across ranks (weak scaling: fixed pages per rank); there is no collective in the timed region.

Chaining.  The detector's checkpoint is random-init except for one hand-built channel that detects the synthetic pages'
text (synth_weights._db_text_signal), so detection yields ~75 boxes per page and THOSE boxes are what the recogniser
crops and reads -- software-pipelined: the recogniser of step k works on the boxes whose host post-process finished
during step k-1 (the bench's pages are the same every step).  Layout -> table structure is chained the same way: the
layout checkpoint is the seeded random PicoDet with the stride-64 branch of its head fitted to the generator's pages
(tools/fit_layout_head.py, picodet_state_dict(table_head=True): a workload device that memorises pages 0..511, not a
detector), and the table-structure stage crops the regions the layout stage labels "table" with score >= 0.2
(get_layout_by_type, ocr_system_task.py:184-198).  --gt-tables feeds it the generator's rectangles instead (rounds 1-2).

Output.  stdout carries ONE compact JSON line (< 6 KB, `compact_line()`: the contract keys, `roofline`, `cpu_baseline`, `summary` -- numbers only);
the full record (every leg with its notes and count tables) is written to bench_detail.json next to this file (--detail-out) and to stderr.
The default run (`--legs core`) is the headline, the roofline legs, the precision legs `summary` is built from and the CPU baseline; `--legs all`
adds the diagnostic / configs[4] legs (overlap_rec, ConvNextViT, ONNX recogniser, MtlTabNet / TableMaster).

Objects of the full record:
  roofline        the dominant kernel class (3x3 MFMA convolutions of all stages), HIP-event timed inside the timed region,
                  FLOP counted from real channel counts and the rows row-limited launches really computed;
                  ``det_backbone``: BASELINE.md section 5's figure, 111.71 GFLOP x det-only pages/s / peak, from a det-only
                  leg of the same run
  tolerance_mode  the same step in PT_PRECISION_BF16X3 -- the arithmetic whose GPU tests assert the north-star tolerance
                  (1e-3 on logits, ids exact outside the oracle's own ties) -- timed over a few steps
  cpu_baseline    the oracle restatement of the same stages on the host cores (bounded sample, rank 0 at N = 1 only),
                  plus BASELINE.json configs[0] (one 640x640 page, det + rec) on CPU and on the GPU
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PAGE = 1024
PAGES_PER_STEP = int(os.environ.get("PT_BENCH_PAGES", "64"))   # per rank
DISTINCT = int(os.environ.get("PT_BENCH_DISTINCT", str(PAGES_PER_STEP)))   # distinct synthetic pages per rank
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_BYTES = 8.0e12          # HBM3E, bytes/s (same guide; ~6.3e12 achievable)
DB_GFLOP_960 = 111.71            # DB-ResNet18 at the reference-preprocessed 960x960


# ---------------------------------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8,
                    help="untimed steps; the default covers the software pipeline's depth (3 batches) and the clock ramp of a GPU that idled "
                         "while the process imported and packed weights (a first process on a fresh box read 554 pages/s at 3, 605 afterwards)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--legs", default="core", choices=["core", "all"],
                    help="core (default): the headline, the roofline legs (det-only, by-class), the precision legs the summary is built from (BF16X3, "
                         "f16, end-to-end agreement), host_pages / one_eighth_host and the CPU baseline.  all: also the diagnostic and configs[4] legs "
                         "(overlap_rec, ConvNextViT, ONNX recogniser, MtlTabNet / TableMaster).  Either way stdout carries ONE compact JSON line "
                         "(< 6 KB); the full record goes to bench_detail.json next to this file and to stderr")
    ap.add_argument("--detail-out", default=os.environ.get("PT_BENCH_DETAIL", os.path.join(REPO, "bench_detail.json")),
                    help="where the full record (every leg, notes, count tables) is written")
    ap.add_argument("--by-class-only", action="store_true", help="diagnostic: the timed region, then only the roofline.by_class leg")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the det-only and BF16X3 legs (diagnostic runs)")
    ap.add_argument("--det-backbone", default="resnet18", choices=["resnet18", "proxylessnas"],
                    help="DB detector network: DBModel (BASELINE.json's configuration) or DBNasModel (diagnostic variant)")
    ap.add_argument("--no-post", action="store_true", help="device half only (diagnostic; not a valid headline)")
    ap.add_argument("--host-pages-leg", action="store_true",
                    help="also run the host_pages leg (pinned host batches, H2D inside the timed region) when --no-extra-legs is given; "
                         "the one_eighth_host child uses it")
    ap.add_argument("--host-share", type=int, default=0,
                    help="(leg of the default run) re-run the timed region in a child process pinned with sched_setaffinity to 1/N of "
                         "the host's cores: what every rank of an N-GPU node gets")
    ap.add_argument("--private-loop", action="store_true",
                    help="time bench.py's own software-pipelined loop (HipRunner.run_private) instead of OcrTablePipeline.predict_stream() "
                         "(diagnostic A/B: the product API is what `value` is measured on)")
    ap.add_argument("--aux-stream", type=int, default=0,
                    help="1: the small-kernel work without 3x3 convolutions (PicoDet layout, the Lore processor and their "
                         "D2H copies) runs on a second stream beside the conv-heavy nets")
    ap.add_argument("--overlap-rec", action="store_true",
                    help="diagnostic: the recogniser runs on a second stream, concurrently with the other stages "
                         "(per-kernel HIP-event durations then include contention, so the roofline object reads low)")
    ap.add_argument("--gt-chain", action="store_true",
                    help="diagnostic: feed the recogniser the generator's text-line rectangles (round-1 behaviour)")
    ap.add_argument("--gt-tables", action="store_true",
                    help="diagnostic: feed the table-structure stage the generator's table rectangles (grown by 8 px) instead of the "
                         "layout stage's own 'table' regions (rounds 1-2 behaviour)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "f16x2"],
                    help="arithmetic mode of the TIMED region (bf16x3: the tolerance mode as the headline; the default line "
                         "reports it as tolerance_mode)")
    ap.add_argument("--stages", default=os.environ.get("PT_BENCH_STAGES", "layout,det,rec,tsr"),
                    help="comma list of stages in the timed step: layout (PicoDet), det (configs[1]), rec, tsr (Lore)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N without a launcher: become N ranks under torch.distributed.run (one process per GPU)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if os.environ.get("PT_BENCH_STUB") not in ("1", "host"):
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible -- refusing to report a "
                             f"{args.gpus}-GPU number from fewer devices")
    port = os.environ.get("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    os.execvp(sys.executable, cmd)


def cpu_info():
    """CPU model / sockets / cores / threads of the host (BASELINE.md section 4.3 asks for them beside the baseline)."""
    model, phys, cores = "?", set(), set()
    try:
        cur = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "?":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    cur = ln.split(":", 1)[1].strip()
                    phys.add(cur)
                elif ln.startswith("core id"):
                    cores.add((cur, ln.split(":", 1)[1].strip()))
    except OSError:
        pass
    return {"model": model, "sockets": max(1, len(phys)), "cores": len(cores) or None, "threads": os.cpu_count()}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the ONLY part of this file that touches oracle/)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_tsr(lsd, psd, page, box):
    """One table through the oracle chain (fp32 torch convs in the reference's op order, DCN restatement, the decode
    with its Python vertex-snapping loop, processor) on the host cores -> seconds per table and a note."""
    import torch
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
                     f"DLA-34+DCN fp32 {t2 - t1:.2f}, decode {t3 - t2:.2f}, processor {t4 - t3:.2f})"), np.asarray(polys, np.float64).reshape(-1, 8)


def cpu_baseline(sd, pages_np, cfg, csd=None, tsr=None, layout=None, gpu=None):
    """The oracle (port of the reference CPU path: fp32 torch ops in the reference's op order -- torch's own LSTM operator
    for the CRNN, as the reference's nn.LSTM -- + numpy/python pre/post) on the host cores, batch 1 per call as the
    reference runs it.  ``gpu``: engine outputs for the sampled page / lines, checked against the oracle here."""
    import torch
    from oracle import crnn as ocrnn
    from oracle import db_nas, db_net, db_post, db_pre
    det_fwd = db_nas.dbnas_forward_fp32 if "backbone.first_conv.0.weight" in sd else db_net.db_forward_fp32
    # 32 threads: on the 256-thread GPU host, batch-1 convolutions get SLOWER beyond a few dozen threads (47 s for two
    # pages at 256 threads vs ~3 s/page at 8); "cores" reports what was really used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    budget_t0 = time.time()
    with torch.no_grad():   # one un-timed warm-up forward (thread pool start-up, oneDNN primitive cache)
        det_fwd(sd, torch.zeros(1, 3, 960, 960))
    t_pre = t_net = t_post = 0.0
    done = 0
    boxes_all, parity = [], {}
    for img in pages_np:
        if done >= 1 and time.time() - budget_t0 > 12.0:      # bounded sample
            break
        t0 = time.time()
        chw, shape_list = db_pre.preprocess_db_pp(img)
        t1 = time.time()
        with torch.no_grad():
            prob = det_fwd(sd, torch.from_numpy(np.ascontiguousarray(chw))[None])[0, 0].numpy()
        t2 = time.time()
        boxes = db_post.db_postprocess(prob, shape_list, img.shape, cfg.thresh, cfg.box_thresh, cfg.unclip_ratio,
                                       cfg.use_dilation, cfg.max_candidates)
        t3 = time.time()
        if done == 0 and gpu is not None:
            for mode in ("bf16", "bf16x3"):
                if gpu.get("det_prob_" + mode) is not None:
                    parity["det_max_abs_dprob_" + mode] = float(np.abs(gpu["det_prob_" + mode] - prob).max())
            parity["det_boxes_oracle_vs_gpu_bf16"] = [int(len(boxes)), int(gpu.get("det_nboxes", -1))]
        t_pre += t1 - t0
        t_net += t2 - t1
        t_post += t3 - t2
        boxes_all.append(boxes)
        done += 1
    n = done
    dt = t_pre + t_net + t_post
    lines_per_page = float(np.mean([len(b) for b in boxes_all]))
    note = ""
    if csd is not None:
        # recognition: one call per text line like the reference (ocr_system_task.py:309-320); timed on a bounded
        # number of the oracle's own detected lines and scaled to the sampled pages' line count
        t_rec, nl = 0.0, 0
        rec_t0 = time.time()
        ids_o = []
        with torch.no_grad():
            ocrnn.crnn_forward_fp32(csd, torch.zeros(1, 3, 32, 640), native_lstm=True)      # warm-up
        # the lines: the GPU's own boxes of page 0 when given (both sides then read identical quads and the token ids can be
        # compared line by line), else the oracle's
        rec_quads = gpu["rec_quads"] if gpu is not None and gpu.get("rec_quads") is not None else boxes_all[0]
        for q in rec_quads:
            if nl >= 2 and time.time() - rec_t0 > 8.0:     # bounded sample
                break
            t0 = time.time()
            crop = ocrnn.crop_image(pages_np[0], ocrnn.order_point(q))
            x = ocrnn.rec_preprocess(crop)
            with torch.no_grad():
                lg = ocrnn.crnn_forward_fp32(csd, x, native_lstm=True)[0]
            ids_o.append(lg.argmax(-1).numpy())
            ocrnn.ctc_greedy_ids(ids_o[-1][None])
            t_rec += time.time() - t0
            nl += 1
        per_line = t_rec / max(1, nl)
        dt += per_line * lines_per_page * n
        note += (f"; recognition {per_line:.3f} s/line measured on {nl} lines (crop + CRNN fp32 with torch's LSTM operator "
                 f"+ CTC), scaled to {lines_per_page:.0f} lines/page")
        if gpu is not None and nl:
            for mode in ("bf16", "bf16x3"):
                g = gpu.get("rec_ids_" + mode)
                if g is not None and len(g) >= nl:
                    parity["rec_token_ids_differing_" + mode] = [int(sum(int((g[i] != ids_o[i]).sum()) for i in range(nl))), int(nl * 160)]
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
        note += f"; layout (resize + LCNet/CSP-PAN/PicoHead fp32 + numpy NMS) {t_lay:.2f} s/page"
    if tsr is not None:
        lsd, psd, boxes, tables_per_page = tsr
        per_table, tnote, polys_o = cpu_baseline_tsr(lsd, psd, pages_np[0], boxes[0][0])
        dt += per_table * tables_per_page * n
        note += tnote + f", scaled to {tables_per_page:.2f} tables/page"
        if gpu is not None:      # the same table (same crop) through the engine's table stage: oracle cells found with >= 3 of 4 vertices within 0.1 / 1 px
            tdir = os.path.join(REPO, "tests")
            if tdir not in sys.path:
                sys.path.insert(0, tdir)
            from e2e_agreement import match_cells
            for mode in ("bf16", "bf16x3"):
                gp = gpu.get("tsr_polys_" + mode)
                if gp is not None:
                    parity["tsr_cells_oracle_engine_matched0p1px_matched1px_" + mode] = [int(len(polys_o)), int(len(gp)), len(match_cells(polys_o, gp, 0.1)[0]),
                                                                                         len(match_cells(polys_o, gp, 1.0)[0])]
    # BASELINE.json configs[0]: one 640x640 page, DB-ResNet18 det + CRNN rec, CPU only (OcrDocument.__call__,
    # model/ocr_pdf/modeling_ocr_pdf.py:313): detection, then one recogniser call per detected line
    cfg0 = None
    if csd is not None:
        img = np.ascontiguousarray(pages_np[0][192:832, 192:832])
        t0 = time.time()
        chw, shape_list = db_pre.preprocess_db_pp(img)
        with torch.no_grad():
            prob = det_fwd(sd, torch.from_numpy(np.ascontiguousarray(chw))[None])[0, 0].numpy()
        b0 = db_post.db_postprocess(prob, shape_list, img.shape, cfg.thresh, cfg.box_thresh, cfg.unclip_ratio,
                                    cfg.use_dilation, cfg.max_candidates)
        t1 = time.time()
        for q in b0:
            x = ocrnn.rec_preprocess(ocrnn.crop_image(img, ocrnn.order_point(q)))
            with torch.no_grad():
                ocrnn.ctc_greedy_ids(ocrnn.crnn_forward_fp32(csd, x, native_lstm=True).numpy())
        t2 = time.time()
        cfg0 = {"workload": "BASELINE.json configs[0]: one 640x640 page, DB-ResNet18 det + CRNN rec of every detected line",
                "cpu_s_per_page": t2 - t0, "cpu_det_s": t1 - t0, "cpu_rec_s": t2 - t1, "lines": int(len(b0)),
                "gpu_ms_per_page_bf16": None if gpu is None else gpu.get("config0_ms"),
                "gpu_lines": None if gpu is None else gpu.get("config0_lines")}
    # BASELINE.json configs[4], recogniser half: the ConvNextViT oracle on a few of those lines (batch 1 per line as the reference)
    cvit = None
    if csd is not None and cfg0 is not None and len(b0):
        from oracle import convnext_vit as ocv
        from pdf_table_amd.synth_weights import convnext_vit_state_dict
        vsd = ocv.canonical_state_dict(convnext_vit_state_dict(seed=7))
        k = min(4, len(b0))
        t0 = time.time()
        for q in b0[:k]:
            x = ocv.chunk_preprocess(ocrnn.crop_image(img, ocrnn.order_point(q)))
            with torch.no_grad():
                ocv.convnext_vit_forward_fp32(vsd, x).argmax(-1)
        cvit = {"lines_per_s": k / (time.time() - t0), "sample": f"{k} lines of the configs[0] page, one line (three chunks) per call, fp32 torch CPU"}
    info = cpu_info()
    out = {"value": n / dt, "unit": "pages/s", "cores": cores, "kind": "port", "cpu": info,
           "sample": f"{n} synthetic 1024x1024 page(s), batch 1 per call as the reference runs it, "
                     f"torch.set_num_threads({cores}) on {info['model']} ({info['sockets']} socket(s), {info['cores']} cores, "
                     f"{info['threads']} threads); per page: det pre (numpy) {t_pre / n:.2f} s, DB-ResNet18 fp32 "
                     f"(torch CPU) {t_net / n:.2f} s, det post (pure-Python restatement of cv2/pyclipper) {t_post / n:.2f} s" + note
                     + "; the reference's default ONNX models (PP-OCR, PicoDet) cannot be timed: neither onnxruntime nor the "
                       "weights exist offline",
           "net_only_pages_per_s": n / t_net}
    if cfg0 is not None:
        out["config0"] = cfg0
    if cvit is not None:
        out["convnext_vit"] = cvit
    if parity:
        out["gpu_vs_oracle_on_the_sample"] = parity
    return out


def one_eighth_host_leg(args, value):
    """Multi-GPU readiness that one GPU can measure: eight ranks share the host, so the same timed region runs once
    more in a child process pinned to 1/8 of the cores this process may use, its post-process pool sized for that share."""
    import subprocess
    cpus = sorted(os.sched_getaffinity(0))
    share = cpus[:max(1, len(cpus) // 8)]
    import tempfile
    detail = os.path.join(tempfile.gettempdir(), f"bench_detail_child_{os.getpid()}.json")     # the child's full record (its stdout line is the compact one)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
           "--no-extra-legs", "--host-pages-leg", "--stages", args.stages, "--precision", args.precision, "--detail-out", detail]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, preexec_fn=lambda: os.sched_setaffinity(0, share),
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        if p.returncode != 0 or not os.path.exists(detail):
            return {"error": f"child exited with {p.returncode}: " + p.stderr[-400:]}
        with open(detail) as f:
            d = json.load(f)
        os.unlink(detail)
        hp = d.get("host_pages") or {}
        return {"pages_per_s_at_one_eighth_host": d["value"], "ratio_to_value": d["value"] / value, "cores": len(share), "of_cores": len(cpus),
                "host_pages_pages_per_s_at_one_eighth_host": hp.get("pages_per_s"),
                "host_pages_ratio_to_value": (hp["pages_per_s"] / value) if hp.get("pages_per_s") else None,
                "note": "same timed region (OcrTablePipeline.predict_stream, 64-page batches), child process pinned to the first 1/8 of the "
                        "cores, detection post-process pool capped to that share"}
    except Exception as e:      # noqa: BLE001 -- a diagnostic leg must not take the bench line down
        return {"error": repr(e)[:300]}


# ---------------------------------------------------------------------------------------------------------------------
# runners
# ---------------------------------------------------------------------------------------------------------------------
class StubRunner:
    """PT_BENCH_STUB=1: no GPU, no engine -- a step is a fixed sleep.  Exercises the launch path (self-launch, rendezvous,
    barrier, max-over-ranks timing, the JSON line) on CPU with gloo (tests/test_dist_gloo.py)."""

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world = args, rank, world
        self.stages = [x for x in args.stages.split(",") if x]
        self.host = None
        if os.environ.get("PT_BENCH_STUB") == "host":
            self.init_host()

    def sync(self):
        pass

    def run(self, steps, count=False):
        if self.host is not None:
            return self.run_host(steps)
        time.sleep(0.01 * steps * (1 + self.rank))      # rank-dependent: the reported time must be the slowest rank's
        return {}

    def config(self, counts, steps):
        if self.host is not None:
            return {"workload": "stub (PT_BENCH_STUB=host): no device work; the REAL host halves of a step per rank on its own page shard -- contours, "
                                "mini-boxes, unclip, filter (the library's thread pool, capped at cores / world), reading order, CTC collapse",
                    "pages_per_step_per_gpu": self.host["pages"], "stages": self.stages, "parallelism": f"page-shard x{self.world}",
                    "post_workers": self.host["workers"], "boxes_per_page": counts.get("boxes", 0) / max(1, self.host["pages"] * steps),
                    "first_page_of_rank": self.host["first_page"]}
        return {"workload": "stub (PT_BENCH_STUB=1): no device work", "pages_per_step_per_gpu": PAGES_PER_STEP,
                "stages": self.stages, "parallelism": f"page-shard x{self.world}"}

    def init_host(self):
        """PT_BENCH_STUB=host: what a rank's CPU does per step, with the device outputs replaced by what the generator knows: a bit-packed bitmap
        with the text lines filled in (the detector's output on these pages), box scores of 0.9, random token ids.  PT_BENCH_STUB_PAGES pages per rank."""
        import numpy as np
        from pdf_table_amd.dist_utils import shard_range
        from pdf_table_amd.synth_pages import make_page
        pages = int(os.environ.get("PT_BENCH_STUB_PAGES", "8"))
        lo, hi = shard_range(self.world * pages, self.rank, self.world)
        assert hi - lo == pages
        net = 960
        bm = np.zeros((pages, net, net), dtype=bool)
        for i in range(pages):
            lines = make_page(lo + i, PAGE)[1]["lines"].astype(np.float64) * (net / PAGE)
            for x0, y0, x1, y1 in lines:
                bm[i, int(y0) + 2:int(y1) - 1, int(x0) + 1:int(x1) - 1] = True
        words = np.packbits(bm.reshape(pages, net, net // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(pages, net, net // 32)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
        rng = np.random.default_rng(self.rank)
        self.host = {"pages": pages, "first_page": lo, "words": words, "net": net, "workers": max(1, min(32, ncpu // max(1, self.world))),
                     "ids": rng.integers(0, 40, (pages * 80, 160)).astype(np.int32)}

    def run_host(self, steps):
        import numpy as np
        from pdf_table_amd.det_stage import sort_boxes_reading_order
        from pdf_table_amd.engine import db_candidates_batch, db_finalize_batch
        from pdf_table_amd.rec_stage import ctc_collapse
        h = self.host
        c = {"boxes": 0, "tok": 0}
        for _ in range(steps):
            boxes, counts = db_candidates_batch(h["words"], 1000, 3.0, n_threads=h["workers"])
            scores = np.full(boxes.shape[:2], 0.9, dtype=np.float32)
            out = db_finalize_batch(boxes, scores, counts, (h["net"], h["net"]), (PAGE, PAGE), 0.6, 1.5, 3.0, filter_tag=True, n_threads=h["workers"])
            out = [sort_boxes_reading_order(b) for b in out]
            c["boxes"] += sum(len(b) for b in out)
            c["tok"] += sum(len(t) for t in ctc_collapse(h["ids"]))
        return c


class HipRunner:
    def __init__(self, args, rank, local_rank, world, dist):
        import torch
        from pdf_table_amd import lib as L
        from pdf_table_amd.det_stage import DetConfig, DetStage
        from pdf_table_amd.dist_utils import shard_range
        from pdf_table_amd.engine import HipEngine
        from pdf_table_amd.synth_pages import make_page
        self.torch, self.L = torch, L
        self.args, self.rank, self.world, self.dist = args, rank, world, dist
        torch.cuda.set_device(local_rank)
        self.dev = dev = torch.device("cuda", local_rank)
        self.stages = stages = [x for x in args.stages.split(",") if x]
        assert set(stages) <= {"layout", "det", "rec", "tsr", "cls"} and stages
        # the (hi, lo) weight tiles are only packed / broadcast when a BF16X3 leg will run (extra legs: one rank only)
        self.x3_leg = ((not args.no_extra_legs and world == 1) or args.precision != "bf16") and args.det_backbone == "resnet18"
        x3 = self.x3_leg            # blobs then also carry the (hi, lo) weight tiles of PT_PRECISION_BF16X3
        self.eng = eng = HipEngine(local_rank)
        self.aux = torch.cuda.Stream(device=dev) if args.aux_stream else None
        self.rec_stream = None
        first = rank == 0 or world == 1

        def load(kind, make_blob):
            """packed once on rank 0, broadcast over RCCL/xGMI, loaded from device memory everywhere"""
            if dist is not None:
                from pdf_table_amd.dist_utils import broadcast_blob
                eng.load_weights_device(kind, broadcast_blob(make_blob() if rank == 0 else None, dev))
            else:
                eng.load_weights(kind, make_blob())

        nas = args.det_backbone == "proxylessnas"
        if nas:
            from pdf_table_amd.synth_weights import db_nas_state_dict
            from pdf_table_amd.weights import pack_db_nas as pack_det
            self.sd = db_nas_state_dict(seed=0) if first else None
        else:
            from pdf_table_amd.synth_weights import db_resnet18_state_dict
            from pdf_table_amd.weights import pack_db_resnet18 as pack_det
            # random init + the hand-built text channel: the detector's own boxes feed the recogniser
            self.sd = db_resnet18_state_dict(seed=0, text_signal=not args.gt_chain) if first else None
        self.nas = nas
        load(L.PT_MODEL_DB_NAS if nas else L.PT_MODEL_DB_RESNET18, lambda: pack_det(self.sd, x3=x3))

        self.rec = self.csd = None
        if "rec" in stages:
            from pdf_table_amd.rec_stage import RecStage
            from pdf_table_amd.synth_weights import crnn_state_dict
            from pdf_table_amd.weights import pack_crnn
            # the fitted classifier (tools/fit_crnn_classifier.py: trained-like arg-max margins); the conv stack and BiLSTMs stay seeded random init
            self.csd = crnn_state_dict(seed=1, conditioned=True) if first else None
            load(L.PT_MODEL_CRNN, lambda: pack_crnn(self.csd, x3=x3))
            self.rec = RecStage(eng)
            if args.overlap_rec:       # same engine (every stage has its own activation arena and scratch), second stream
                self.rec_stream = torch.cuda.Stream(device=dev)
                eng.set_lstm_cluster(False)       # the cluster LSTM needs the GPU to itself (pt_engine_set_lstm_cluster)

        self.layout = self.ysd = None
        self.layout_chain = False
        if "layout" in stages:
            from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
            from pdf_table_amd.synth_weights import picodet_state_dict
            from pdf_table_amd.weights import pack_picodet
            # layout -> table structure chained by the layout stage's own "table" regions: needs the fitted head (which knows pages 0..511)
            self.layout_chain = "tsr" in stages and not args.gt_tables and (world * DISTINCT) <= 512 and PAGE == 1024
            self.ysd = picodet_state_dict(seed=4, num_classes=5, table_head=self.layout_chain) if first else None
            load(L.PT_MODEL_PICODET, lambda: pack_picodet(self.ysd, 5, x3=x3))
            self.layout = LayoutStage(eng, PicodetConfig(task_type="en"))

        self.tsr = self.lsd = self.psd = None
        if "tsr" in stages:
            from pdf_table_amd.synth_weights import lore_dla34_state_dict, lore_processor_state_dict
            from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
            from pdf_table_amd.weights import pack_lore_dla34, pack_lore_processor
            # the Lore conditioning of the end-to-end fixture and of the 1e-3 parity assertions (synth_weights.conditioned_state_dicts): what is
            # timed is what is asserted
            self.lsd = lore_dla34_state_dict(seed=2, dcn_gain=0.02, hm_bias=(-2.0, -2.0), hm_gain=0.25) if first else None
            self.psd = lore_processor_state_dict(seed=3) if first else None
            load(L.PT_MODEL_LORE_DLA34, lambda: pack_lore_dla34(self.lsd, x3=x3))
            load(L.PT_MODEL_LORE_PROCESSOR, lambda: pack_lore_processor(self.psd, x3=x3))
            self.tsr = TsrStage(eng, LoreConfig(task_type="wtw"), micro_batch=int(os.environ.get("PT_TSR_MICROBATCH", "128")))

        self.cls_line = self.cls_page = None
        if "cls" in stages:      # SURVEY 8f-1 (not part of BASELINE.json's metric; opt-in): PP-LCNet text-line + page orientation
            from pdf_table_amd.cls_stage import ClsStage
            from pdf_table_amd.synth_weights import pplcnet_state_dict
            from pdf_table_amd.weights import pack_pplcnet
            for slot, (seed, ncls) in enumerate(((5, 2), (6, 4))):
                csd_ = pplcnet_state_dict(seed, ncls) if first else None
                load(L.PT_MODEL_PPLCNET + slot, lambda: pack_pplcnet(csd_, x3=x3))
            self.cls_line, self.cls_page = ClsStage(eng, "textline_orientation", 0), ClsStage(eng, "text_image_orientation", 1)

        # pages: rank r owns pages [r*P, (r+1)*P) of the global batch (static contiguous shard, dist_utils.shard_range)
        lo, hi = shard_range(world * PAGES_PER_STEP, rank, world)
        assert hi - lo == PAGES_PER_STEP
        made = [make_page(rank * DISTINCT + i, PAGE) for i in range(DISTINCT)]
        self.pages_np = np.stack([made[i % DISTINCT][0] for i in range(PAGES_PER_STEP)])
        # ground truth of the generator: text-line rectangles (only with --gt-chain) and table rectangles.  The layout net
        # has random-init weights (its boxes are not tables), so the table-structure stage is fed the generator's own table
        # rectangles (1-2 per page) where the reference feeds it the layout boxes with label "table"
        # (ocr_system_task.py:192-198), grown by 8 px like a detector's box
        self.gt_quads, self.table_boxes = [], []
        for i in range(PAGES_PER_STEP):
            l = made[i % DISTINCT][1]["lines"].astype(np.float64)
            self.gt_quads.append(np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1))
            t = made[i % DISTINCT][1]["tables"].astype(np.int64).reshape(-1, 4)
            self.table_boxes.append(np.stack([np.maximum(t[:, 0] - 8, 0), np.maximum(t[:, 1] - 8, 0), np.minimum(t[:, 2] + 8, PAGE),
                                              np.minimum(t[:, 3] + 8, PAGE)], 1))
        self.gt_lines_per_page = float(np.mean([len(q) for q in self.gt_quads]))
        self.tables_per_page = float(np.mean([len(t) for t in self.table_boxes]))
        self.pages = torch.from_numpy(self.pages_np).to(dev)
        self.cfg = DetConfig(flavour="db_pp", thresh=0.3, box_thresh=0.6, unclip_ratio=1.5)
        # N ranks share the host: the contour / Clipper pool of each rank stays inside its share of the cores
        # (--host-share N: this process was pinned to 1/N of the cores -- the host an 8-rank job leaves each rank -- and sizes its pool so)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
        workers = max(1, min(32, ncpu // max(1, world)))
        self.host_cores, self.post_workers = ncpu, workers
        self.stage = DetStage(eng, self.cfg, workers=workers)
        # the product API under the clock: the timed loop is OcrTablePipeline.predict_stream() over these stages (weights already
        # loaded / broadcast above); bench.py's own loop stays for the single-stage legs and as the --private-loop A/B
        self.pipe = None
        if self.rec is not None and "det" in stages and "cls" not in stages:
            from pdf_table_amd.pipeline import OcrTablePipeline
            self.pipe = OcrTablePipeline.from_engine(eng, self.stage, self.rec, self.layout, self.tsr, overlap_rec=False,
                                                     aux_layout=bool(args.aux_stream), tsr_on_aux=bool(args.aux_stream),
                                                     lookahead=int(os.environ.get("PT_PIPE_LOOKAHEAD", "1")))
        self.rec_boxes = None        # detection boxes of the last post-processed step: what the recogniser reads
        self.trace = {} if os.environ.get("PT_BENCH_TRACE") else None
        for s_ in (self.rec_stream, self.aux):
            if s_ is not None:
                s_.wait_stream(torch.cuda.current_stream(dev))          # the resident pages were uploaded on the default stream

    def sync(self):
        self.torch.cuda.synchronize()

    def _on(self, stream):
        return self.torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _tick(self, name, t0):
        if self.trace is not None:
            self.trace[name] = self.trace.get(name, 0.0) + time.perf_counter() - t0

    def prime(self):
        """pipeline fill (untimed): one synchronous detection so that the first step's recogniser has boxes to read"""
        if "det" in self.stages and "rec" in self.stages and not self.args.gt_chain and self.rec_boxes is None:
            cur = self.stage.forward(self.pages, slot=0)
            self.rec_boxes = self.stage.boxes(cur[0], cur[1], (PAGE, PAGE), cur[2])

    def uses_pipeline(self, stages=None):
        a = self.args
        return (self.pipe is not None and (stages is None or list(stages) == list(self.stages)) and self.rec_stream is None
                and not (a.private_loop or a.gt_chain or a.no_post))

    def run(self, steps, count=False, stages=None, batch=None):
        """`steps` 64-page batches through the four stages (`batch`: the tensor every step feeds; default the device-resident pages).  Default: the product API -- OcrTablePipeline.predict_stream() over a
        stream of device-resident batches (pdf_table_amd/pipeline.py; results arrive two batches behind the input and the
        generator drains inside the timed region).  Single-stage legs and --private-loop: run_private()."""
        if not self.uses_pipeline(stages):
            assert batch is None
            return self.run_private(steps, count, stages)
        import itertools
        c = {"boxes": 0, "tok": 0, "cells": 0, "layout": 0, "cls_lines": 0, "rec_lines": 0}
        # table regions: the layout stage's own (table_boxes=None: OcrTablePipeline._layout_table_boxes) unless --gt-tables / no fitted head
        tb = itertools.repeat(self.table_boxes, steps) if self.tsr is not None and not self.layout_chain else None
        c["tables"] = 0
        for res in self.pipe.predict_stream(itertools.repeat(self.pages if batch is None else batch, steps), table_boxes=tb):
            if count:
                for r in res:
                    c["boxes"] += len(r.det_result)
                    c["rec_lines"] += len(r.ocr_result)
                    c["tok"] += sum(len(o["text"]) for o in r.ocr_result)
                    c["layout"] += len(r.layout_result or ())
                    c["cells"] += sum(len(t["polygons"]) for t in (r.table_structure_result or ()))
                    c["tables"] += len(r.table_structure_result or ())
        if self.trace is not None:
            try:
                hs = self.torch.cuda.host_memory_stats()
                self.trace["pinned_allocs_so_far"] = {k_: hs[k_] for k_ in hs if "num_host_alloc" in k_ or "num_host_free" in k_ or "host_alloc_time.total" in k_ or "allocated_bytes.current" in k_ or "reserved_bytes.current" in k_}
            except Exception:      # noqa: BLE001
                pass
            for k_, v_ in self.pipe.metric["host_seconds"].items():
                self.trace[k_] = self.trace.get(k_, 0.0) + v_
            self.trace["det_boxes_parts"] = {a: round(b, 3) for a, b in getattr(self.stage, "timing", {}).items()}
            for key in ("gpu_ms_inside_phases", "gpu_ms_between_phases"):
                if key in self.pipe.metric:
                    self.trace[key + f"[{steps} steps]"] = {a: round(b, 1) for a, b in self.pipe.metric[key].items()}
        return c

    def run_private(self, steps, count=False, stages=None):
        """software pipeline: all device work of step k is enqueued, THEN the host halves of step k-1 run (detection
        post-process, layout decode, CTC collapse) and the table results of steps k-1 / k-2 are advanced: the GPU queue never
        drains while the host works, and the host never waits for work of the step it has just queued"""
        torch, args, eng = self.torch, self.args, self.eng
        stages = stages or self.stages
        layout = self.layout if "layout" in stages else None
        rec = self.rec if "rec" in stages else None
        tsr = self.tsr if "tsr" in stages else None
        cls_line = self.cls_line if "cls" in stages else None
        c = {"boxes": 0, "tok": 0, "cells": 0, "layout": 0, "cls_lines": 0, "rec_lines": 0}
        prev = None          # detection maps of the previous step (host post-process pending)
        lay_prev = None      # layout candidates of the previous step (D2H + decode + NMS pending)
        rec_prev = None      # token ids of the previous step (D2H + CTC collapse pending)
        cls_prev = None
        tprev = None         # table-structure state of the previous step (cell counts, processor, host shaping pending)
        tproc = None         # ... of two steps ago (processor queued, rows on their way to pinned memory)
        self.prime()

        def host_half(prev, lay_prev, rec_prev, cls_prev):
            """host halves of the PREVIOUS step: nothing here waits for work queued in the current step, so the GPU queue
            always holds at least one step of work while the host decodes"""
            t0 = time.perf_counter()
            if prev is not None and not args.no_post:
                res = self.stage.boxes(prev[0], prev[1], (PAGE, PAGE), prev[2])
                self.rec_boxes = res
                if count:
                    c["boxes"] += sum(len(r) for r in res)
            self._tick("det_post", t0)
            t0 = time.perf_counter()
            if lay_prev is not None:
                lres = layout.finish(lay_prev[0], lay_prev[1], (PAGE, PAGE))      # D2H of the candidates, decode + per-class hard NMS
                if count:
                    c["layout"] += sum(len(r) for r in lres)
            self._tick("layout_post", t0)
            t0 = time.perf_counter()
            if rec_prev is not None:
                from pdf_table_amd.rec_stage import ctc_collapse
                _, nl_, _, ids_h, ids_ev = rec_prev[:5]
                toks = []
                if nl_:
                    ids_ev.synchronize()         # the copy of THAT step's ids: never waits for work queued since
                    toks = ctc_collapse(ids_h.numpy())     # int32 [lines, 160], host collapse
                eng.check()
                if count:
                    c["tok"] += sum(len(t) for t in toks)
            self._tick("ctc", t0)
            if cls_prev is not None:
                t0 = time.perf_counter()
                lid, lsc = cls_line.top1(cls_prev[0])       # D2H of [lines, 2] logits, soft-max + top-1 (vectorised host)
                o = 0
                for q in cls_prev[2]:                       # the reference's per-page upright / upside-down vote
                    cls_line.vote_top1(lid[o:o + len(q)], lsc[o:o + len(q)])
                    o += len(q)
                self.cls_page.top1(cls_prev[1])
                if count:
                    c["cls_lines"] += len(lid)
                self._tick("cls_post", t0)

        for k in range(steps):
            t0 = time.perf_counter()
            with self._on(self.aux):
                lay = layout.forward(self.pages) if layout is not None else None  # resize, LCNet/CSP-PAN/PicoHead, candidates (async)
            cur = self.stage.forward(self.pages, slot=k & 1, early_copy=len(stages) > 1) if "det" in stages else None
            rec_ids = None
            if rec is not None:
                quads = self.gt_quads if (args.gt_chain or self.rec_boxes is None) else self.rec_boxes
                with self._on(self.rec_stream):
                    rec_ids = rec.start(self.pages, quads)      # host quad geometry + one pt_rec_forward + ids -> pinned memory (async)
                if count:
                    c["rec_lines"] += rec_ids[1]
            tpend = None
            if tsr is not None:
                tsr_tables, tsr_metas = tsr.tables((PAGE, PAGE), self.table_boxes)    # host: one affine map per table
                tpend = (tsr.start(self.pages, tsr_tables), tsr_metas)                # warp, DLA-34+DCN, decode (async)
                if self.aux is not None:      # the processor of these tables will run on the auxiliary stream, behind this event
                    ev = torch.cuda.Event()
                    ev.record()
                    for p_ in tpend[0]:
                        for t_ in p_[2:5]:
                            t_.record_stream(self.aux)
                    tpend = tpend + (ev,)
            cls_out = None
            if cls_line is not None:
                from pdf_table_amd.rec_stage import build_lines
                cls_quads = self.gt_quads if (args.gt_chain or self.rec_boxes is None) else self.rec_boxes
                cls_out = (eng.cls_forward_lines(self.pages, build_lines(cls_quads), (80, 160), 0, True),
                           eng.cls_forward_pages(self.pages, (224, 224), 1, False), cls_quads)
            self._tick("enqueue", t0)
            host_half(prev, lay_prev, rec_prev, cls_prev)
            t0 = time.perf_counter()
            if tproc is not None:      # tables of two steps ago: their rows reached pinned memory during the last step
                tres = tsr.collect(tproc[0], tproc[1])
                if count:
                    c["cells"] += sum(len(t["polygons"]) for t in tres)
            tproc = None
            if tprev is not None:      # tables of the previous step: counts are ready, the processor and its D2H are queued
                with self._on(self.aux):
                    if self.aux is not None:
                        self.aux.wait_event(tprev[2])
                    tproc = (tsr.process(tprev[0]), tprev[1])  # behind this step's work; nothing here blocks on this step
            self._tick("tsr_finish", t0)
            prev, lay_prev, rec_prev, cls_prev, tprev = cur, lay, rec_ids, cls_out, tpend
        host_half(prev, lay_prev, rec_prev, cls_prev)       # drain: the last step's host halves
        for fin in ((lambda: tsr.collect(tproc[0], tproc[1])) if tproc is not None else None,
                    (lambda: tsr.finish(tprev[0], tprev[1])) if tprev is not None else None):
            if fin is not None:
                tres = fin()
                if count:
                    c["cells"] += sum(len(t["polygons"]) for t in tres)
        return c

    # ---- extra legs (after the timed region) ------------------------------------------------------------------------
    def timed(self, steps, warm, stages=None):
        self.run(warm, stages=stages)
        self.sync()
        t0 = time.perf_counter()
        c = self.run(steps, count=True, stages=stages)
        self.sync()
        return time.perf_counter() - t0, c

    def host_pages_leg(self, steps=10, warm=3):
        """PCIe under the clock: the SAME step, but every batch starts in pinned HOST memory -- predict_stream() uploads it on a
        copy stream (192 MiB per 64-page batch, queued as soon as the enqueue thread reaches the batch, i.e. while the GPU still computes the
        batches before it) and the results come back as in the headline leg.  `value` keeps its definition (inputs resident in HBM); this is the
        rate when they are not."""
        if not self.uses_pipeline():
            return None
        host = self.torch.from_numpy(self.pages_np).pin_memory()
        self.run(warm, batch=host)
        self.sync()
        t0 = time.perf_counter()
        self.run(steps, count=True, batch=host)
        self.sync()
        dt = time.perf_counter() - t0
        return {"pages_per_s": PAGES_PER_STEP * steps / dt, "steps": steps, "ms_per_step": dt / steps * 1e3,
                "h2d_bytes_per_step": int(host.numel()), "source": "one pinned uint8 [64,1024,1024,3] host batch per step, H2D on a copy stream "
                                                                   "inside the timed region (pipeline.py:predict_stream)"}

    # (label prefix, class) in match order: every launch label of the four stages falls into one class of roofline.by_class
    BY_CLASS = (("conv3x3 c16", "thin DLA levels (7x7 stem + two 3x3, 16 channels)"), ("dla thin", "thin DLA levels (7x7 stem + two 3x3, 16 channels)"), ("conv3x3", "conv3x3 implicit GEMM"), ("conv1x1", "conv1x1 / row GEMM"),
                ("rows gemm", "conv1x1 / row GEMM"), ("classifier", "classifier GEMM + arg-max"), ("cvit", "ConvNextViT"), ("db head", "DB head (2 x convT)"),
                ("stem", "7x7 / 3x3 stems"), ("lcnet stem", "7x7 / 3x3 stems"), ("dcn", "deformable conv (gather + blend + GEMM)"),
                ("dw convT", "depthwise convT up-sampler + add"), ("lstm", "BiLSTM recurrences"), ("layout dwconv", "depthwise convs (layout)"),
                ("lcnet SE", "depthwise convs (layout)"), ("maxpool", "max-pools"), ("preprocess", "pre-processing (det / rec / tsr / layout)"),
                ("rec ", "pre-processing (det / rec / tsr / layout)"), ("tsr pre", "pre-processing (det / rec / tsr / layout)"),
                ("layout pre", "pre-processing (det / rec / tsr / layout)"), ("tsr attention", "Lore processor (attention, norms)"),
                ("tsr norm", "Lore processor (attention, norms)"), ("tsr tok", "Lore processor (attention, norms)"), ("tsr cvt", "Lore processor (attention, norms)"),
                ("tsr", "Lore decode (peaks, top-K, gathers, snapping)"), ("crnn", "CRNN fills / limits / conv0"), ("bitmap", "DB bitmap / box scores"),
                ("box score", "DB bitmap / box scores"), ("argmax", "classifier GEMM + arg-max"))

    def by_class_leg(self, steps=3, warm=1):
        """roofline.by_class: the same step with HIP events around EVERY launch (pt_profile_enable(1); costs 1-3 % of the step in
        idle GPU time, hence its own leg), the launches grouped into kernel classes: ms per step, launches, algorithmic FLOP and bytes per step
        (what the launchers state: conv / GEMM FLOP = 2 x pixels x real N x K, bytes = inputs + outputs + residual + weights once), and the
        fraction of the bf16 MFMA peak and of the 8 TB/s HBM peak that makes -- `bound` is the larger of the two.  Launches whose launcher
        states neither are listed with their time only."""
        if not self.uses_pipeline():
            return None
        eng = self.eng
        self.run(warm)
        self.sync()
        eng.profile_enable(1)
        t0 = time.perf_counter()
        self.run(steps)
        self.sync()
        dt = (time.perf_counter() - t0) / steps
        labels = eng.profile_read_labels()
        eng.profile_enable(False)
        cls = {}
        for lab, r in labels.items():
            name = next((c for p_, c in self.BY_CLASS if lab.startswith(p_)), "other")
            d = cls.setdefault(name, {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0})
            for k_ in d:
                d[k_] += r[k_]
        out = {}
        for name, d in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
            ms = d["ms"] / steps
            row = {"ms_per_step": round(ms, 3), "launches_per_step": round(d["launches"] / steps, 1)}
            if d["flop"] > 0 and ms > 0:
                row["gflop_per_step"] = round(d["flop"] / steps / 1e9, 1)
                row["frac_mfma_peak"] = round(d["flop"] / steps / (ms * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12), 4)
            if d["bytes"] > 0 and ms > 0:
                row["gbytes_per_step"] = round(d["bytes"] / steps / 1e9, 2)
                row["frac_hbm_peak"] = round(d["bytes"] / steps / (ms * 1e-3) / HBM_PEAK_BYTES, 4)
            if "frac_mfma_peak" in row or "frac_hbm_peak" in row:
                row["bound"] = "mfma" if row.get("frac_mfma_peak", 0) >= row.get("frac_hbm_peak", 0) else "hbm"
            out[name] = row
        if os.environ.get("PT_BENCH_LABELS_OUT"):     # every launch label of the step (diagnostic: tools read it to size a kernel class)
            with open(os.environ["PT_BENCH_LABELS_OUT"], "w") as f:
                for lab, r in sorted(labels.items(), key=lambda kv: -kv[1]["ms"]):
                    f.write(f"{r['ms'] / steps:9.3f} ms  {r['launches'] / steps:6.1f} launches  {r['flop'] / steps / 1e9:10.1f} GFLOP  "
                            f"{r['bytes'] / steps / 1e9:8.2f} GB  {lab}\n")
        top = sorted(labels.items(), key=lambda kv: -kv[1]["ms"])[:16]
        top_labels = {lab: {"ms_per_step": round(r["ms"] / steps, 3), "launches_per_step": round(r["launches"] / steps, 1),
                            **({"frac_mfma_peak": round(r["flop"] / (r["ms"] * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12), 4)} if r["flop"] > 0 and r["ms"] > 0 else {}),
                            **({"frac_hbm_peak": round(r["bytes"] / (r["ms"] * 1e-3) / HBM_PEAK_BYTES, 4)} if r["bytes"] > 0 and r["ms"] > 0 else {})}
                      for lab, r in top}
        kern = sum(r["ms_per_step"] for r in out.values())
        return {"ms_per_step_wall": round(dt * 1e3, 2), "ms_per_step_kernels": round(kern, 2), "steps": steps,
                "peaks": {"mfma_tflops": MFMA_PEAK_TFLOPS, "hbm_tb_s": HBM_PEAK_BYTES / 1e12}, "classes": out, "top_labels": top_labels}

    def det_only_leg(self, steps=20, warm=5):
        """BASELINE.json configs[1] in the same run: the det stage alone (pre, DB-ResNet18, bitmap, host post overlapped); steps / warm-up as
        the stand-alone `bench.py --stages det` defaults (a 10-step leg read 2-3 % low: one software-pipeline fill + drain in 0.1 s)"""
        dt, c = self.timed(steps, warm, stages=["det"])
        pps = PAGES_PER_STEP * steps / dt
        # the same leg without the host post-process and its box-score kernel (device half only: pre-process, network, head + bitmap), and the
        # FLOP the engine EXECUTES per page: the 111.71 GFLOP credit is the reference graph's; the exact out2 / binarize.0 refactorings
        #  run fewer -- both stated so that `frac` can be read either way
        net = None
        if not self.args.no_post:
            self.args.no_post = True
            try:
                dtn, _ = self.timed(steps, warm, stages=["det"])
                self.eng.profile_enable(1)
                self.run(2, stages=["det"])
                self.sync()
                lab = self.eng.profile_read_labels()
                self.eng.profile_enable(False)
            finally:
                self.args.no_post = False
            ppsn = PAGES_PER_STEP * steps / dtn
            ex = sum(r["flop"] for r in lab.values()) / (2 * PAGES_PER_STEP) / 1e9
            net = {"pages_per_s": ppsn, "frac": DB_GFLOP_960 * 1e9 * ppsn / (MFMA_PEAK_TFLOPS * 1e12),
                   "executed_gflop_per_page": ex, "frac_of_executed_flop": ex * 1e9 * ppsn / (MFMA_PEAK_TFLOPS * 1e12),
                   "note": "device half only (--no-post): no contour / box-score / unclip work beside the network"}
        return {"pages_per_s_det_only": pps, "steps": steps, "boxes_per_page": c["boxes"] / (PAGES_PER_STEP * steps),
                "gflop_per_page": DB_GFLOP_960, "achieved_tflops": DB_GFLOP_960 * 1e9 * pps / 1e12,
                "frac": DB_GFLOP_960 * 1e9 * pps / (MFMA_PEAK_TFLOPS * 1e12), "net_only": net,
                "definition": "BASELINE.md section 5: 111.71e9 x det-only pages/s / 2.5e15 (960x960 graph, whole det stage "
                              "incl. pre-process, bitmap and the overlapped host post-process in the time)"}

    def overlap_leg(self, steps=8, warm=2):
        """the same step with the recogniser on a second stream beside layout / detection / TSR (what OcrTablePipeline does by
        default).  Not the driver's `value`: kernels of two streams share the CUs, so per-launch durations -- the roofline's
        denominator -- would no longer be those of the kernel alone."""
        torch = self.torch
        if self.rec is None or self.rec_stream is not None:
            return None
        self.rec_stream = torch.cuda.Stream(device=self.dev)
        self.rec_stream.wait_stream(torch.cuda.current_stream(self.dev))
        self.eng.set_lstm_cluster(False)      # the cluster LSTM needs the GPU to itself (pt_engine_set_lstm_cluster)
        try:
            dt, _ = self.timed(steps, warm)
        finally:
            self.sync()
            self.rec_stream = None
            self.eng.set_lstm_cluster(os.environ.get("PT_LSTM_CLUSTER", "1") != "0")
        return {"pages_per_s": PAGES_PER_STEP * steps / dt, "steps": steps,
                "schedule": "recogniser on a second stream (streaming LSTM kernel), everything else as in the timed region"}

    def x3_leg_run(self, steps=8, warm=2):
        """the same step in PT_PRECISION_BF16X3 (the mode whose tests assert 1e-3 / id-exact parity); eight timed steps: the software pipeline's fill
        and drain are one step's worth of a three-step run (the bf16 region reads 3 % lower at 10 steps than at 20 for the same reason); the default
        bench passes the headline's K"""
        L = self.L
        self.eng.set_precision(L.PT_PRECISION_BF16X3)
        try:
            dt, c = self.timed(steps, warm)
        finally:
            self.eng.set_precision(L.PT_PRECISION_BF16)
        n = PAGES_PER_STEP * steps
        return {"precision": "bf16x3 (hi/lo bf16 pairs, 3 MFMA passes, fp32 accumulate)", "pages_per_s": n / dt, "steps": steps,
                "ms_per_step": dt / steps * 1e3, "boxes_per_page": c["boxes"] / n, "tokens_per_page": c["tok"] / n,
                "table_cells_per_page": c["cells"] / n,
                "asserted_by": "tests/test_gpu_fullsize.py::test_fullsize_*_oracle_parity (x3: <= 1e-3 of the logit scale at "
                               "BASELINE sizes, token ids exact outside the oracle's own <= 2e-3 ties); bf16 drift recorded there"}

    def convnext_vit_leg(self, steps=3, warm=1):
        """BASELINE.json configs[4], recogniser half: the SAME lines of the step (the detector's boxes on the 64 pages) through
        the ConvNextViT recogniser (crop, 32 x 804 chunking pre-processor, ConvNext + ViT, arg-max), bf16 and BF16X3"""
        torch, L, eng = self.torch, self.L, self.eng
        if self.rec is None:
            return None
        from pdf_table_amd.rec_stage import build_lines
        from pdf_table_amd.synth_weights import convnext_vit_state_dict
        from pdf_table_amd.weights import pack_convnext_vit
        eng.load_weights(L.PT_MODEL_CONVNEXT_VIT, pack_convnext_vit(convnext_vit_state_dict(seed=7)))
        quads = self.gt_quads if (self.args.gt_chain or self.rec_boxes is None) else self.rec_boxes
        lines = build_lines(quads)
        import numpy as np
        ok = (lines["crop_w"] > 0) & (lines["crop_h"] > 0)
        ratio = lines["crop_w"] / np.maximum(lines["crop_h"], 1).astype(np.float64)
        tw = np.where(ok, np.where(ratio > 804 / 32, 804, (32 * ratio).astype(np.int64)), 0)
        chunks = float(((tw > 0).astype(int) + (tw > 252) + (tw > 504)).mean()) if len(lines) else 0.0
        out = {"lines_per_step": int(len(lines)), "steps": steps, "tokens_per_line": 201,
               "chunks_with_text_per_line": chunks,      # of 3: all-padding chunks are computed once per micro-batch and shared
               "gflop_per_line": 0.59 + 11.7 * chunks / 3, "asserted_by": "tests/test_gpu_cvit.py (x3: <= 1e-3 on the winning logit against the reference "
                                                      "module's own output, ids exact outside <= 2e-3 ties; bf16 drift recorded there)"}
        for name, prec in (("bf16", L.PT_PRECISION_BF16), ("bf16x3", L.PT_PRECISION_BF16X3)):
            eng.set_precision(prec)
            try:
                for _ in range(warm):
                    eng.rec_cvit_forward(self.pages, lines)
                self.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.rec_cvit_forward(self.pages, lines)
                self.sync()
                dt = (time.perf_counter() - t0) / steps
            finally:
                eng.set_precision(L.PT_PRECISION_BF16)
            out[name] = {"lines_per_s": len(lines) / dt, "ms_per_step": dt * 1e3, "pages_per_s_rec_only": PAGES_PER_STEP / dt}
        return out

    def onnx_rec_leg(self, n_lines=256):
        """The reference's DEFAULT recogniser route: OcrRecognitionTask(model="PP-OCRv4") on an ONNX graph through the generic
        layer-list executor (ocr_recognition_task.py:81-116 -> onnxruntime in the reference; pdf_table_amd/onnx_exec.py here).  The real PP-OCRv4
        file is not available offline: the graph is the SVTR-type stand-in PyTorch's exporter writes (tools/onnx_export.py: conv stem, two
        LayerNorm / fused-qkv attention / MLP blocks, CTC head with Softmax; [batch, 3, 48, 320] with a symbolic batch like the shipped exports), seeded.
        Text-line crops of the step's pages -> PPOcrRecPreProcessor kernel -> one graph run per line -> CTCLabelDecode; lines/s in the executor's
        bf16 mode and in its tolerance mode (precision="fp32" -> (hi | lo) activations, <= 1e-3 against the fp32 module: tests/test_gpu_onnx_seq.py)."""
        import tempfile
        tdir = os.path.join(REPO, "tools")
        if tdir not in sys.path:
            sys.path.insert(0, tdir)
        try:
            import onnx_export as X
        except Exception as e:      # noqa: BLE001 -- the exporter needs torch.onnx; a diagnostic leg must not take the bench line down
            return {"error": repr(e)[:200]}
        from pdf_table_amd.ocr_recognition_task import OcrRecognitionTask
        torch = self.torch
        crops = []
        for pi in range(PAGES_PER_STEP):
            for q in self.gt_quads[pi]:
                x0, y0, x1, y1 = int(q[0]), int(q[1]), int(q[4]), int(q[5])
                if x1 - x0 >= 16 and y1 - y0 >= 8:
                    crops.append(np.ascontiguousarray(self.pages_np[pi][y0:y1, x0:x1]))
                if len(crops) >= n_lines:
                    break
            if len(crops) >= n_lines:
                break
        out = {"lines": len(crops), "graph": "SvtrTiny stand-in (tools/onnx_export.py), dynamic batch like the shipped exports (Shape / Gather / Concat glue), static 48 x 320 input: a six-line mini-batch is one walk, replayed from a captured HIP graph",
               "asserted_by": "tests/test_gpu_onnx_seq.py (bf16: similarity bounds; precision='fp32': strings identical to the fp32 module's)"}
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "ppocr_keys_v1.txt"), "w", encoding="utf-8") as f:
                f.write("\n".join(chr(0x4E00 + i) for i in range(95)) + "\n")
            with open(os.path.join(td, "inference.onnx"), "wb") as f:
                f.write(X.torch_export(X.seeded(X.SvtrTiny(classes=97), 31), torch.zeros(2, 3, 48, 320), dynamic_batch=True))
            for name, prec in (("bf16", "bf16"), ("bf16x3", "fp32")):
                task = OcrRecognitionTask(model="PP-OCRv4", task_type="ch", task_path=td, engine=self.eng, precision=prec)
                task(crops[:16])
                self.sync()
                t0 = time.perf_counter()
                texts = task(crops)
                self.sync()
                dt = time.perf_counter() - t0
                out[name] = {"lines_per_s": len(crops) / dt, "ms_per_line": dt / max(1, len(crops)) * 1e3, "characters": sum(len(t) for t in texts),
                             "fused_residual_adds": len(task._exec._fuse), "layers": len(task._exec.layers)}
        return out

    def mtl_tabnet_leg(self, steps=2, warm=2, modes=("bf16", "bf16_kv8", "bf16x3")):
        """BASELINE.json configs[4], table-structure half: the SAME table regions of the step through MtlTabNet (480 x 480 pre-processing
        kernel, ResNet-GC backbone, KV-cached structure / box / cell-content decoders, label convertor + HTML post-processor on the
        host) at the reference's sequence limits (500 structure tokens, 150 cell-content tokens), bf16 and BF16X3"""
        torch, L, eng = self.torch, self.L, self.eng
        if self.tsr is None:
            return None
        from pdf_table_amd.mtl_stage import MtlStage, MtlTabNetConvertor
        from pdf_table_amd.synth_weights import mtl_table_signal_from, mtl_tabnet_backbone_state_dict, mtl_tabnet_decoder_state_dict
        from pdf_table_amd.weights import pack_mtl_backbone, pack_mtl_decoder
        conv = MtlTabNetConvertor()
        eng.load_weights(L.PT_MODEL_MTL_BACKBONE, pack_mtl_backbone(mtl_tabnet_backbone_state_dict(seed=41)))
        # seeded random decoders + the hand-built table channels (synth_weights._mtl_table_signal): every table decodes <tbody>, 20 rows of
        # <tr><td></td><td colspan="2"></td><eb></eb></tr>, </tbody>, <EOS> (163 structure positions) and its 40 content cells 8 characters + <EOS>
        eng.load_weights(L.PT_MODEL_MTL_DECODER, pack_mtl_decoder(mtl_tabnet_decoder_state_dict(
            seed=43, num_classes=conv.num_classes(), num_classes_cell=conv.num_classes_cell(), table_signal=mtl_table_signal_from(conv, rows_until=150)),
            conv.decoder_cfg()))
        n_tab = int(sum(len(t) for t in self.table_boxes))
        out = {"tables_per_step": n_tab, "steps": steps, "max_seq_len": conv.max_seq_len, "max_seq_len_cell": conv.max_seq_len_cell,
               "note": "seeded random decoders with hand-built table channels (synth_weights._mtl_table_signal: reserved residual dimensions carry the "
                       "previous token and the position to the classifier): every table decodes 163 structure positions (20 rows, </tbody>, <EOS>) and "
                       "40 content cells of 9 positions each -- structure loop, box head AND cell-content decoder are under the clock; sequence limits "
                       "are the reference's (500 / 150), the loops stop at <EOS> like a trained model's",
               "asserted_by": "tests/test_gpu_mtl.py (x3: tokens identical, boxes / tag / cell logits <= 1e-3 of scale against the oracle "
                              "pinned to the reference's own MtlTabNetDecoder; task end to end against the composed oracle chain), "
                              "tests/test_mtl_host.py (convertor + post-processor identical to the reference's own classes)"}
        # bf16_kv8: bf16 arithmetic, the structure loop's source-attention keys / values streamed as fp8 (pt_engine_set_mtl_kv_fp8: the fp8 of
        # configs[4] where it pays on this path -- half the bytes of the loop's dominant HBM stream; drift recorded in tests/test_gpu_mtl.py)
        for name, prec in (("bf16", L.PT_PRECISION_BF16), ("bf16_kv8", L.PT_PRECISION_BF16), ("bf16x3", L.PT_PRECISION_BF16X3)):
            if name not in modes:
                continue
            eng.set_precision(prec)
            eng.set_mtl_kv_fp8(name == "bf16_kv8")
            try:
                stage = MtlStage(eng, conv, micro_batch=int(os.environ.get("PT_MTL_MICROBATCH", "128")))
                for _ in range(warm):
                    stage(self.pages, self.table_boxes)
                self.sync()
                stage.stats = {k: 0 for k in stage.stats}
                t0 = time.perf_counter()
                for _ in range(steps):
                    res = stage(self.pages, self.table_boxes)
                self.sync()
                dt = (time.perf_counter() - t0) / steps
                # the same steps through MtlStage.stream: the host half of step k (convertor + post-processor, Python) on a worker thread under step k + 1's
                # device loop -- what a caller with more than one batch gets; `tables_per_s` stays the synchronous call
                if name != "bf16x3":
                    stats_keep = dict(stage.stats)
                    n_str = max(steps, 3)
                    self.sync()
                    t0 = time.perf_counter()
                    for res_s in stage.stream([(self.pages, self.table_boxes)] * n_str):
                        pass
                    self.sync()
                    dt_stream = (time.perf_counter() - t0) / n_str
                    stage.stats = stats_keep
                else:
                    dt_stream = None
            finally:
                eng.set_precision(L.PT_PRECISION_BF16)
                eng.set_mtl_kv_fp8(False)
            st = stage.stats
            out[name] = {"tables_per_s": n_tab / dt, "ms_per_step": dt * 1e3, "pages_per_s_tsr_only": PAGES_PER_STEP / dt,
                         "tables_per_s_streamed": n_tab / dt_stream if dt_stream else None,
                         "structure_tokens_per_table": st["tokens"] / max(1, st["tables"]), "cells_per_table": st["cells"] / max(1, st["tables"]),
                         "cell_steps_per_table": st["cell_steps"] / max(1, st["tables"]),
                         "structure_tokens_per_s": st["tokens"] / steps / dt,
                         "boxes_per_table": float(sum(len(r["polygons"]) for pg in res for r in pg)) / max(1, n_tab)}
        # model="TableMaster" (table_master_config.py: TableMasterDecoder = the same layers without the cell-content decoder): the same tables, the same
        # hand-built structure chain, bf16 -- what the cell-content decoder and its formatting cost is the difference to `bf16` above
        if "bf16" in modes:
            from pdf_table_amd.mtl_stage import TableMasterConvertor
            from pdf_table_amd.synth_weights import table_master_decoder_state_dict
            tconv = TableMasterConvertor()
            eng.load_weights(L.PT_MODEL_MTL_DECODER, pack_mtl_decoder(table_master_decoder_state_dict(
                seed=43, num_classes=tconv.num_classes(), table_signal=mtl_table_signal_from(conv, rows_until=150)), tconv.decoder_cfg()))
            stage = MtlStage(eng, tconv, micro_batch=int(os.environ.get("PT_MTL_MICROBATCH", "128")))
            stage(self.pages, self.table_boxes)
            self.sync()
            stage.stats = {k: 0 for k in stage.stats}
            t0 = time.perf_counter()
            for _ in range(steps):
                res = stage(self.pages, self.table_boxes)
            self.sync()
            dt = (time.perf_counter() - t0) / steps
            st = stage.stats
            out["table_master_bf16"] = {"tables_per_s": n_tab / dt, "ms_per_step": dt * 1e3, "structure_tokens_per_table": st["tokens"] / max(1, st["tables"]),
                                        "cells_per_table": st["cells"] / max(1, st["tables"]),
                                        "boxes_per_table": float(sum(len(r["polygons"]) for pg in res for r in pg)) / max(1, n_tab),
                                        "asserted_by": "tests/test_table_master.py, tests/test_gpu_mtl.py::test_table_master_* (the reference's own TableMasterDecoder / "
                                                       "TableMasterConvertor outputs)"}
        return out

    def second_engine(self, precision):
        """a second engine on the same GPU with the SAME checkpoint set packed for `precision` ("f16": fp16 tiles, PT_PRECISION_F16), its stages and its
        OcrTablePipeline -- what a user gets from OcrTablePipeline(precision="fp16").  Rank 0 only (it holds the state dicts)."""
        from pdf_table_amd.det_stage import DetStage
        from pdf_table_amd.engine import HipEngine
        from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
        from pdf_table_amd.pipeline import OcrTablePipeline
        from pdf_table_amd.rec_stage import RecStage
        from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
        from pdf_table_amd.weights import pack_crnn, pack_db_resnet18, pack_lore_dla34, pack_lore_processor, pack_picodet
        L = self.L
        e = HipEngine(self.dev.index)
        e.set_precision({"f16": L.PT_PRECISION_F16}[precision])
        fmt = e.weight_fmt
        e.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(self.sd, fmt=fmt))
        e.load_weights(L.PT_MODEL_CRNN, pack_crnn(self.csd, fmt=fmt))
        e.load_weights(L.PT_MODEL_PICODET, pack_picodet(self.ysd, 5, fmt=fmt))
        e.load_weights(L.PT_MODEL_LORE_DLA34, pack_lore_dla34(self.lsd, fmt=fmt))
        e.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(self.psd, fmt=fmt))
        stage = DetStage(e, self.cfg, workers=self.post_workers)
        rec, layout = RecStage(e), LayoutStage(e, PicodetConfig(task_type="en"))
        tsr = TsrStage(e, LoreConfig(task_type="wtw"), micro_batch=int(os.environ.get("PT_TSR_MICROBATCH", "128")))
        pipe = OcrTablePipeline.from_engine(e, stage, rec, layout, tsr, overlap_rec=False, aux_layout=bool(self.args.aux_stream),
                                            tsr_on_aux=bool(self.args.aux_stream), lookahead=int(os.environ.get("PT_PIPE_LOOKAHEAD", "1")))
        return dict(eng=e, stage=stage, rec=rec, layout=layout, tsr=tsr, pipe=pipe)

    @contextlib.contextmanager
    def on_engine(self, other):
        """run()/timed() on another engine's stages (second_engine): the timed loop is the same code"""
        keep = {k: getattr(self, k) for k in other}
        for k, v in other.items():
            setattr(self, k, v)
        try:
            yield
        finally:
            for k, v in keep.items():
                setattr(self, k, v)

    def f16_leg_run(self, steps=8, warm=3):
        """the same step in PT_PRECISION_F16 -- single-pass IEEE half, the reference's own GPU arithmetic (base_infer_task.py:56-57 precision="fp16"):
        same kernels (instantiated for the half storage format, csrc/act16.h), same bytes, same MFMA rate, 11 significant bits instead of 8"""
        if not (self.layout_chain and self.pipe is not None and self.tsr is not None and not self.nas):
            return None
        if getattr(self, "_f16", None) is None:
            self._f16 = self.second_engine("f16")
        with self.on_engine(self._f16):
            dt, c = self.timed(steps, warm)
        n = PAGES_PER_STEP * steps
        return {"precision": "f16 (IEEE half activations and weights, one MFMA pass, fp32 accumulate, saturating stores)", "pages_per_s": n / dt,
                "steps": steps, "ms_per_step": dt / steps * 1e3, "boxes_per_page": c["boxes"] / n, "tokens_per_page": c["tok"] / n,
                "table_cells_per_page": c["cells"] / n,
                "asserted_by": "tests/test_gpu_fullsize.py::test_fullsize_*_oracle_parity[f16] (drift bounds 8x tighter than bf16's), tests/test_gpu_f16.py "
                               "(half-ulp operator parity, saturation), tests/test_gpu_e2e.py::test_headline_mode_agreement"}

    def e2e_agreement_leg(self):
        """What each arithmetic outputs, as fractions of the oracle chain's outputs: the two pages of the committed
        end-to-end fixture (tests/golden/e2e_page.npz -- boxes, token ids, cells, logical locations, table HTML of the composed fp32 oracle chain) through
        OcrTablePipeline.predict() in bf16, f16 and BF16X3, compared by tests/e2e_agreement.py.  The fixture's nets ARE the timed step's
        (synth_weights.conditioned_state_dicts).  `*_oracle_crops`: the table stage fed the oracle chain's layout regions -- a layout box that rounds
        one pixel differently is a different crop, i.e. a different input to a random-init net, and would otherwise be charged to the table arithmetic.
        A checker outside every timed region, like cpu_baseline."""
        L, eng = self.L, self.eng
        if not (self.layout_chain and self.pipe is not None and self.tsr is not None and not self.nas):
            return None
        tdir = os.path.join(REPO, "tests")
        if tdir not in sys.path:
            sys.path.insert(0, tdir)
        from e2e_agreement import agreement
        from e2e_synth import E2E_PAGES
        from pdf_table_amd.pipeline import OcrTablePipeline
        from pdf_table_amd.synth_pages import make_page
        from pdf_table_amd.tsr_stage import LoreConfig, TsrStage
        g = np.load(os.path.join(tdir, "golden", "e2e_page.npz"))
        pages = [make_page(i, PAGE)[0] for i in E2E_PAGES]
        tbs = [g[f"p{pi}_table_boxes"] for pi in range(len(pages))]
        out = {"fixture": "tests/golden/e2e_page.npz (2 pages, 3 tables; oracle chain in fp32 on synth_weights.conditioned_state_dicts, the timed step's nets)",
               "definition": "tests/e2e_agreement.py: fraction of the oracle's boxes / strings / cells / table HTML the engine reproduces"}
        runs = [("bf16", L.PT_PRECISION_BF16, None)]
        if self.x3_leg:
            runs.append(("bf16x3", L.PT_PRECISION_BF16X3, None))
        if getattr(self, "_f16", None) is not None:
            runs.append(("f16", L.PT_PRECISION_F16, self._f16))
        for mode, prec, other in runs:
            e = eng if other is None else other["eng"]
            st = (self.stage, self.rec, self.layout) if other is None else (other["stage"], other["rec"], other["layout"])
            pipe = OcrTablePipeline.from_engine(e, st[0], st[1], st[2], TsrStage(e, LoreConfig(task_type="wtw")), table_html=True)
            if other is None:
                eng.set_precision(prec)
            try:
                a = agreement(g, pipe.predict(pages), self.rec.label, tbs)
                b = agreement(g, pipe.predict(pages, table_boxes=[np.asarray(t) for t in tbs]), self.rec.label, tbs)
            finally:
                if other is None:
                    eng.set_precision(L.PT_PRECISION_BF16)
            out[mode] = a["frac"]
            out[mode + "_oracle_crops"] = {k: v for k, v in b["frac"].items() if k.startswith(("cells", "logi", "tables"))}
            out[mode + "_counts"] = {k: v for k, v in a.items() if k != "frac"}
        # The constructive form of `*_oracle_crops`: the layout net ALONE in the pair mode (LayoutStage.precision = BF16X3; OcrTablePipeline(layout_precision=
        # "fp32")) under an engine that otherwise computes in 16 bits -- the table crops are then the oracle chain's own, chained, at the price of
        # the layout stage's three passes.  f16 engine: its PicoDet blob is reloaded as a bf16 blob with the pair tiles (blobs carry their format per model).
        from pdf_table_amd.layout_stage import LayoutStage, PicodetConfig
        from pdf_table_amd.weights import pack_picodet
        mixed = [("bf16_layout_fp32", eng, (self.stage, self.rec), None)] if self.x3_leg else []
        if getattr(self, "_f16", None) is not None:
            f = self._f16
            f["eng"].load_weights(L.PT_MODEL_PICODET, pack_picodet(self.ysd, 5, x3=True))
            mixed.append(("f16_layout_fp32", f["eng"], (f["stage"], f["rec"]), f))
        for mode, e, st, other in mixed:
            lay = LayoutStage(e, PicodetConfig(task_type="en"), precision=L.PT_PRECISION_BF16X3)
            pipe = OcrTablePipeline.from_engine(e, st[0], st[1], lay, TsrStage(e, LoreConfig(task_type="wtw")), table_html=True)
            a = agreement(g, pipe.predict(pages), self.rec.label, tbs)
            out[mode] = {k: v for k, v in a["frac"].items() if k.startswith(("boxes_within", "strings_identical_on_2px", "cells", "logi", "tables"))}
            if other is not None:          # and what it costs: the same timed loop with that layout stage
                keep = other["layout"], other["pipe"]
                other["layout"] = lay
                other["pipe"] = OcrTablePipeline.from_engine(e, other["stage"], other["rec"], lay, other["tsr"], overlap_rec=False,
                                                             aux_layout=bool(self.args.aux_stream), tsr_on_aux=bool(self.args.aux_stream),
                                                             lookahead=int(os.environ.get("PT_PIPE_LOOKAHEAD", "1")))
                try:
                    with self.on_engine(other):
                        dt, _ = self.timed(8, 3)
                    out[mode]["pages_per_s"] = PAGES_PER_STEP * 8 / dt
                finally:
                    other["layout"], other["pipe"] = keep
        return out

    def parity_sample(self):
        """engine outputs for the page / lines the CPU-baseline leg runs through the oracle (checked THERE)"""
        torch, L, eng = self.torch, self.L, self.eng
        out = {}
        if self.nas:
            return out
        from pdf_table_amd.rec_stage import ctc_collapse
        page0 = self.pages[:1]
        for mode, prec in (("bf16", L.PT_PRECISION_BF16), ("bf16x3", L.PT_PRECISION_BF16X3)):
            if mode == "bf16x3" and not self.x3_leg:
                continue
            eng.set_precision(prec)
            try:
                prob, bm = eng.det_forward(page0, L.PT_DET_PRE_DB_PP, self.cfg.thresh)
                out["det_prob_" + mode] = prob[0].cpu().numpy()
                boxes = self.stage.boxes(prob, bm, (PAGE, PAGE))
                if mode == "bf16":
                    out["det_nboxes"] = len(boxes[0])
                    self._boxes0 = boxes
                    out["rec_quads"] = boxes[0]
                if self.rec is not None:
                    # both precisions (and the oracle, in the CPU leg) read the SAME quads: the bf16 boxes of page 0
                    ids, _ = self.rec.ids(page0, self._boxes0)
                    out["rec_ids_" + mode] = ids.cpu().numpy()
                if self.tsr is not None and len(self.table_boxes[0]):
                    # the table the CPU leg runs through the oracle: page 0, the generator's first table rectangle (crop frame)
                    rt = self.tsr(page0, [np.asarray(self.table_boxes[0][:1])])[0][0]
                    out["tsr_polys_" + mode] = np.asarray(rt["polygons"], np.float64).reshape(-1, 8)
            finally:
                eng.set_precision(L.PT_PRECISION_BF16)
        if self.rec is not None:      # configs[0] on the GPU: one 640x640 page, det + rec, synchronous, bf16
            p640 = self.pages[:1, 192:832, 192:832].contiguous()
            st = self.stage
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                b = st(p640)
                ids, _ = self.rec.ids(p640, b)
                ctc_collapse(ids.cpu().numpy())
                torch.cuda.synchronize()
                out["config0_ms"] = (time.perf_counter() - t0) * 1e3
                out["config0_lines"] = int(len(b[0]))
        return out

    def config(self, c, steps):
        nas, stages, args = self.nas, self.stages, self.args
        n = max(1, PAGES_PER_STEP * steps)
        chained = "det" in stages and "rec" in stages and not args.gt_chain
        return {"workload": ("BASELINE.json configs[2], full pipeline: " if set(stages) >= {"layout", "det", "rec", "tsr"} else "")
                            + ("PicoDet layout detection (resize to 800x608, LCNet + CSP-PAN + PicoHead, hard NMS) + "
                               if "layout" in stages else "")
                            + ("BASELINE.json configs[1] batched DB text detection (db_pp pre/post around "
                               + ("DB-ProxylessNAS [--det-backbone variant, not the BASELINE.json network]" if nas
                                  else "DB-ResNet18") + ", 1024x1024 synthetic pages -> 960x960 net input, boxes out)"
                               if "det" in stages else "")
                            + (" + CRNN text-line recognition (crop, resize, CRNN, arg-max, CTC collapse) of "
                               + ("the boxes the detection stage produced (software-pipelined by one step)" if chained
                                  else "the page generator's text-line rectangles") if "rec" in stages else "")
                            + (" + Lore table-structure recognition of the page's tables (wtw: warp to 1024x1024, "
                               "DLA-34+DCN, heat-map/corner decode with vertex snapping, 2 x 4-layer processor, "
                               "quads + logical locations)" if "tsr" in stages else "")
                            + (" + PP-LCNet text-line orientation of every text line and page orientation of every page "
                               "[opt-in stage, not part of BASELINE.json's metric]" if "cls" in stages else "")
                            + ("; timed loop = OcrTablePipeline.predict_stream() over device-resident 64-page batches (the product API: "
                               "PageResult objects out, three batches behind the input, drained inside the timed region)" if self.uses_pipeline() else
                               "; timed loop = bench.py's own schedule (run_private)")
                            + (" [DEVICE HALF ONLY]" if args.no_post else "")
                            + (" [recogniser on a second stream: --overlap-rec diagnostic]" if args.overlap_rec else "")
                            + (" [layout and the Lore processor on an auxiliary stream]" if args.aux_stream else "")
                            + "; weights are seeded random init"
                            + (" except one hand-built detector channel that finds the synthetic text (synth_weights._db_text_signal), "
                               "so detection -> recognition is chained by the detector's own boxes" if chained else "")
                            + (("; layout -> table structure is chained by the layout stage's own regions: label 'table', score >= 0.2, cropped "
                                "(the seeded PicoDet's stride-64 head branch is fitted to the generator's pages by tools/fit_layout_head.py -- it "
                                "memorises these pages, it is not a trained detector); the Lore nets are plain random init"
                                if self.layout_chain and self.uses_pipeline() else
                                "; the layout and Lore nets are plain random init, so the table regions are the page generator's ground truth")
                               if "tsr" in stages else ""),
                "pages_per_step_per_gpu": PAGES_PER_STEP, "distinct_pages_per_gpu": DISTINCT, "page": [PAGE, PAGE],
                "stages": stages, "parallelism": f"page-shard x{self.world}",
                "text_lines_per_page_generated": self.gt_lines_per_page,
                "text_lines_recognised_per_page": c["rec_lines"] / n,
                "tokens_per_page": c["tok"] / n,
                "boxes_per_page": c["boxes"] / n,
                "tables_per_page": (c["tables"] / n if "tables" in c else self.tables_per_page) if "tsr" in stages else 0,
                "tables_per_page_generated": self.tables_per_page if "tsr" in stages else 0,
                "table_regions_from": ("layout stage" if self.layout_chain and self.uses_pipeline() else "page generator") if "tsr" in stages else None,
                "layout_regions_per_page": c["layout"] / n,
                "table_cells_per_page": c["cells"] / n,
                "classified_lines_per_page": c["cls_lines"] / n,
                "weights": "seeded random init (reference state_dict layout)" + (" + text-signal channel in the detector" if chained else "")}


# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6000      # bytes of the ONE stdout line (the 20.8 KB line of round 5 was not parsed by the driver)


def _r(x, nd=4):
    """numbers to nd significant-ish decimals, containers recursively"""
    if isinstance(x, float):
        return round(x, nd) if abs(x) < 1e6 else float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out):
    """The stdout line: the contract keys, `roofline`, `cpu_baseline` and `summary`, numbers only (notes, accounting prose, count tables and the
    side legs live in bench_detail.json).  Kept under LINE_LIMIT bytes -- main() asserts it, tests/test_bench_line.py pins it on the stub runner."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    stages = cfg.get("stages") or []
    full = set(stages) >= {"layout", "det", "rec", "tsr"}
    line["config"] = {"workload": ("BASELINE.json configs[2]: full pipeline (PicoDet layout + DB-ResNet18 text detection + CRNN recognition + Lore table structure) "
                                   if full else "stages " + "+".join(stages) + " ")
                                  + f"over {cfg.get('pages_per_step_per_gpu')} device-resident synthetic {PAGE}x{PAGE} pages per GPU per step, "
                                  + "OcrTablePipeline.predict_stream(), seeded synthetic checkpoints",
                      # every short entry of the full config (numbers, short lists / strings); the prose stays in bench_detail.json
                      **{k: v for k, v in cfg.items() if k != "workload" and v is not None and
                         (isinstance(v, (int, float, bool)) or (isinstance(v, (list, tuple)) and len(v) <= 8) or (isinstance(v, str) and len(v) <= 40))}}
    roof = out.get("roofline")
    if roof:
        r = _pick(roof, "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "algorithmic_flop_per_launch")
        r["kernel"] = "3x3 implicit-GEMM convs (conv3x3_pipe_kernel, conv3x3_ws64_kernel, conv3x3_dma16_kernel, conv_igemm_kernel<3,*>)"
        db = roof.get("det_backbone")
        if isinstance(db, dict):
            r["det_backbone"] = {**_pick(db, "frac", "pages_per_s_det_only", "gflop_per_page", "error"),
                                 "net_only": _pick(db.get("net_only") or {}, "frac", "pages_per_s", "frac_of_executed_flop")}
        bc = roof.get("by_class")
        if isinstance(bc, dict) and "classes" in bc:
            def cls(v):
                f = v.get("frac_mfma_peak") if v.get("bound") == "mfma" else v.get("frac_hbm_peak")
                o = {"ms_per_step": v.get("ms_per_step"), "launches_per_step": v.get("launches_per_step")}
                if f is not None:
                    o["frac"], o["bound"] = f, v.get("bound")
                return o
            top = sorted(bc["classes"].items(), key=lambda kv: -kv[1].get("ms_per_step", 0))
            r["by_class"] = {"ms_per_step_kernels": bc.get("ms_per_step_kernels"), "launches_per_step": sum(v.get("launches_per_step", 0) for _, v in top),
                             "classes": {k: cls(v) for k, v in top[:9]}}
            tl = bc.get("top_labels") or {}
            conv = [(k, v) for k, v in tl.items() if k.startswith("conv3x3")]
            r["by_class"]["conv3x3_layers"] = {k.replace("conv3x3 ", ""): [v.get("ms_per_step"), v.get("frac_mfma_peak")] for k, v in conv[:10]}
        elif isinstance(bc, dict):
            r["by_class"] = _pick(bc, "error")
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, "value", "unit", "cores", "kind", "net_only_pages_per_s")
        c["cpu"] = _pick(cb.get("cpu") or {}, "model", "sockets", "cores", "threads")
        c["sample"] = (cb.get("sample") or "").split(";")[0][:200]
        c["config0"] = _pick(cb.get("config0") or {}, "cpu_s_per_page", "gpu_ms_per_page_bf16", "lines", "gpu_lines")
        g = cb.get("gpu_vs_oracle_on_the_sample") or {}
        c["parity_sample"] = _pick(g, "det_max_abs_dprob_bf16", "det_max_abs_dprob_bf16x3", "det_boxes_oracle_vs_gpu_bf16", "rec_token_ids_differing_bf16",
                                   "rec_token_ids_differing_bf16x3", "tsr_cells_oracle_engine_matched0p1px_matched1px_bf16x3")
        line["cpu_baseline"] = c
    s = dict(out.get("summary") or {})
    hp = out.get("host_pages")
    if isinstance(hp, dict) and "ratio_to_value" in hp:
        s["host_pages_ratio"] = hp["ratio_to_value"]
    oe = out.get("one_eighth_host")
    if isinstance(oe, dict):
        s["one_eighth_host"] = _pick(oe, "ratio_to_value", "host_pages_ratio_to_value", "cores", "of_cores", "error")
    for k, name, rate in (("mtl_tabnet", "mtl_tabnet_tables_per_s", "tables_per_s"), ("convnext_vit_recogniser", "convnext_vit_lines_per_s", "lines_per_s"),
                          ("onnx_recogniser", "onnx_rec_lines_per_s", "lines_per_s")):
        leg = out.get(k)
        if isinstance(leg, dict):     # --legs all: one rate per precision mode
            s[name] = {m: v[rate] for m, v in leg.items() if isinstance(v, dict) and rate in v} or _pick(leg, "error")
    if isinstance(out.get("overlap_rec"), dict):
        s["overlap_rec_pages_per_s"] = out["overlap_rec"].get("pages_per_s")
    s["detail"] = "bench_detail.json (every leg, notes, count tables); side legs: --legs all"
    line["summary"] = s
    line = _r(line)
    for k in ("value", "ms_per_step"):      # the contract's own numbers keep every digit (value == pages / time to the last bit)
        if k in out:
            line[k] = out[k]
    return line


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    self_launch(args, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus {args.gpus} does it itself)")
    stub = os.environ.get("PT_BENCH_STUB") in ("1", "host")
    import torch
    if world > 1:      # N ranks share the host: keep each rank's CPU-side torch / BLAS work inside its share of the cores
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    dist = None
    # PT_BENCH_FORCE_DIST=1: initialise RCCL and take the broadcast / barrier / all-reduce path even with one rank
    # (exercises the N > 1 code on a single-GPU box; launch with torchrun --nproc-per-node 1 or set MASTER_PORT)
    use_dist = world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1"
    saved_stdout = None
    rccl_init_s = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f"rank {rank}: no GPU {local_rank} (visible: {torch.cuda.device_count()})")
            torch.cuda.set_device(local_rank)
            # librccl prints its version banner ("RCCL version : ...", five lines) on STDOUT when the first communicator is created (measured:
            # profiles/r05/force_dist_1gpu.txt).  The contract is ONE JSON line on stdout, so file descriptor 1 points at stderr until rank 0 prints it
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            t_init = time.perf_counter()
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()          # the communicator is created lazily: by the first collective
            torch.cuda.synchronize()
            rccl_init_s = time.perf_counter() - t_init

    runner = StubRunner(args, rank, world) if stub else HipRunner(args, rank, local_rank, world, dist)

    def barrier():
        runner.sync()
        if dist is not None:
            dist.barrier()
        runner.sync()

    if not stub and args.precision != "bf16":
        runner.eng.set_precision(runner.L.PT_PRECISION_BF16X3 if args.precision == "bf16x3" else runner.L.PT_PRECISION_F16X2)
    # Engine spin-up, part of initialisation and NOT one of the W warm-up steps: while this process imported torch, generated pages and packed weights the
    # GPU idled at its low clocks; the first second of work runs in the ramp (a first process on a fresh box read 554 pages/s with W = 3, 604-606 in
    # the next two processes).  PT_BENCH_SPINUP steps (default 6, ~0.6 s) of the same pipeline precede the contract's W untimed + K timed steps.
    spin = 0 if stub else int(os.environ.get("PT_BENCH_SPINUP", "6"))
    if spin > 0:
        runner.run(spin)
    runner.run(args.warmup)
    barrier()
    # HIP events around the launches of the roofline's kernel class only (mode 2 + class 0 = the 3x3 convs): an event pair per
    # launch of every class (PT_BENCH_PROF=1, fills all_kernel_classes_ms) costs 1-3 % of the step in idle GPU time
    prof_mode = int(os.environ.get("PT_BENCH_PROF", "2"))
    if not stub:
        runner.eng.profile_enable(prof_mode)
    t0 = time.perf_counter()
    counts = runner.run(args.steps, count=True)
    barrier()
    dt = time.perf_counter() - t0
    prof = None
    if not stub:
        prof = runner.eng.profile_read()
        runner.eng.profile_enable(False)
        if runner.trace is not None and rank == 0:
            runner.trace.setdefault("det_boxes_parts", {a: round(b, 3) for a, b in getattr(runner.stage, "timing", {}).items()})
            print("[bench trace] host seconds over warm-up + timed steps:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in runner.trace.items()},
                  file=sys.stderr)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=None if stub else runner.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_pages = world * (runner.host["pages"] if getattr(runner, "host", None) else PAGES_PER_STEP) * args.steps
        out = {"metric": "pages/s", "value": total_pages / dt, "unit": "pages/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": runner.config(counts, args.steps)}
        out["config"]["spinup_steps_before_warmup"] = spin
        if prof is not None:
            c3 = prof["conv3x3"]
            achieved = (c3["flop"] / (c3["ms"] * 1e-3)) / 1e12 if c3["ms"] > 0 else 0.0
            traffic = None
            try:   # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_summary.py)
                with open(os.path.join(REPO, "profiles", "pmc_latest.json")) as f:
                    pm = json.load(f)
                tot = nl = 0.0
                for kname, rec_ in pm.items():       # every 3x3 conv kernel variant, launch-weighted
                    if ("conv_igemm_kernel<3," in kname or "conv3x3_dma" in kname or "conv3x3_pipe" in kname or "conv3x3_ws64" in kname) and "hbm_bytes_per_launch" in rec_:
                        n_ = rec_["FETCH_SIZE"]["dispatches"]
                        tot += rec_["hbm_bytes_per_launch"] * n_
                        nl += n_
                traffic = tot / nl if nl else None
            except Exception:
                traffic = None
            roof = {"bound": "mfma", "kernel": "conv3x3_pipe_kernel / conv3x3_ws64_kernel / conv3x3_dma16_kernel / conv_igemm_kernel<3,*> (3x3 implicit-GEMM convolutions of all stages)",
                    "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_note": "bytes/launch over the 3x3 conv kernels, PMC passes of profiles/pmc_latest.json "
                                                        "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                    "launches": c3["launches"], "avg_launch_ms": c3["ms"] / max(1, c3["launches"]),
                    "algorithmic_flop_per_launch": c3["flop"] / max(1, c3["launches"]),
                    "flop_accounting": "2 x output pixels x REAL output channels x Cin x 9 per launch (padded GEMM columns are not "
                                       "counted); row-limited sparse-head launches are credited the rows below the device limit "
                                       "they read back, and 1/9 of those (a 3x3 patch yields one used pixel); ragged CRNN launches are "
                                       "credited the 32-column row-tiles they multiply (from the round-4 binary on the padding half of a line's "
                                       "last 64-column conv3.* patch is neither multiplied nor credited: -3.9 of 31 TFLOP per step for -3.0 ms, "
                                       "i.e. frac reads 0.31 where the whole-patch credit of the earlier rounds would read 0.35 on the same run)",
                    "events": "3x3 class only" if prof_mode == 2 else "every launch"}
            if prof_mode == 1:       # PT_BENCH_PROF=1: events around every launch
                roof["all_kernel_classes_ms"] = {k: v["ms"] for k, v in prof.items()}
                roof["all_kernel_classes_flop"] = {k: v["flop"] for k, v in prof.items()}
            out["roofline"] = roof
    if not stub and world == 1 and (args.host_pages_leg or not (args.no_extra_legs or args.by_class_only)) and not args.no_post:
        try:
            # as many timed steps as the headline region: a 10-step leg reads ~3 % lower than a 20-step one on the SAME resident input (one pipeline fill + drain
            # in half the time), which is most of the 0.96 the round-4 driver line showed for this ratio
            leg = runner.host_pages_leg(steps=max(10, args.steps), warm=5)
        except Exception as e:      # noqa: BLE001 -- a diagnostic leg
            leg = {"error": f"{type(e).__name__}: {e}"[:300]}
        if leg is not None:
            if "pages_per_s" in leg:
                leg["ratio_to_value"] = leg["pages_per_s"] / out["value"]
            out["host_pages"] = leg
    # extra legs: every rank runs them (they are outside the timed region; rank 0 reports its own).  A diagnostic leg that raises is reported
    # as {"error": ...} on the line and the engine goes back to the timed region's precision: it must never take the headline down with it
    def guarded(fn):
        try:
            return fn()
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            try:
                runner.eng.set_precision(runner.L.PT_PRECISION_BF16)
                runner.torch.cuda.synchronize()
            except Exception:       # noqa: BLE001
                pass
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    if not stub and args.by_class_only and world == 1:
        leg = guarded(runner.by_class_leg)
        if rank == 0 and leg is not None:
            out["roofline"]["by_class"] = leg
    if not stub and not args.no_extra_legs and not args.by_class_only and world == 1:
        if "det" in runner.stages and not runner.nas:
            leg = guarded(runner.det_only_leg)
            if rank == 0:
                out["roofline"]["det_backbone"] = leg
        if len(runner.stages) > 1 and not args.no_post:
            leg = guarded(runner.by_class_leg)
            if rank == 0 and leg is not None:
                out["roofline"]["by_class"] = leg
        if len(runner.stages) > 1 and args.legs == "all":
            leg = guarded(runner.overlap_leg)
            if rank == 0 and leg is not None:
                out["overlap_rec"] = leg
        if runner.x3_leg and not args.no_post:
            leg = guarded(lambda: runner.x3_leg_run(steps=max(8, args.steps)))      # as many timed steps as the headline region
            if rank == 0:
                leg["bf16_pages_per_s"] = out["value"]
                out["tolerance_mode"] = leg
        if rank == 0 and not args.no_post and "tsr" in runner.stages and "layout" in runner.stages:
            leg = guarded(lambda: runner.f16_leg_run(steps=max(8, args.steps), warm=max(3, args.warmup)))      # a fresh engine: as many untimed steps as the headline's
            if leg is not None:
                leg["ratio_to_bf16"] = leg["pages_per_s"] / out["value"] if "pages_per_s" in leg else None
                out.setdefault("tolerance_mode", {})["f16"] = leg
                out["tolerance_mode"]["f16_pages_per_s"] = leg.get("pages_per_s")
        if rank == 0 and not args.no_post:
            leg = guarded(runner.e2e_agreement_leg)
            if leg is not None:
                out.setdefault("tolerance_mode", {})["e2e_agreement"] = leg
            f16 = getattr(runner, "_f16", None)
            if f16 is not None:       # the second engine's arenas and weights go back before the transformer legs allocate theirs
                try:
                    f16["eng"].close()
                except Exception:      # noqa: BLE001
                    pass
                runner._f16 = None
        if "rec" in runner.stages and not args.no_post and args.legs == "all":
            leg = guarded(runner.convnext_vit_leg)
            if rank == 0 and leg is not None:
                out["convnext_vit_recogniser"] = leg
        if "rec" in runner.stages and not args.no_post and rank == 0 and args.legs == "all":
            leg = guarded(runner.onnx_rec_leg)
            if leg is not None:
                out["onnx_recogniser"] = leg
        if "tsr" in runner.stages and not args.no_post and args.legs == "all":
            leg = guarded(runner.mtl_tabnet_leg)
            if rank == 0 and leg is not None:
                out["mtl_tabnet"] = leg
    if rank == 0:
        if not stub and world == 1 and not args.no_cpu_baseline and "det" in runner.stages:
            r = runner
            gpu = r.parity_sample()
            out["cpu_baseline"] = cpu_baseline(r.sd, r.pages_np[:2], r.cfg, r.csd if r.rec is not None else None,
                                               tsr=(r.lsd, r.psd, r.table_boxes, r.tables_per_page) if r.tsr is not None else None,
                                               layout=r.ysd if r.layout is not None else None, gpu=gpu)
        if not stub and not args.no_extra_legs and not args.by_class_only and world == 1 and len(runner.stages) > 1 and not args.no_post:
            # last: the child needs the GPU's memory, so this process gives its engine (activation arenas, weights) and torch's cache back first
            try:
                runner.eng.close()
                runner.torch.cuda.empty_cache()
            except Exception:      # noqa: BLE001
                pass
            out["one_eighth_host"] = one_eighth_host_leg(args, out["value"])
        if rccl_init_s is not None:
            out["config"]["rccl_init_s"] = round(rccl_init_s, 3)
        # The driver records the TAIL of this line: the figures the targets are stated on go last, compactly
        def dig(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 4) if isinstance(d, float) else d
        summ = {"pages_per_s_bf16": round(out["value"], 1), "roofline_frac_3x3": dig(out, "roofline", "frac"),
                "det_backbone_frac": dig(out, "roofline", "det_backbone", "frac"),
                "det_backbone_net_only_frac": dig(out, "roofline", "det_backbone", "net_only", "frac"),
                "pages_per_s_f16": dig(out, "tolerance_mode", "f16_pages_per_s"), "pages_per_s_bf16x3": dig(out, "tolerance_mode", "pages_per_s")}
        ag = dig(out, "tolerance_mode", "e2e_agreement") or {}
        for mode in ("bf16", "f16", "bf16x3"):
            if mode in ag:
                summ["agreement_" + mode] = {"boxes_2px": ag[mode].get("boxes_within_2px"), "strings": ag[mode].get("strings_identical_on_2px_quads"),
                                             "cells_1px": ag[mode].get("cells_matched_1px"),
                                             "cells_1px_oracle_crops": (ag.get(mode + "_oracle_crops") or {}).get("cells_matched_1px"),
                                             "tables_html_identical": ag[mode].get("tables_html_identical")}
        for mode in ("f16_layout_fp32", "bf16_layout_fp32"):
            if mode in ag:
                summ["agreement_" + mode] = {"cells_1px": ag[mode].get("cells_matched_1px"), "strings": ag[mode].get("strings_identical_on_2px_quads"),
                                             "tables_html_identical": ag[mode].get("tables_html_identical"), "pages_per_s": dig(ag[mode], "pages_per_s")}
        out["summary"] = summ
        # the full record: a sidecar file and stderr; stdout gets the compact projection only (the driver parses ONE line of at most a few KB)
        try:
            with open(args.detail_out, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print(f"[bench] could not write {args.detail_out}: {e}", file=sys.stderr)
        print("[bench detail] " + json.dumps(out), file=sys.stderr)
        line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(line) < LINE_LIMIT, f"bench line is {len(line)} bytes (limit {LINE_LIMIT}): trim compact_line()"
        sys.stdout.flush()
        sys.stderr.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(line)
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
