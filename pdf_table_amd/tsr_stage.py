"""Batched table-structure recognition stage (Lore): geometry and result shaping on the host, everything else on the GPU.

Replaces, for a batch of table crops of resident pages, the reference's per-table chain
``TableLorePreProcessor.process`` (lore/processer_lore.py:66-109) -> ``LoreModel.forward`` (lore/modeling_lore.py:125-194:
DLASeg detector, ``process_detect_output``, ``LoreProcessModel``) -> ``TableLorePostProcessor.__call__``
(processer_lore.py:163-188).  The host computes one 2x3 affine map per table (float64, like cv2.getAffineTransform)
and, after the device decode, maps the cell quads back to source pixels and rounds the logical locations.
"""
from __future__ import annotations

import os
import time

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from .streams import shared_stream

from . import lib as L
from .engine import TSR_TABLE_DTYPE, HipEngine
from .table_html import structure_html

__all__ = ["LoreConfig", "TsrStage", "lore_geometry", "affine_from_center_scale", "affine_upper_left", "invert_affine",
           "transform_quads", "process_logic_output"]


@dataclass
class LoreConfig:
    """The fields of lore/configuration_lore.py:29-116 that the inference path reads."""
    task_type: str = "wtw"
    model_name: str = "Lore"
    model_path: str = ""
    backbone: str = "DLA-34"
    resolution: Tuple[int, int] = (1024, 1024)
    stacking_layers: int = 4
    tsfm_layers: int = 4
    upper_left: bool = False
    wiz_2dpe: bool = False
    wiz_stacking: bool = True
    wiz_rev: bool = True
    vis_thresh: float = 0.2

    def __post_init__(self):
        if self.task_type == "wireless":                  # configuration_lore.py:66-78
            self.backbone, self.resolution = "ResNet-18", (768, 768)
            self.stacking_layers = self.tsfm_layers = 4
            self.upper_left, self.wiz_2dpe, self.wiz_rev, self.vis_thresh = True, True, False, 0.2
        elif self.task_type == "wtw":                     # configuration_lore.py:79-92
            self.backbone, self.resolution = "DLA-34", (1024, 1024)
            self.stacking_layers = self.tsfm_layers = 4
            self.upper_left, self.wiz_2dpe, self.wiz_rev, self.vis_thresh = False, False, True, 0.2
        else:                                             # "ptn" and anything else (:94-108)
            self.backbone, self.resolution = "DLA-34", (512, 512)
            self.stacking_layers = self.tsfm_layers = 3
            self.upper_left, self.wiz_2dpe, self.wiz_rev, self.vis_thresh = False, True, False, 0.35


def _affine_3pt(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """the 2x3 map through three point pairs (what cv2.getAffineTransform solves), float64"""
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        a[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def affine_from_center_scale(center, scale, out_size, inv: bool = False) -> np.ndarray:
    """get_affine_transform(center, scale, 0, out_size, inv) of lineless_table_process.py:403-438 (rot = 0, no shift):
    three float32 anchor points on each side -- centre, centre + (0, -scale/2), and their perpendicular third."""
    s = np.float32(scale)
    dst_w, dst_h = out_size
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = np.asarray(center) + np.asarray([0.0, s * -0.5])
    dst[0] = (dst_w * 0.5, dst_h * 0.5)
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    return _affine_3pt(dst, src) if inv else _affine_3pt(src, dst)


def affine_upper_left(center, scale, out_size, inv: bool = False) -> np.ndarray:
    """get_affine_transform_upper_left (lineless_table_process.py:441-468): the origin stays at the top-left corner;
    (cx, scale) -> (0, out_w) when cx >= cy, else (scale, cy) -> (out_w, 0); third point perpendicular."""
    s = np.float32(scale)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    if center[0] < center[1]:
        src[1] = (s, center[1])
        dst[1] = (out_size[0], 0)
    else:
        src[1] = (center[0], s)
        dst[1] = (0, out_size[0])
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    return _affine_3pt(dst, src) if inv else _affine_3pt(src, dst)


def invert_affine(m: np.ndarray) -> np.ndarray:
    """the inverse map cv2.warpAffine derives from M (float64, same operation order)"""
    m = np.asarray(m, np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11
    m[1] *= -d
    m[3] *= -d
    m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m.reshape(2, 3)


def lore_geometry(crop_h: int, crop_w: int, inp_h: int, inp_w: int, upper_left: bool = False):
    """-> (minv 2x3 for pt_tsr_table, meta int64 [cx, cy, s, in_h, in_w, out_h, out_w]) -- processer_lore.py:74-126:
    c = (w/2, h/2) float32 (or (0, 0) with upper_left), s = max(h, w); meta is cast to int64 (fractions of c are
    dropped, as in the reference)."""
    s = max(crop_h, crop_w) * 1.0
    if upper_left:
        c = np.array([0, 0], dtype=np.float32)
        trans = affine_upper_left(c, s, (inp_w, inp_h))
    else:
        c = np.array([crop_w / 2.0, crop_h / 2.0], dtype=np.float32)
        trans = affine_from_center_scale(c, s, (inp_w, inp_h))
    meta = np.array([c[0], c[1], s, inp_h, inp_w, inp_h // 4, inp_w // 4]).astype(np.int64)
    return invert_affine(trans), meta


def transform_quads(quads: np.ndarray, meta: np.ndarray, upper_left: bool = False) -> np.ndarray:
    """ctdet_4ps_post_process[_upper_left] (lineless_table_process.py:489-533): the four vertices of every quad through
    the inverse map built from the int64 meta; float64 matrix x float32 point, stored back as float32."""
    t = (affine_upper_left if upper_left else affine_from_center_scale)(meta[:2], meta[2], (meta[6], meta[5]), inv=True)
    p = quads.astype(np.float32).astype(np.float64).reshape(-1, 4, 2)
    out = np.empty_like(p)
    out[..., 0] = t[0, 0] * p[..., 0] + t[0, 1] * p[..., 1] + t[0, 2]
    out[..., 1] = t[1, 0] * p[..., 0] + t[1, 1] * p[..., 1] + t[1, 2]
    return out.reshape(-1, 8).astype(np.float32)


def process_logic_output(logi: np.ndarray) -> np.ndarray:
    """lineless_table_process.py:658-663: fractional part > 0.5 rounds up, otherwise down."""
    fl = np.floor(logi)
    return np.where(logi - fl > 0.5, fl + 1, fl).astype(np.float32)


class TsrStage:
    def __init__(self, eng: HipEngine, config: Optional[LoreConfig] = None, micro_batch: int = 8, bgr: bool = True,
                 with_html: bool = True):
        self.eng = eng
        self.config = config or LoreConfig()
        self.micro_batch = micro_batch
        self.bgr = bgr
        self.with_html = with_html      # also emit 'structure_str_list' (show_results :292-303)
        self._copy_stream = None
        self.wait_s = 0.0
        self.fused_decode = os.environ.get("PT_TSR_FUSED", "1") != "0"     # pt_tsr_forward_decode vs the two-call path

    def tables(self, page_shape: Tuple[int, int], boxes_per_page: Sequence[np.ndarray]):
        """integer table boxes [k,4] (x1,y1,x2,y2) per page, cropped like crop_image_by_box
        (utils/ocr/ocr_common_utils.py:269-284: img[y1:y2, x1:x2]) -> (pt_tsr_table records, metas)."""
        ph, pw = page_shape
        inp_h, inp_w = self.config.resolution
        recs, metas = [], []
        for pi, boxes in enumerate(boxes_per_page):
            for b in np.asarray(boxes).reshape(-1, 4):
                x1, y1, x2, y2 = (int(v) for v in b)
                x1, y1 = max(x1, 0), max(y1, 0)
                x2, y2 = min(x2, pw), min(y2, ph)
                cw, ch = x2 - x1, y2 - y1
                if cw <= 0 or ch <= 0:
                    raise ValueError(f"empty table crop {b.tolist()} on a {ph}x{pw} page")
                minv, meta = lore_geometry(ch, cw, inp_h, inp_w, self.config.upper_left)
                r = np.zeros((), dtype=TSR_TABLE_DTYPE)
                r["minv"], r["page"], r["x0"], r["y0"], r["crop_w"], r["crop_h"] = minv.reshape(6), pi, x1, y1, cw, ch
                recs.append(r)
                metas.append(meta)
        return (np.array(recs, dtype=TSR_TABLE_DTYPE) if recs else np.zeros(0, dtype=TSR_TABLE_DTYPE)), metas

    def start(self, pages: torch.Tensor, tables: np.ndarray):
        """device half 1, nothing synchronises: per micro-batch warp -> DLA-34+DCN -> decode; returns the pending state"""
        cfg = self.config
        inp_h, inp_w = cfg.resolution
        pending = []
        # balanced micro-batches (87 tables at micro_batch 80 -> 44 + 43, not 80 + 7: a 7-table launch leaves most CUs idle)
        nmb = max(1, -(-len(tables) // self.micro_batch))
        size = -(-len(tables) // nmb) if len(tables) else 1
        fused = cfg.backbone != "ResNet-18" and self.fused_decode
        if fused and nmb > 1:
            # all micro-batches decode into slices of one allocation, so that the processor (a chain of ~60 small launches)
            # runs ONCE over all tables of the call instead of once per micro-batch
            nt = len(tables)
            dev = pages.device
            counts_all = torch.zeros((nt,), dtype=torch.int32, device=dev)
            dets_all = torch.empty((nt, L.PT_TSR_MAX_CELLS, 9), dtype=torch.float32, device=dev)
            logi_all = torch.empty((nt, L.PT_TSR_MAX_CELLS, 256), dtype=torch.float32, device=dev)
        for i in range(0, len(tables), size):
            tb = tables[i:i + size]
            x = self.eng.tsr_preprocess(pages, tb, inp_h, inp_w, bgr=self.bgr)
            if fused:      # one call, ax / cr heads only where the decode reads them
                out = (counts_all[i:i + len(tb)], dets_all[i:i + len(tb)], logi_all[i:i + len(tb)]) if nmb > 1 else None
                counts, dets, logi = self.eng.tsr_forward_decode(x, wiz_rev=cfg.wiz_rev, vis_thresh=cfg.vis_thresh, sync=False, out=out)
            else:
                heads = self.eng.tsr_forward_net(x, wireless=cfg.backbone == "ResNet-18")
                counts, dets, logi = self.eng.tsr_decode(heads, wiz_rev=cfg.wiz_rev, vis_thresh=cfg.vis_thresh, sync=False)
            pending.append((i, len(tb), counts, dets, logi))
        if fused and nmb > 1:
            pending = [(0, len(tables), counts_all, dets_all, logi_all)]
        # the cell counts go to pinned host memory right behind the decode, with an event of their own: process() then waits
        # for THESE tables' decode only -- a `.cpu()` there would wait for everything queued on the stream since (a pipelined
        # caller has queued the next batch by then)
        out = []
        for (i, nt, counts, dets, logi) in pending:
            host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            out.append((i, nt, counts, dets, logi, host, ev))
        return out

    def process(self, pending):
        """device half 2: cell counts from their pinned copy (waits for half 1 of these tables only when it is still running), the
        processor over all cells of each micro-batch, then the valid rows to pinned host memory on a copy stream behind
        an event -- nothing here waits for work queued after start()."""
        cfg = self.config
        staged = []
        for (i, nt, counts_d, dets, logi, counts_h, counts_ev) in pending:
            t0 = time.perf_counter()
            counts_ev.synchronize()
            self.wait_s += time.perf_counter() - t0      # host seconds spent waiting for the decode (diagnostics: pipeline.metric)
            counts = counts_h.numpy().copy()
            logic, stacked = self.eng.tsr_process(logi, dets, counts, use_2dpe=cfg.wiz_2dpe)
            staged.append((i, nt, counts, dets, logic, stacked))
        if not staged:
            return [], None
        if self._copy_stream is None:
            self._copy_stream = shared_stream(staged[0][3].device, "tsr_copy")
        ready = torch.cuda.Event()
        ready.record()
        host = []
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            for (i, nt, counts, dets, logic, stacked) in staged:
                nmax = max(1, int(counts.max()) if len(counts) else 1)
                bufs = []
                for src in (dets, logic, stacked):
                    dst = torch.empty((src.shape[0], nmax, src.shape[2]), dtype=torch.float32, pin_memory=True)
                    dst.copy_(src[:, :nmax], non_blocking=True)
                    src.record_stream(self._copy_stream)
                    bufs.append(dst)
                host.append((i, nt, counts, bufs))
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return host, done

    def collect(self, processed, metas: List[np.ndarray], offsets: Optional[np.ndarray] = None) -> List[Dict]:
        """host half: quads back to source pixels, logical rounding -> per table {'polygons' f32 [n,8], 'logi' f32 [n,4]
        (integer valued), 'logic_axis', 'stacked_axis' (unrounded), 'scores'} like TableLorePostProcessor's result dict.
        ``offsets`` int [tables, 2]: the crop's (x0, y0) on its page, added to every vertex so that the quads are in page
        pixels -- what the reference's system path does after the per-crop call (convert_table_sep_to_merge ->
        box_list_move_point, pdf_table/table_common.py:1811-1825); None keeps them relative to the crop (the task's own
        result, processer_lore.py:175-182)."""
        cfg = self.config
        host, done = processed
        if done is not None:
            done.synchronize()
        out: List[Dict] = []
        for (i, nt, counts, bufs) in host:
            dets_h, logic_h, stacked_h = (b.numpy() for b in bufs)
            for k in range(nt):
                n = int(counts[k])
                if n == 0:       # LoreModel.forward's empty case (modeling_lore.py:171-173)
                    out.append({"polygons": np.zeros((1, 8), np.float32), "logi": np.zeros((1, 4), np.float32),
                                "logic_axis": np.zeros((1, 4), np.float32), "stacked_axis": np.zeros((1, 4), np.float32),
                                "scores": np.zeros((0,), np.float32)})
                    continue
                final = stacked_h[k, :n] if cfg.wiz_stacking else logic_h[k, :n]
                polys = transform_quads(dets_h[k, :n, :8], metas[i + k], cfg.upper_left)
                if offsets is not None:
                    # box_list_move_point works on .tolist() values: python floats + the rounded int corner -> float64
                    polys = polys.astype(np.float64) + np.tile(np.asarray(offsets[i + k], np.float64), 4)[None]
                r = {"polygons": polys,
                     "logi": process_logic_output(final), "logic_axis": logic_h[k, :n].copy(),
                     "stacked_axis": stacked_h[k, :n].copy(), "scores": dets_h[k, :n, 8].copy()}
                if self.with_html:       # table_html.table_cells_from_logits(polygons, logi) gives the cell objects on demand
                    r["structure_str_list"] = [structure_html(r["polygons"], r["logi"])]
                out.append(r)
        return out

    def finish(self, pending, metas: List[np.ndarray], offsets: Optional[np.ndarray] = None) -> List[Dict]:
        return self.collect(self.process(pending), metas, offsets)

    def run(self, pages: torch.Tensor, tables: np.ndarray, metas: List[np.ndarray], offsets: Optional[np.ndarray] = None) -> List[Dict]:
        return self.finish(self.start(pages, tables), metas, offsets)

    def regroup(self, flat: List[Dict], boxes_per_page: Sequence[np.ndarray]) -> List[List[Dict]]:
        res, o = [], 0
        for b in boxes_per_page:
            k = len(np.asarray(b).reshape(-1, 4))
            res.append(flat[o:o + k])
            o += k
        return res

    def __call__(self, pages: torch.Tensor, boxes_per_page: Sequence[np.ndarray], page_frame: bool = False) -> List[List[Dict]]:
        """page_frame=True: quads in page pixels (the crop's clamped x0, y0 added), like the reference's merged result"""
        tables, metas = self.tables(tuple(pages.shape[1:3]), boxes_per_page)
        offs = np.stack([tables["x0"], tables["y0"]], 1).astype(np.float32) if page_frame and len(tables) else None
        flat = self.run(pages, tables, metas, offs) if len(tables) else []
        return self.regroup(flat, boxes_per_page)
