"""Python handle on one MI355X engine (one per GPU / per process).

PyTorch is used here only as plumbing: it owns the caller-side device buffers (inputs / outputs) and
the HIP stream; every compute call goes through the C ABI of libpdftable_hip.so.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import threading
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

__all__ = ["HipEngine", "REC_LINE_DTYPE", "CLS_IMAGE_DTYPE", "REC_PP_ITEM_DTYPE"]

# mirrors struct pt_rec_line in include/pdftable_hip.h (88 bytes)
TSR_TABLE_DTYPE = np.dtype([("minv", "<f8", (6,)), ("page", "<i4"), ("x0", "<i4"), ("y0", "<i4"),
                            ("crop_w", "<i4"), ("crop_h", "<i4"), ("reserved", "<i4")])   # struct pt_tsr_table
CLS_IMAGE_DTYPE = np.dtype([("offset", "<i8"), ("h", "<i4"), ("w", "<i4")])      # pt_cls_image
REC_LINE_DTYPE = np.dtype([("minv", np.float64, (9,)), ("page", np.int32), ("crop_w", np.int32), ("crop_h", np.int32),
                           ("reserved", np.int32)])


REC_PP_ITEM_DTYPE = np.dtype([("line", "<i4"), ("resized_w", "<i4"), ("img_w", "<i4"), ("reserved", "<i4"),
                              ("out_off", "<i8")])                               # struct pt_rec_pp_item (24 bytes)


def _upload(arr: np.ndarray, device) -> torch.Tensor:
    """small host array -> device through pinned memory, asynchronously: a pageable `.to(device)` would block the host
    until everything already queued on the stream has run"""
    return torch.from_numpy(np.ascontiguousarray(arr)).pin_memory().to(device, non_blocking=True)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class HipEngine:
    def __init__(self, device: int = 0):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.PtError("no HIP device visible: the engine has no CPU fallback")
        self.device = int(device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        L.check(self.lib.pt_engine_create(self.device, C.byref(h)), "pt_engine_create")
        self._h = h
        self._tdev = torch.device("cuda", self.device)
        self.precision = L.PT_PRECISION_BF16
        # Precision is engine-wide HOST state that every call reads when it is queued (the dispatcher picks the pt_bf16 / pt_f16 namespace and the
        # pair layout from it).  The C ABI leaves serialising calls on one engine to the caller; this handle does it: every public method takes
        # this re-entrant lock, and precision_scope() holds it across {set, queue the stage's calls, restore}, so that a call from another host
        # thread (MtlStage.stream's worker, a user's own) can never be queued in somebody else's temporary precision.
        self._lock = threading.RLock()

    @contextlib.contextmanager
    def precision_scope(self, precision: Optional[int]):
        """``with eng.precision_scope(L.PT_PRECISION_BF16X3): eng.layout_forward(...)`` -- the calls QUEUED inside run in `precision` (None: no
        change), the engine's own precision is back afterwards; other threads' engine calls wait outside the scope."""
        with self._lock:
            keep = self.precision
            if precision is not None and int(precision) != keep:
                self.set_precision(int(precision))
            try:
                yield
            finally:
                if self.precision != keep:
                    self.set_precision(keep)

    def set_precision(self, precision: int):
        """L.PT_PRECISION_BF16 (throughput), L.PT_PRECISION_BF16X3 (fp32-class parity mode: three bf16 passes), L.PT_PRECISION_F16X2 (the
        same activation pairs against single fp16 weights: two fp16 passes, weight rounding ~1.5e-4 of the logit scale) or L.PT_PRECISION_F16
        (single-pass IEEE half, the reference's own GPU arithmetic: same speed as BF16 with 11 significant bits instead of 8; the weight blobs
        must be packed with ``fmt="f16"`` -- ``weight_fmt`` -- and the 16-bit tensors of the API are torch.float16)."""
        L.check(self.lib.pt_engine_set_precision(self._h, int(precision)), "pt_engine_set_precision")
        self.precision = int(precision)

    @property
    def split(self) -> bool:
        """activations cross the ABI as (hi | lo) bf16 pairs (twice the channels): the two pair modes"""
        return self.precision in (L.PT_PRECISION_BF16X3, L.PT_PRECISION_F16X2)

    @property
    def act_dtype(self):
        """torch dtype of the 16-bit activation tensors of the current precision (csrc/act16.h): float16 under PT_PRECISION_F16, else bfloat16"""
        return torch.float16 if self.precision == L.PT_PRECISION_F16 else torch.bfloat16

    @property
    def weight_fmt(self) -> str:
        """the ``fmt`` argument weights.pack_* need for blobs this engine will accept under its current precision"""
        return "f16" if self.precision == L.PT_PRECISION_F16 else "bf16"

    def set_mtl_kv_fp8(self, on: bool):
        """MtlTabNet, bf16 mode: stream the structure loop's source-attention keys / values as fp8 (half the bytes of the loop's dominant
        HBM stream; a throughput option with recorded drift, off by default)"""
        L.check(self.lib.pt_engine_set_mtl_kv_fp8(self._h, 1 if on else 0), "pt_engine_set_mtl_kv_fp8")

    def set_dcn_mfma(self, on):
        """Lore detector, bf16 mode: deformable convolutions with the bilinear blend on the matrix pipe (on) or on the VALU with fp32
        weights (off, the default); see include/pdftable_hip.h"""
        L.check(self.lib.pt_engine_set_dcn_mfma(self._h, int(on)), "pt_engine_set_dcn_mfma")      # 0 VALU blend, 1 dcn_mfma_kernel, 2 dcn_mfma2_kernel (64-output layers)

    def set_lstm_cluster(self, on: bool):
        """False: the streaming LSTM kernel (no co-residency requirement) -- whenever the recogniser shares the GPU with
        work on another stream; True (default): the weight-stationary cluster kernel."""
        L.check(self.lib.pt_engine_set_lstm_cluster(self._h, 1 if on else 0), "pt_engine_set_lstm_cluster")

    def check(self):
        """Raise PtError if the device flagged a failure in work already executed (pt_engine_check): call after the
        stream was synchronised, before consuming recognition results."""
        L.check(self.lib.pt_engine_check(self._h), "pt_engine_check")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pt_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing -------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._tdev).cuda_stream)

    def _chk(self, t: torch.Tensor, dtype, name):
        if not (t.is_cuda and t.device.index == self.device and t.dtype == dtype and t.is_contiguous()):
            raise L.PtError(f"{name}: expected contiguous {dtype} tensor on cuda:{self.device}, got "
                            f"{t.dtype} on {t.device} (contiguous={t.is_contiguous()})")

    # ---- weights ----------------------------------------------------------------------------------
    def load_weights(self, kind: int, blob: bytes):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        L.check(self.lib.pt_weights_load(self._h, kind, C.cast(buf, C.c_void_p), len(blob)), "pt_weights_load")

    def load_weights_device(self, kind: int, blob_u8: torch.Tensor):
        self._chk(blob_u8, torch.uint8, "blob")
        L.check(self.lib.pt_weights_load_device(self._h, kind, _ptr(blob_u8), blob_u8.numel(), self._stream()),
                "pt_weights_load_device")

    # ---- detection --------------------------------------------------------------------------------
    def det_plan(self, h: int, w: int, flavour: int = L.PT_DET_PRE_DB_PP) -> Tuple[int, int]:
        nh, nw = C.c_int(), C.c_int()
        L.check(self.lib.pt_det_plan(h, w, flavour, C.byref(nh), C.byref(nw)), "pt_det_plan")
        return nh.value, nw.value

    def det_forward(self, pages: torch.Tensor, flavour: int = L.PT_DET_PRE_DB_PP, thresh: float = 0.3,
                    use_dilation: bool = False, want_bitmap: bool = True, out_prob=None, out_bitmap=None):
        """pages uint8 [n,h,w,3] RGB on the GPU -> (prob f32 [n,nh,nw], bitmap int32 [n,nh,nw/32] or None)."""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, c = pages.shape
        assert c == 3
        nh, nw = self.det_plan(h, w, flavour)
        prob = out_prob if out_prob is not None else torch.empty((n, nh, nw), dtype=torch.float32, device=self._tdev)
        bitmap = None
        if want_bitmap:
            bitmap = out_bitmap if out_bitmap is not None else torch.empty((n, nh, nw // 32), dtype=torch.int32,
                                                                             device=self._tdev)
        L.check(self.lib.pt_det_forward(self._h, _ptr(pages), n, h, w, flavour, float(thresh), int(use_dilation),
                                        _ptr(prob), _ptr(bitmap), self._stream()), "pt_det_forward")
        return prob, bitmap

    def det_preprocess(self, pages: torch.Tensor, flavour: int = L.PT_DET_PRE_DB_PP) -> torch.Tensor:
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        nh, nw = self.det_plan(h, w, flavour)
        out = torch.empty((n, nh, nw, 8 if self.split else 4), dtype=self.act_dtype,
                          device=self._tdev)
        L.check(self.lib.pt_det_preprocess(self._h, _ptr(pages), n, h, w, flavour, _ptr(out), self._stream()),
                "pt_det_preprocess")
        return out

    def det_forward_net(self, x: torch.Tensor, want_logits: bool = False):
        """x bf16 NHWC4 [n,H,W,4] (BF16X3 mode: [n,H,W,8] = hi rgb0 | lo rgb0) -> prob f32 [n,H,W] (and logits)."""
        self._chk(x, self.act_dtype, "x")
        n, H, W, c = x.shape
        assert c == (8 if self.split else 4)
        prob = torch.empty((n, H, W), dtype=torch.float32, device=self._tdev)
        logits = torch.empty((n, H, W), dtype=torch.float32, device=self._tdev) if want_logits else None
        L.check(self.lib.pt_det_forward_net(self._h, _ptr(x), n, H, W, _ptr(prob), _ptr(logits), self._stream()),
                "pt_det_forward_net")
        return (prob, logits) if want_logits else prob

    # ---- layout (PicoDet) ----------------------------------------------------------------------------------------
    def layout_plan(self, inp_h: int, inp_w: int):
        fh, fw = (C.c_int * 4)(), (C.c_int * 4)()
        L.check(self.lib.pt_layout_plan(inp_h, inp_w, fh, fw), "pt_layout_plan")
        return list(fh), list(fw)

    def layout_preprocess(self, pages: torch.Tensor, inp_h: int = 800, inp_w: int = 608) -> torch.Tensor:
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        out = torch.empty((n, inp_h, inp_w, 8 if self.split else 4), dtype=self.act_dtype,
                          device=self._tdev)
        L.check(self.lib.pt_layout_preprocess(self._h, _ptr(pages), n, h, w, inp_h, inp_w, _ptr(out), self._stream()),
                "pt_layout_preprocess")
        return out

    def layout_forward_net(self, x: torch.Tensor):
        """x bf16 NHWC4 [n,H,W,4|8] -> 4 head maps f32 [n, A_l, 40] (class logits, then box-distribution logits)"""
        self._chk(x, self.act_dtype, "x")
        n, H, W, _ = x.shape
        fh, fw = self.layout_plan(H, W)
        heads = [torch.empty((n, fh[l] * fw[l], L.PT_LAYOUT_HEAD_CS), dtype=torch.float32, device=self._tdev) for l in range(4)]
        L.check(self.lib.pt_layout_forward_net(self._h, _ptr(x), n, H, W, *[_ptr(t) for t in heads], self._stream()),
                "pt_layout_forward_net")
        return heads

    def layout_forward(self, pages: torch.Tensor, inp_h: int = 800, inp_w: int = 608, num_classes: int = 5,
                       thr_lo: float = 0.45, max_cands: int = 2048):
        """pages uint8 [n,h,w,3] -> (counts int32 [n] on the host, candidate records f32 [n, max_cands, 48] on the device)"""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        counts = torch.zeros((n,), dtype=torch.int32, device=self._tdev)
        cands = torch.empty((n, max_cands, L.PT_LAYOUT_CAND_FLOATS), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_layout_forward(self._h, _ptr(pages), n, h, w, inp_h, inp_w, num_classes, float(thr_lo), max_cands,
                                           _ptr(counts), _ptr(cands), self._stream()), "pt_layout_forward")
        return counts, cands

    def tsr_preprocess(self, pages: torch.Tensor, tables, inp_h: int = 1024, inp_w: int = 1024, bgr: bool = True):
        """pages uint8 [np,h,w,3] on the device, tables: numpy array of TSR_TABLE_DTYPE -> bf16 NHWC4 [n,inp_h,inp_w,4|8]."""
        self._chk(pages, torch.uint8, "pages")
        npg, ph, pw, _ = pages.shape
        n = len(tables)
        tb = _upload(tables.view(np.uint8).reshape(n, -1), self._tdev)
        out = torch.empty((n, inp_h, inp_w, 8 if self.split else 4), dtype=self.act_dtype,
                          device=self._tdev)
        L.check(self.lib.pt_tsr_preprocess(self._h, _ptr(pages), npg, ph, pw, _ptr(tb), n, inp_h, inp_w, int(bgr), _ptr(out),
                                           self._stream()), "pt_tsr_preprocess")
        return out

    def tsr_forward_net(self, x: torch.Tensor, wireless: bool = False):
        """Lore detector (DLA-34 + DCN, or the ResNet-18 'wireless' one): x bf16 NHWC4 [n,H,W,4] (BF16X3: 8 channels) ->
        dict of fp32 NHWC head maps at H/4 x W/4 ({'hm': [n,h,w,2], 'st': [.,8], 'wh': [.,8], 'ax': [.,256],
        'cr': [.,256], 'reg': [.,2]})."""
        self._chk(x, self.act_dtype, "x")
        n, H, W, c = x.shape
        assert c == (8 if self.split else 4)
        h, w = H // 4, W // 4
        bufs = {k: torch.empty((n, h, w, 256 if k in ("ax", "cr") else 8), dtype=torch.float32, device=self._tdev)
                for k in ("hm", "st", "wh", "ax", "cr", "reg")}
        fn = self.lib.pt_tsr_forward_net_wireless if wireless else self.lib.pt_tsr_forward_net
        L.check(fn(self._h, _ptr(x), n, H, W, _ptr(bufs["hm"]), _ptr(bufs["st"]), _ptr(bufs["wh"]), _ptr(bufs["ax"]),
                   _ptr(bufs["cr"]), _ptr(bufs["reg"]), self._stream()), "pt_tsr_forward_net")
        bufs["hm"] = bufs["hm"][..., :2]
        bufs["reg"] = bufs["reg"][..., :2]
        return bufs

    def tsr_decode(self, heads, wiz_rev: bool = True, vis_thresh: float = 0.2, sync: bool = True):
        """heads: dict of fp32 NHWC maps (hm/st/wh/reg with 8-channel stride or their [..., :k] views, ax/cr 256)
        -> (counts int32 [n] on the host -- or still on the device when sync=False --, dets f32 [n,3000,9],
        logi f32 [n,3000,256] on the device)."""
        def full(t, c):
            if t.shape[-1] != c or not t.is_contiguous():
                base = t._base if t._base is not None else t
                assert base.shape[-1] == c and base.is_contiguous(), "head map must come from tsr_forward_net"
                return base
            return t
        hm, st, wh, reg = (full(heads[k], 8) for k in ("hm", "st", "wh", "reg"))
        ax, cr = full(heads["ax"], 256), full(heads["cr"], 256)
        n, h, w, _ = ax.shape
        counts = torch.zeros((n,), dtype=torch.int32, device=self._tdev)
        dets = torch.empty((n, L.PT_TSR_MAX_CELLS, 9), dtype=torch.float32, device=self._tdev)
        logi = torch.empty((n, L.PT_TSR_MAX_CELLS, 256), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_tsr_decode(self._h, _ptr(hm), _ptr(st), _ptr(wh), _ptr(ax), _ptr(cr), _ptr(reg), n, h, w,
                                       int(wiz_rev), float(vis_thresh), _ptr(counts), _ptr(dets), _ptr(logi),
                                       self._stream()), "pt_tsr_decode")
        return (counts.cpu().numpy() if sync else counts), dets, logi

    def tsr_forward_decode(self, x: torch.Tensor, wiz_rev: bool = True, vis_thresh: float = 0.2, sync: bool = True, out=None):
        """DLA-34 forward + decode in one call (sparse ax / cr heads): x bf16 NHWC4 [n,H,W,4|8] -> the outputs of tsr_decode.
        out: (counts i32 [n] zeroed, dets f32 [n,3000,9], logi f32 [n,3000,256]) contiguous device tensors to write into
        (slices of a larger allocation: several micro-batches then feed ONE processor call)"""
        self._chk(x, self.act_dtype, "x")
        n, H, W, _ = x.shape
        if out is not None:
            counts, dets, logi = out
            assert counts.shape == (n,) and dets.shape == (n, L.PT_TSR_MAX_CELLS, 9) and logi.shape == (n, L.PT_TSR_MAX_CELLS, 256)
            assert counts.is_contiguous() and dets.is_contiguous() and logi.is_contiguous()
        else:
            counts = torch.zeros((n,), dtype=torch.int32, device=self._tdev)
            dets = torch.empty((n, L.PT_TSR_MAX_CELLS, 9), dtype=torch.float32, device=self._tdev)
            logi = torch.empty((n, L.PT_TSR_MAX_CELLS, 256), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_tsr_forward_decode(self._h, _ptr(x), n, H, W, int(wiz_rev), float(vis_thresh), _ptr(counts),
                                               _ptr(dets), _ptr(logi), self._stream()), "pt_tsr_forward_decode")
        return (counts.cpu().numpy() if sync else counts), dets, logi

    def tsr_process(self, logi: torch.Tensor, dets: torch.Tensor, counts, use_2dpe: bool = False):
        """logic features of pt_tsr_decode -> (logic_axis, stacked_axis) f32 [n,3000,4]; rows [0, counts[i]) valid."""
        import numpy as np
        self._chk(logi, torch.float32, "logi")
        n = logi.shape[0]
        hc = np.ascontiguousarray(counts, dtype=np.int32)
        assert hc.shape == (n,)
        logic = torch.zeros((n, L.PT_TSR_MAX_CELLS, 4), dtype=torch.float32, device=self._tdev)
        stacked = torch.zeros((n, L.PT_TSR_MAX_CELLS, 4), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_tsr_process(self._h, _ptr(logi), _ptr(dets), hc.ctypes.data, n, int(use_2dpe), _ptr(logic),
                                        _ptr(stacked), self._stream()), "pt_tsr_process")
        return logic, stacked

    def det_bitmap(self, prob: torch.Tensor, thresh: float, use_dilation: bool = False) -> torch.Tensor:
        self._chk(prob, torch.float32, "prob")
        n, H, W = prob.shape
        bm = torch.empty((n, H, W // 32), dtype=torch.int32, device=self._tdev)
        L.check(self.lib.pt_det_bitmap(self._h, _ptr(prob), n, H, W, float(thresh), int(use_dilation), _ptr(bm),
                                       self._stream()), "pt_det_bitmap")
        return bm

    def det_box_scores(self, prob: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
        """boxes f32 [nb, 9] = (page, x0,y0,...,x3,y3) -> scores f32 [nb]."""
        self._chk(prob, torch.float32, "prob")
        self._chk(boxes, torch.float32, "boxes")
        n, H, W = prob.shape
        nb = boxes.shape[0]
        scores = torch.empty((nb,), dtype=torch.float32, device=self._tdev)
        if nb:
            L.check(self.lib.pt_det_box_scores(self._h, _ptr(prob), n, H, W, _ptr(boxes), nb, _ptr(scores),
                                               self._stream()), "pt_det_box_scores")
        return scores

    # ---- image classification (PP-LCNet) ------------------------------------------------------------
    def _cls_batch(self, images: Sequence[np.ndarray]):
        """RGB uint8 images of any sizes -> (flat device bytes, device pt_cls_image records, max_h, max_w)"""
        desc = np.zeros(len(images), dtype=CLS_IMAGE_DTYPE)
        off = 0
        parts = []
        for k, im in enumerate(images):
            im = np.ascontiguousarray(im[:, :, :3], dtype=np.uint8)
            desc[k] = (off, im.shape[0], im.shape[1])
            off += im.size
            parts.append(im.reshape(-1))
        flat = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        return (_upload(flat, self._tdev), _upload(desc.view(np.uint8).reshape(-1), self._tdev),
                int(desc["h"].max()), int(desc["w"].max()))

    def cls_preprocess(self, images: Sequence[np.ndarray], out_hw) -> torch.Tensor:
        """-> bf16 NHWC4 [n, out_h, out_w, 4] (8 channels in BF16X3 mode)"""
        n = len(images)
        base, desc, mh, mw = self._cls_batch(images)
        ch = 8 if self.split else 4
        out = torch.empty((n, out_hw[0], out_hw[1], ch), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_cls_preprocess(self._h, _ptr(base), _ptr(desc), n, mh, mw, out_hw[0], out_hw[1], _ptr(out),
                                           self._stream()), "pt_cls_preprocess")
        return out

    def cls_forward_net(self, x: torch.Tensor, slot: int = 0, textline: bool = False) -> torch.Tensor:
        """x bf16 NHWC4 [n,H,W,4|8] -> logits f32 [n, class_num] (device)"""
        self._chk(x, self.act_dtype, "x")
        n, H, W, _ = x.shape
        logits = torch.empty((n, L.PT_CLS_MAX_CLASSES), dtype=torch.float32, device=self._tdev)
        nc = C.c_int(0)
        L.check(self.lib.pt_cls_forward_net(self._h, slot, _ptr(x), n, H, W, int(textline), _ptr(logits), C.byref(nc),
                                            self._stream()), "pt_cls_forward_net")
        return logits[:, :nc.value]

    def cls_forward(self, images: Sequence[np.ndarray], out_hw, slot: int = 0, textline: bool = False) -> torch.Tensor:
        """RGB uint8 images (any sizes, host) -> logits f32 [n, class_num] (device): resize + normalise + PP-LCNet"""
        n = len(images)
        base, desc, mh, mw = self._cls_batch(images)
        logits = torch.empty((n, L.PT_CLS_MAX_CLASSES), dtype=torch.float32, device=self._tdev)
        nc = C.c_int(0)
        L.check(self.lib.pt_cls_forward(self._h, slot, _ptr(base), _ptr(desc), n, mh, mw, out_hw[0], out_hw[1], int(textline),
                                        _ptr(logits), C.byref(nc), self._stream()), "pt_cls_forward")
        return logits[:, :nc.value]

    def cls_forward_pages(self, pages: torch.Tensor, out_hw, slot: int = 0, textline: bool = False) -> torch.Tensor:
        """pages uint8 [n,h,w,3] resident on the device -> logits f32 [n, class_num]"""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        desc = np.zeros(n, dtype=CLS_IMAGE_DTYPE)
        desc["offset"] = np.arange(n, dtype=np.int64) * (h * w * 3)
        desc["h"], desc["w"] = h, w
        d = _upload(desc.view(np.uint8).reshape(-1), self._tdev)
        logits = torch.empty((n, L.PT_CLS_MAX_CLASSES), dtype=torch.float32, device=self._tdev)
        nc = C.c_int(0)
        L.check(self.lib.pt_cls_forward(self._h, slot, _ptr(pages), _ptr(d), n, h, w, out_hw[0], out_hw[1], int(textline),
                                        _ptr(logits), C.byref(nc), self._stream()), "pt_cls_forward")
        return logits[:, :nc.value]

    def cls_forward_lines(self, pages: torch.Tensor, lines: np.ndarray, out_hw, slot: int = 0, textline: bool = True):
        """text lines (REC_LINE_DTYPE records) cut from resident pages -> logits f32 [n_lines, class_num]"""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        nl = len(lines)
        logits = torch.empty((nl, L.PT_CLS_MAX_CLASSES), dtype=torch.float32, device=self._tdev)
        nc = C.c_int(0)
        if nl:
            d, px = self._lines_to_device(lines)
            mh, mw = max(1, int(lines["crop_h"].max())), max(1, int(lines["crop_w"].max()))
            L.check(self.lib.pt_cls_forward_lines(self._h, slot, _ptr(pages), n, h, w, _ptr(d), px.ctypes.data_as(C.c_void_p), nl,
                                                  mh, mw, out_hw[0], out_hw[1], int(textline), _ptr(logits), C.byref(nc),
                                                  self._stream()), "pt_cls_forward_lines")
        return logits[:, :nc.value] if nl else logits[:, :0]

    # ---- recognition ------------------------------------------------------------------------------
    def _lines_to_device(self, lines: np.ndarray):
        """structured array (REC_LINE_DTYPE) -> (uint8 device tensor holding the pt_rec_line records, host crop px)."""
        lines = np.ascontiguousarray(lines)
        assert lines.dtype == REC_LINE_DTYPE
        px = (lines["crop_w"].astype(np.int64) * lines["crop_h"].astype(np.int64)).clip(min=0)
        d = _upload(lines.view(np.uint8).reshape(-1), self._tdev)
        return d, np.ascontiguousarray(px)

    def rec_forward(self, pages: torch.Tensor, lines: np.ndarray, want_maxlogit: bool = True):
        """pages uint8 [n,h,w,3] on the GPU, lines: REC_LINE_DTYPE records -> (ids int32 [L,160], maxlogit f32 [L,160])."""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        nl = len(lines)
        ids = torch.empty((nl, L.PT_REC_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((nl, L.PT_REC_T), dtype=torch.float32, device=self._tdev) if want_maxlogit else None
        if nl:
            d, px = self._lines_to_device(lines)
            L.check(self.lib.pt_rec_forward(self._h, _ptr(pages), n, h, w, _ptr(d), px.ctypes.data_as(C.c_void_p), nl,
                                            _ptr(ids), _ptr(mx), self._stream()), "pt_rec_forward")
        return ids, mx

    def rec_forward_crops(self, crops: Sequence[np.ndarray]):
        """already-cropped RGB uint8 images (any sizes) -> (ids int32 [L,160], maxlogit f32 [L,160])."""
        nl = len(crops)
        lines = np.zeros(nl, dtype=REC_LINE_DTYPE)
        lines["crop_w"] = [c.shape[1] for c in crops]
        lines["crop_h"] = [c.shape[0] for c in crops]
        flat = np.concatenate([np.ascontiguousarray(c[:, :, :3], dtype=np.uint8).reshape(-1) for c in crops]) if nl else \
            np.zeros(0, np.uint8)
        ids = torch.empty((nl, L.PT_REC_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((nl, L.PT_REC_T), dtype=torch.float32, device=self._tdev)
        if nl:
            d, px = self._lines_to_device(lines)
            dc = torch.from_numpy(flat).to(self._tdev)
            L.check(self.lib.pt_rec_forward_crops(self._h, _ptr(dc), _ptr(d), px.ctypes.data_as(C.c_void_p), nl, _ptr(ids),
                                                  _ptr(mx), self._stream()), "pt_rec_forward_crops")
        return ids, mx

    def rec_pp_preprocess(self, pages: Optional[torch.Tensor], lines: np.ndarray, items: np.ndarray, total_floats: int,
                          img_h: int = 48, crops_flat: Optional[np.ndarray] = None) -> torch.Tensor:
        """PPOcrRecPreProcessor on the device: ``items`` (REC_PP_ITEM_DTYPE, the host's width-sorted plan) -> one flat fp32
        buffer holding every mini-batch's [b, 3, img_h, img_w] array.  Lines are cut from resident ``pages`` or, with
        ``crops_flat`` (concatenated uint8 crops), are already cropped."""
        assert items.dtype == REC_PP_ITEM_DTYPE and len(items) and len(lines)
        out = torch.empty((int(total_floats),), dtype=torch.float32, device=self._tdev)
        d, px = self._lines_to_device(lines)
        di = _upload(np.ascontiguousarray(items).view(np.uint8).reshape(-1), self._tdev)
        max_w = int(items["img_w"].max())
        if crops_flat is None:
            self._chk(pages, torch.uint8, "pages")
            n, h, w, _ = pages.shape
            L.check(self.lib.pt_rec_pp_preprocess(self._h, _ptr(pages), n, h, w, _ptr(d), px.ctypes.data_as(C.c_void_p), len(lines),
                                                  _ptr(di), len(items), img_h, max_w, _ptr(out), self._stream()),
                    "pt_rec_pp_preprocess")
        else:
            dc = torch.from_numpy(np.ascontiguousarray(crops_flat)).to(self._tdev)
            L.check(self.lib.pt_rec_pp_preprocess_crops(self._h, _ptr(dc), _ptr(d), px.ctypes.data_as(C.c_void_p), len(lines),
                                                        _ptr(di), len(items), img_h, max_w, _ptr(out), self._stream()),
                    "pt_rec_pp_preprocess_crops")
        return out

    def rec_preprocess(self, pages: torch.Tensor, lines: np.ndarray) -> torch.Tensor:
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        nl = len(lines)
        shape = (nl, L.PT_REC_H, L.PT_REC_W, 2) if self.split else (nl, L.PT_REC_H, L.PT_REC_W)
        gray = torch.empty(shape, dtype=self.act_dtype, device=self._tdev)
        d, px = self._lines_to_device(lines)
        L.check(self.lib.pt_rec_preprocess(self._h, _ptr(pages), n, h, w, _ptr(d), px.ctypes.data_as(C.c_void_p), nl,
                                           _ptr(gray), self._stream()), "pt_rec_preprocess")
        return gray

    def rec_forward_net(self, gray: torch.Tensor):
        """gray bf16 [n,32,640] (BF16X3: [n,32,640,2]) -> (ids int32 [n,160], maxlogit f32 [n,160])."""
        self._chk(gray, self.act_dtype, "gray")
        n = gray.shape[0]
        ids = torch.empty((n, L.PT_REC_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((n, L.PT_REC_T), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_rec_forward_net(self._h, _ptr(gray), n, _ptr(ids), _ptr(mx), self._stream()),
                "pt_rec_forward_net")
        return ids, mx

    # ---- ConvNextViT recogniser ---------------------------------------------------------------
    def _crops_to_device(self, crops: Sequence[np.ndarray]):
        nl = len(crops)
        lines = np.zeros(nl, dtype=REC_LINE_DTYPE)
        lines["crop_w"] = [c.shape[1] for c in crops]
        lines["crop_h"] = [c.shape[0] for c in crops]
        flat = np.concatenate([np.ascontiguousarray(c[:, :, :3], dtype=np.uint8).reshape(-1) for c in crops])
        d, px = self._lines_to_device(lines)
        return d, px, torch.from_numpy(flat).to(self._tdev), self._crop_wh(lines)

    @staticmethod
    def _crop_wh(lines: np.ndarray) -> np.ndarray:
        """host int32 [n, 2] = (crop_w, crop_h): lets the engine share the all-padding chunks of short lines"""
        return np.ascontiguousarray(np.stack([lines["crop_w"], lines["crop_h"]], 1).astype(np.int32))

    def rec_cvit_forward(self, pages: torch.Tensor, lines: np.ndarray):
        """pages uint8 [n,h,w,3] on the GPU, lines: REC_LINE_DTYPE records -> (ids int32 [L,201], maxlogit f32 [L,201])."""
        self._chk(pages, torch.uint8, "pages")
        n, h, w, _ = pages.shape
        nl = len(lines)
        ids = torch.empty((nl, L.PT_CVIT_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((nl, L.PT_CVIT_T), dtype=torch.float32, device=self._tdev)
        if nl:
            d, px = self._lines_to_device(lines)
            wh = self._crop_wh(lines)
            L.check(self.lib.pt_rec_cvit_forward(self._h, _ptr(pages), n, h, w, _ptr(d), px.ctypes.data_as(C.c_void_p),
                                                 wh.ctypes.data_as(C.c_void_p), nl, _ptr(ids), _ptr(mx), self._stream()),
                    "pt_rec_cvit_forward")
        return ids, mx

    def rec_cvit_forward_crops(self, crops: Sequence[np.ndarray]):
        """already-cropped RGB uint8 images (any sizes) -> (ids int32 [L,201], maxlogit f32 [L,201])."""
        nl = len(crops)
        ids = torch.empty((nl, L.PT_CVIT_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((nl, L.PT_CVIT_T), dtype=torch.float32, device=self._tdev)
        if nl:
            d, px, dc, wh = self._crops_to_device(crops)
            L.check(self.lib.pt_rec_cvit_forward_crops(self._h, _ptr(dc), _ptr(d), px.ctypes.data_as(C.c_void_p),
                                                       wh.ctypes.data_as(C.c_void_p), nl, _ptr(ids), _ptr(mx), self._stream()),
                    "pt_rec_cvit_forward_crops")
        return ids, mx

    def rec_cvit_preprocess_crops(self, crops: Sequence[np.ndarray]) -> torch.Tensor:
        """keep-ratio resize to 32 x 804 + gray of already-cropped lines -> fp32 [L,32,804]."""
        nl = len(crops)
        gray = torch.empty((nl, L.PT_REC_H, L.PT_CVIT_W), dtype=torch.float32, device=self._tdev)
        d, px, dc, _ = self._crops_to_device(crops)
        L.check(self.lib.pt_rec_cvit_preprocess_crops(self._h, _ptr(dc), _ptr(d), px.ctypes.data_as(C.c_void_p), nl, _ptr(gray),
                                                      self._stream()), "pt_rec_cvit_preprocess_crops")
        return gray

    def rec_cvit_forward_net(self, gray: torch.Tensor, text_w: Optional[Sequence[int]] = None):
        """gray fp32 [3n,32,300] (chunks) or [n,32,804] (lines) -> (ids int32 [n,201], maxlogit f32 [n,201]).
        ``text_w`` (per line): columns >= text_w[i] are zero padding -- all-padding chunks are then computed once and shared."""
        self._chk(gray, torch.float32, "gray")
        assert gray.dim() == 3 and gray.shape[1] == L.PT_REC_H and gray.shape[2] in (L.PT_CVIT_CHUNK_W, L.PT_CVIT_W)
        layout = 1 if gray.shape[2] == L.PT_CVIT_W else 0
        assert layout == 1 or gray.shape[0] % 3 == 0, "three chunks per line"
        n = gray.shape[0] if layout else gray.shape[0] // 3
        ids = torch.empty((n, L.PT_CVIT_T), dtype=torch.int32, device=self._tdev)
        mx = torch.empty((n, L.PT_CVIT_T), dtype=torch.float32, device=self._tdev)
        tw = None if text_w is None else np.ascontiguousarray(np.asarray(text_w, dtype=np.int32))
        assert tw is None or len(tw) == n
        L.check(self.lib.pt_rec_cvit_forward_net(self._h, _ptr(gray), layout, n, None if tw is None else tw.ctypes.data_as(C.c_void_p),
                                                 _ptr(ids), _ptr(mx), self._stream()), "pt_rec_cvit_forward_net")
        return ids, mx

    def mtl_backbone_forward(self, x: torch.Tensor) -> torch.Tensor:
        """MtlTabNet backbone (TableResNetExtra): x fp32 [n, 3, H, W] (normalised image, on the GPU) -> the last feature map
        fp32 [n, 512, H/8, W/8].  torch only moves data here (NCHW fp32 -> padded NHWC bf16, hi/lo in BF16X3)."""
        self._chk(x, torch.float32, "x")
        n, c, h, w = x.shape
        assert c == 3 and h % 8 == 0 and w % 8 == 0
        xp = torch.zeros((n, h, w, 32), dtype=torch.float32, device=self._tdev)
        xp[..., :3] = x.permute(0, 2, 3, 1)
        hi = xp.to(self.act_dtype)
        if self.split:
            hi = torch.cat([hi, (xp - hi.float()).to(self.act_dtype)], -1)
        hi = hi.contiguous()
        f3 = torch.empty((n, h // 8, w // 8, 512), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_tsr_mtl_backbone_net(self._h, _ptr(hi), n, h, w, _ptr(f3), self._stream()), "pt_tsr_mtl_backbone_net")
        return f3.permute(0, 3, 1, 2)

    def mtl_backbone_features(self, xin: torch.Tensor, size_hw: Tuple[int, int]) -> torch.Tensor:
        """pre-processed input bf16 [n, H, W, 32 (x 2 in BF16X3)] (mtl_preprocess) -> f3 fp32 [n, H/8 * W/8, 512] (NHWC, flattened)"""
        n = xin.shape[0]
        h, w = size_hw
        f3 = torch.empty((n, (h // 8) * (w // 8), 512), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_tsr_mtl_backbone_net(self._h, _ptr(xin), n, h, w, _ptr(f3), self._stream()), "pt_tsr_mtl_backbone_net")
        return f3

    def mtl_preprocess(self, pages: torch.Tensor, tables: np.ndarray, size: int = 480) -> torch.Tensor:
        """MtlTabNet test pipeline (TableResize keep-ratio long side `size`, TablePad, to_tensor, normalise) of table crops cut from
        resident pages.  pages uint8 [n_pages, h, w, 3]; tables: TSR_TABLE_DTYPE records (page, x0, y0, crop_w, crop_h).
        -> bf16 [n, size, size, 32] ([hi 32 | lo 32] in BF16X3), the backbone's input."""
        self._chk(pages, torch.uint8, "pages")
        assert tables.dtype == TSR_TABLE_DTYPE
        n = len(tables)
        d_tab = _upload(tables.view(np.uint8).reshape(-1), self._tdev)
        m = 2 if self.split else 1
        out = torch.empty((n, size, size, 32 * m), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_tsr_mtl_preprocess(self._h, _ptr(pages), pages.shape[0], pages.shape[1], pages.shape[2], _ptr(d_tab), n, size,
                                               _ptr(out), self._stream()), "pt_tsr_mtl_preprocess")
        return out

    def mtl_resized_size(self, crop_w: int, crop_h: int, size: int = 480) -> Tuple[int, int]:
        ow, oh = C.c_int(), C.c_int()
        self.lib.pt_tsr_mtl_resized_size(int(crop_w), int(crop_h), int(size), C.byref(ow), C.byref(oh))
        return ow.value, oh.value

    def mtl_decoder_config(self) -> dict:
        v = (C.c_int * 13)()
        L.check(self.lib.pt_tsr_mtl_decoder_config(self._h, v), "pt_tsr_mtl_decoder_config")
        k = ("num_classes", "num_classes_cell", "sos", "eos", "pad", "max_len", "sos_cell", "eos_cell", "pad_cell", "max_len_cell", "tag0", "tag1", "d_ff_pad")
        return dict(zip(k, [int(x) for x in v]))

    def mtl_decode(self, f3: torch.Tensor, want_cell_logits: bool = False, force_redecode: bool = False) -> dict:
        """The three MtlTabNet decoders for a batch of tables (pt_tsr_mtl_structure + pt_tsr_mtl_cells).  f3 fp32 [n, hw, 512].
        -> dict: tag_logits fp32 [n, T, classes], boxes fp32 [n, T, 4] (device), lens [n], cell_counts [n], cell_steps [n] (numpy),
        cell_ids int32 / cell_prob fp32 [total cells, Tc] (device; cells ordered by table, then position), cell_logits or None."""
        self._chk(f3, torch.float32, "f3")
        n, hw, d = f3.shape
        assert d == 512
        cfg = self.mtl_decoder_config()
        T, Tc = cfg["max_len"] + 1, cfg["max_len_cell"] + 1
        tag = torch.zeros((n, T, cfg["num_classes"]), dtype=torch.float32, device=self._tdev)
        box = torch.zeros((n, T, 4), dtype=torch.float32, device=self._tdev)
        lens = np.zeros(n, np.int32)
        counts = np.zeros(n, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        L.check(self.lib.pt_tsr_mtl_structure(self._h, _ptr(f3), n, hw, _ptr(tag), _ptr(box), ip(lens), ip(counts), 1 if force_redecode else 0,
                                              self._stream()), "pt_tsr_mtl_structure")
        total = int(counts.sum())
        steps = np.zeros(n, np.int32)
        ids = torch.zeros((max(total, 1), Tc), dtype=torch.int32, device=self._tdev)
        prob = torch.zeros((max(total, 1), Tc), dtype=torch.float32, device=self._tdev)
        logits = torch.zeros((max(total, 1), Tc, cfg["num_classes_cell"]), dtype=torch.float32, device=self._tdev) if want_cell_logits else None
        L.check(self.lib.pt_tsr_mtl_cells(self._h, total, _ptr(ids), _ptr(prob), _ptr(logits), ip(steps), 1 if force_redecode else 0, self._stream()),
                "pt_tsr_mtl_cells")
        return dict(tag_logits=tag, boxes=box, lens=lens, cell_counts=counts, cell_steps=steps, cell_ids=ids[:total], cell_prob=prob[:total],
                    cell_logits=None if logits is None else logits[:total], cfg=cfg)

    def op_conv2d(self, x: torch.Tensor, w_tiled: torch.Tensor, bias: torch.Tensor, ks: int, stride: int = 1,
                  relu: bool = False, res: Optional[torch.Tensor] = None, res_mode: int = 0, rep: int = 1,
                  shuffle_cout: int = 0, out: Optional[torch.Tensor] = None, out_coff: int = 0,
                  split: int = 0) -> torch.Tensor:
        """Single conv on the MFMA kernel.  x bf16 [B,H,W,Cin]; w_tiled int16/bf16 bits; bias f32 [N].  split: 0 plain bf16, 1 (hi | lo)
        tensors with the three-pass tiles, 2 the same tensors with the two-pass fp16 tiles (weights.tile_conv_weight_f16x2)."""
        self._chk(x, self.act_dtype, "x")
        self._chk(bias, torch.float32, "bias")
        B, H, W, Cin = x.shape
        m = 2 if split else 1
        Cin //= m
        N = bias.numel()
        pad = ks // 2
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        if out is None:
            if shuffle_cout:
                out = torch.empty((B, 2 * Ho, 2 * Wo, shuffle_cout * m), dtype=self.act_dtype, device=self._tdev)
            else:
                out = torch.empty((B, Ho * rep, Wo * rep, N * m), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_conv2d(self._h, _ptr(x), B, H, W, Cin, _ptr(w_tiled), _ptr(bias), N, ks, stride,
                                      _ptr(out), out.shape[-1], out_coff, rep, shuffle_cout, _ptr(res), res_mode,
                                      int(relu), int(split), out.shape[-1] // 2, self._stream()), "pt_op_conv2d")      # relu: bool, or the epilogue's activation code (0 none, 1 ReLU, 2 hardswish)
        return out

    def op_dcn(self, x: torch.Tensor, om: torch.Tensor, w_tiled: torch.Tensor, bias: torch.Tensor, relu: bool = True, split: int = 0) -> torch.Tensor:
        """Fused modulated deformable 3x3 convolution (lore/dcnv2.py:71-86) as a single operator.  x bf16 [B,H,W,C] ([hi | lo] when split),
        om fp32 [B,H,W,32] (18 offsets, 9 mask logits, 5 unused), w_tiled = the [N, 9C, 1, 1] weight tiled as a 1x1 conv, bias fp32 [N]."""
        self._chk(x, self.act_dtype, "x")
        self._chk(om, torch.float32, "om")
        self._chk(bias, torch.float32, "bias")
        B, H, W, Cc = x.shape
        m = 2 if split else 1
        Cc //= m
        if tuple(om.shape) != (B, H, W, 32):
            raise ValueError(f"op_dcn: om must be [B,H,W,32], got {tuple(om.shape)}")
        N = bias.numel()
        out = torch.empty((B, H, W, N * m), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_dcn(self._h, _ptr(x), _ptr(om), B, H, W, Cc, _ptr(w_tiled), _ptr(bias), N, _ptr(out), int(relu), int(split),
                                   self._stream()), "pt_op_dcn")
        return out

    # ---- single operators of the generic ONNX executor (pdf_table_amd/onnx_exec.py); bf16 NHWC, C a multiple of 8 -------
    # (split=True: the tolerance mode's (hi | lo) tensors -- the last dimension holds [hi(C) | lo(C)], C = shape[-1] // 2)
    def op_dwconv(self, x: torch.Tensor, w_taps: torch.Tensor, bias: torch.Tensor, k: int, stride: int = 1, act: int = 0, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        self._chk(w_taps, torch.float32, "w_taps")
        self._chk(bias, torch.float32, "bias")
        B, H, W, Cc = x.shape
        pad = k // 2
        out = torch.empty((B, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_dwconv(self._h, _ptr(x), B, H, W, Cc // 2 if split else Cc, _ptr(w_taps), _ptr(bias), k, stride, act, _ptr(out), int(split),
                                      self._stream()), "pt_op_dwconv")
        return out

    def op_add(self, a: torch.Tensor, b: torch.Tensor, split: bool = False) -> torch.Tensor:
        self._chk(a, self.act_dtype, "a")
        self._chk(b, self.act_dtype, "b")
        if a.shape != b.shape:
            raise ValueError(f"op_add: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        L.check(self.lib.pt_op_add(self._h, _ptr(a), _ptr(b), _ptr(out), a.numel() // a.shape[-1], a.shape[-1] // 2 if split else a.shape[-1], int(split),
                                   self._stream()), "pt_op_add")
        return out

    def op_maxpool(self, x: torch.Tensor, k: int, stride: int, pad: int, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        B, H, W, Cc = x.shape
        out = torch.empty((B, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_maxpool(self._h, _ptr(x), B, H, W, Cc // 2 if split else Cc, k, stride, pad, _ptr(out), int(split), self._stream()), "pt_op_maxpool")
        return out

    def op_avgpool(self, x: torch.Tensor, k: int, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        B, H, W, Cc = x.shape
        out = torch.empty((B, H // k, W // k, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_avgpool(self._h, _ptr(x), B, H, W, Cc // 2 if split else Cc, k, _ptr(out), int(split), self._stream()), "pt_op_avgpool")
        return out

    def op_chan_mean(self, x: torch.Tensor, split: bool = False) -> torch.Tensor:
        """GlobalAveragePool: [B, H, W, C] -> [B, 1, 1, C]"""
        self._chk(x, self.act_dtype, "x")
        B, H, W, Cc = x.shape
        ch = Cc // 2 if split else Cc
        scratch = torch.empty((self.lib.pt_op_chan_mean_scratch_floats(B, ch),), dtype=torch.float32, device=self._tdev)
        out = torch.empty((B, 1, 1, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_chan_mean(self._h, _ptr(x), B, H * W, ch, _ptr(scratch), _ptr(out), int(split), self._stream()), "pt_op_chan_mean")
        return out

    def op_scale_channels(self, x: torch.Tensor, gate: torch.Tensor, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        self._chk(gate, self.act_dtype, "gate")
        B, H, W, Cc = x.shape
        if gate.numel() != B * Cc:
            raise ValueError(f"op_scale_channels: gate {tuple(gate.shape)} does not match [{B}, {Cc}]")
        out = torch.empty_like(x)
        L.check(self.lib.pt_op_scale_channels(self._h, _ptr(x), _ptr(gate), B, H * W, Cc // 2 if split else Cc, _ptr(out), int(split), self._stream()),
                "pt_op_scale_channels")
        return out

    def op_act(self, x: torch.Tensor, kind: int, alpha: float = 0.0, beta: float = 0.0, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        out = torch.empty_like(x)
        ch = x.shape[-1] // 2 if split else x.shape[-1]
        L.check(self.lib.pt_op_act(self._h, _ptr(x), x.numel() // (2 if split else 1), kind, float(alpha), float(beta), _ptr(out), ch, int(split),
                                   self._stream()), "pt_op_act")
        return out

    def op_copy_channels(self, src: torch.Tensor, dst: torch.Tensor, n: int, src_coff: int = 0, dst_coff: int = 0):
        """dst[..., dst_coff : dst_coff + n] = src[..., src_coff : src_coff + n] (same pixel count; Concat / Slice / Split over channels)"""
        self._chk(src, self.act_dtype, "src")
        self._chk(dst, self.act_dtype, "dst")
        npix = src.numel() // src.shape[-1]
        assert npix == dst.numel() // dst.shape[-1]
        L.check(self.lib.pt_op_copy_channels(self._h, _ptr(src), npix, src.shape[-1], src_coff, _ptr(dst), dst.shape[-1], dst_coff, n, self._stream()),
                "pt_op_copy_channels")

    def copy_bytes(self, src: torch.Tensor, dst: torch.Tensor):
        """dst <- src (same byte count, contiguous) by a kernel on the current stream; either tensor may live in pinned host memory"""
        nb = src.numel() * src.element_size()
        assert nb == dst.numel() * dst.element_size() and src.is_contiguous() and dst.is_contiguous()
        L.check(self.lib.pt_copy_bytes(self._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), nb, self._stream()), "pt_copy_bytes")

    def op_upsample(self, x: torch.Tensor, f: int) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        B, H, W, Cc = x.shape
        out = torch.empty((B, H * f, W * f, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_upsample_nearest(self._h, _ptr(x), B, H, W, Cc, int(f), _ptr(out), self._stream()), "pt_op_upsample_nearest")
        return out

    def op_mul(self, a: torch.Tensor, b: torch.Tensor, split: bool = False) -> torch.Tensor:
        self._chk(a, self.act_dtype, "a")
        self._chk(b, self.act_dtype, "b")
        if a.shape != b.shape:
            raise ValueError(f"op_mul: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        L.check(self.lib.pt_op_mul(self._h, _ptr(a), _ptr(b), _ptr(out), a.numel() // (2 if split else 1), a.shape[-1] // 2 if split else a.shape[-1], int(split),
                                   self._stream()), "pt_op_mul")
        return out

    def op_layernorm(self, x: torch.Tensor, c: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        self._chk(gamma, torch.float32, "gamma")
        self._chk(beta, torch.float32, "beta")
        out = torch.empty_like(x)
        cp = x.shape[-1] // 2 if split else x.shape[-1]
        L.check(self.lib.pt_op_layernorm(self._h, _ptr(x), x.numel() // x.shape[-1], cp, int(c), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), int(split),
                                         self._stream()), "pt_op_layernorm")
        return out

    def op_softmax(self, x: torch.Tensor, c: int, f32: bool = False, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        rows = x.numel() // x.shape[-1]
        cp = x.shape[-1] // 2 if split else x.shape[-1]
        if f32:
            out = torch.empty(x.shape[:-1] + (int(c),), dtype=torch.float32, device=self._tdev)
            L.check(self.lib.pt_op_softmax(self._h, _ptr(x), rows, cp, int(c), _ptr(out), None, int(split), self._stream()), "pt_op_softmax")
        else:
            out = torch.empty_like(x)
            L.check(self.lib.pt_op_softmax(self._h, _ptr(x), rows, cp, int(c), None, _ptr(out), int(split), self._stream()), "pt_op_softmax")
        return out

    def op_attention(self, qkv: torch.Tensor, heads: int, d: int, scale: float, out_c: int, split: bool = False) -> torch.Tensor:
        """qkv bf16 [B, 1, T, >= 3 heads d] rows of [q | k | v] -> bf16 [B, 1, T, out_c] (channels heads * d .. out_c are zero); split: both tensors
        carry [hi | lo] halves of that width"""
        self._chk(qkv, self.act_dtype, "qkv")
        B, T = qkv.shape[0], qkv.shape[-2]
        m = 2 if split else 1
        out = torch.zeros((B, 1, T, int(out_c) * m), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_attention(self._h, _ptr(qkv), B, T, int(heads), int(d), qkv.shape[-1] // m, float(scale), _ptr(out), int(out_c), int(split),
                                         self._stream()), "pt_op_attention")
        return out

    def op_stem7x7(self, x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        B, H, W, _ = x.shape
        out = torch.empty((B, H // 2, W // 2, 128 if split else 64), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_stem7x7(self._h, _ptr(x), B, H, W, _ptr(w), _ptr(bias), _ptr(out), int(split), self._stream()),
                "pt_op_stem7x7")
        return out

    def op_maxpool3x3s2(self, x: torch.Tensor, split: bool = False) -> torch.Tensor:
        self._chk(x, self.act_dtype, "x")
        B, H, W, Cc = x.shape
        out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=self.act_dtype, device=self._tdev)
        L.check(self.lib.pt_op_maxpool3x3s2(self._h, _ptr(x), B, H, W, Cc // 2 if split else Cc, _ptr(out), int(split),
                                            self._stream()),
                "pt_op_maxpool3x3s2")
        return out

    def op_db_head_final(self, x: torch.Tensor, w4x64: torch.Tensor, bias: torch.Tensor, split: bool = False):
        self._chk(x, self.act_dtype, "x")
        B, H, W, _ = x.shape
        prob = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device=self._tdev)
        logits = torch.empty((B, 2 * H, 2 * W), dtype=torch.float32, device=self._tdev)
        L.check(self.lib.pt_op_db_head_final(self._h, _ptr(x), B, H, W, _ptr(w4x64), _ptr(bias), _ptr(prob),
                                             _ptr(logits), int(split), self._stream()), "pt_op_db_head_final")
        return prob, logits

    # ---- profiling ---------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """True / 1: HIP events around every launch; 2 + L.PT_PROF_CLASSES.index(name): only that kernel class; False: off"""
        L.check(self.lib.pt_profile_enable(self._h, int(on)), "pt_profile_enable")

    def profile_read_labels(self):
        """-> {label: {"launches", "ms", "flop", "bytes"}} of the launches recorded since the last read (profile_enable(1))"""
        buf = C.create_string_buffer(1 << 18)
        L.check(self.lib.pt_profile_read_labels(self._h, buf, len(buf)), "pt_profile_read_labels")
        out = {}
        for ln in buf.value.decode().splitlines():
            lab, n, ms, fl, by = ln.split("\t")
            out[lab] = {"launches": int(n), "ms": float(ms), "flop": float(fl), "bytes": float(by)}
        return out

    def profile_read(self):
        ms = (C.c_double * 4)()
        nl = (C.c_longlong * 4)()
        fl = (C.c_double * 4)()
        L.check(self.lib.pt_profile_read(self._h, ms, nl, fl), "pt_profile_read")
        return {k: {"ms": ms[i], "launches": nl[i], "flop": fl[i]} for i, k in enumerate(L.PT_PROF_CLASSES)}



def _serialised(fn):
    @functools.wraps(fn)
    def call(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)
    return call


# every public method of the handle queues its pt_* calls under the engine's lock (see HipEngine.__init__); close() / __del__ stay free of it
for _name, _fn in list(vars(HipEngine).items()):
    if callable(_fn) and not _name.startswith("_") and _name not in ("close", "precision_scope") and not isinstance(_fn, (property, staticmethod, classmethod)):
        setattr(HipEngine, _name, _serialised(_fn))


# ---- host-side halves of the DB post-process (no GPU needed) -------------------------------------------
def db_candidates(bitmap_words: np.ndarray, max_candidates: int = 1000, min_size: float = 3.0):
    """bitmap uint32/int32 [H, W/32] of one page -> (boxes f32 [k,8] TL,TR,BR,BL ; sside f32 [k])."""
    lib = L.load()
    bm = np.ascontiguousarray(bitmap_words).view(np.uint32)
    H, wpr = bm.shape
    cap = max_candidates
    boxes = np.empty((cap, 8), dtype=np.float32)
    sside = np.empty((cap,), dtype=np.float32)
    n = C.c_int()
    L.check(lib.pt_db_candidates(bm.ctypes.data_as(C.c_void_p), H, wpr * 32, max_candidates, float(min_size),
                                 boxes.ctypes.data_as(C.c_void_p), sside.ctypes.data_as(C.c_void_p), cap,
                                 C.byref(n)), "pt_db_candidates")
    return boxes[:n.value].copy(), sside[:n.value].copy()


def db_candidates_batch(bitmaps: np.ndarray, max_candidates: int = 1000, min_size: float = 3.0, n_threads: int = 0):
    """bitmaps uint32/int32 [n, H, W/32] -> (boxes f32 [n, cap, 8], counts int32 [n]); pages run on threads inside the library"""
    lib = L.load()
    bm = np.ascontiguousarray(bitmaps).view(np.uint32)
    n, H, wpr = bm.shape
    cap = max_candidates
    boxes = np.empty((n, cap, 8), dtype=np.float32)
    counts = np.zeros((n,), dtype=np.int32)
    L.check(lib.pt_db_candidates_batch(bm.ctypes.data_as(C.c_void_p), n, H, wpr * 32, max_candidates, float(min_size), int(n_threads),
                                       boxes.ctypes.data_as(C.c_void_p), None, cap, counts.ctypes.data_as(C.c_void_p)),
            "pt_db_candidates_batch")
    return boxes, counts


def db_finalize_batch(boxes: np.ndarray, scores: np.ndarray, counts: np.ndarray, net_hw, dest_hw, box_thresh: float = 0.6,
                      unclip_ratio: float = 1.5, min_size: float = 3.0, post_flavour: int = L.PT_DET_POST_DB_PP,
                      filter_tag: bool = False, n_threads: int = 0):
    """boxes f32 [n, cap, 8], scores f32 [n, cap], counts int32 [n] -> per page: int32 [k, 8] boxes, or, with filter_tag
    (the db_pp flavour's filter_tag_det_res), float32 [k, 8]"""
    lib = L.load()
    n, cap, _ = boxes.shape
    out = np.empty((n, cap, 8), dtype=np.int32)
    outf = np.empty((n, cap, 8), dtype=np.float32) if filter_tag else None
    nout = np.zeros((n,), dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    L.check(lib.pt_db_finalize_batch(boxes.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p),
                                     counts.ctypes.data_as(C.c_void_p), n, cap, float(box_thresh), float(unclip_ratio), float(min_size),
                                     int(net_hw[0]), int(net_hw[1]), int(dest_hw[0]), int(dest_hw[1]), int(post_flavour),
                                     1 if filter_tag else 0, int(n_threads), out.ctypes.data_as(C.c_void_p),
                                     outf.ctypes.data_as(C.c_void_p) if filter_tag else None, None, nout.ctypes.data_as(C.c_void_p)),
            "pt_db_finalize_batch")
    src = outf if filter_tag else out
    return [src[i, :nout[i]].copy() for i in range(n)]


def db_finalize(boxes: np.ndarray, scores: np.ndarray, net_hw, dest_hw, box_thresh: float = 0.6,
                unclip_ratio: float = 1.5, min_size: float = 3.0, post_flavour: int = L.PT_DET_POST_DB_PP):
    lib = L.load()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 8)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    nb = boxes.shape[0]
    out = np.empty((max(nb, 1), 8), dtype=np.int32)
    osc = np.empty((max(nb, 1),), dtype=np.float32)
    n = C.c_int()
    L.check(lib.pt_db_finalize(boxes.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p), nb,
                               float(box_thresh), float(unclip_ratio), float(min_size), int(net_hw[0]), int(net_hw[1]),
                               int(dest_hw[0]), int(dest_hw[1]), int(post_flavour), out.ctypes.data_as(C.c_void_p),
                               osc.ctypes.data_as(C.c_void_p), max(nb, 1), C.byref(n)), "pt_db_finalize")
    return out[:n.value].copy(), osc[:n.value].copy()
