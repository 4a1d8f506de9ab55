"""Generic executor of the ONNX importer's engine layer list on the HIP engine.

The reference runs its default detection / layout / classification models as ONNX graphs through onnxruntime
(``DeployUtils.prepare_onnx_model`` utils/deploy_utils.py:243-280, ``BaseInferTask.infer`` model/ocr_pdf/base_infer_task.py:
366-370: ``predictor.run(None, input_dict)``).  ``pdf_table_amd.onnx_import.recognise`` maps a graph whose ARCHITECTURE the
engine has a dedicated launch graph for (DB-ResNet18, CRNN, PP-LCNet) onto that graph.  This module executes any other
convolutional graph layer by layer:

  * ``OnnxGraph.layers()`` (BatchNorm folded, activations attached) is walked once at load time: convolution weights are
    padded and tiled for the MFMA implicit-GEMM kernel, depthwise weights put tap-major, everything uploaded;
  * ``run()`` issues ONE engine call per layer through the C ABI -- ``pt_op_conv2d`` (1x1 / 3x3, stride 1 / 2, ReLU /
    hardswish / residual-add epilogue, 2x2 transposed convs as pixel-shuffle GEMMs), ``pt_op_dwconv``, ``pt_op_maxpool``,
    ``pt_op_chan_mean``, ``pt_op_scale_channels``, ``pt_op_add``, ``pt_op_act``;
  * activations are bf16 NHWC tensors whose channel count is padded to a multiple of 64 with zeros (what the GEMM tiles
    want; zero weights keep the padding zero); PyTorch only owns the device memory and does the data MOVEMENT between
    layers that has no arithmetic in it (NCHW <-> NHWC, channel concat, nearest-neighbour up-sampling, the final cast).

Sequence models (SVTR-type recognisers such as PP-OCRv4 rec: the recogniser ``fix_model_names()`` selects,
model/ocr_pdf/configuration_ocr_document.py:138-141): token tensors [B, T, C] live as [B, 1, T, Cpad]; ``MatMul`` + bias with a constant
weight is the 1x1 GEMM, LayerNormalization (native or decomposed), GELU / swish, Softmax and the fused-qkv multi-head attention run on
``pt_op_layernorm`` / ``pt_op_act`` / ``pt_op_softmax`` / ``pt_op_attention``.  The shape plumbing between them (Shape / Slice / Concat /
Reshape / Transpose / Gather / Split / Flatten) does not move data: it is evaluated on the host over INDEX arrays (``_View``: for every
element of the logical tensor, which element of a materialised tensor it is), and a consumer accepts a view only if its index array is
exactly one of the layouts its kernel reads -- NCHW map <-> token rows, a channel slice, the (B, heads, T, d) split of fused q / k / v rows.
Anything else raises.

There is no CPU path: an operator outside this set raises ``UnsupportedOnnxGraph`` naming it (oracle/onnx_ref.py executes
graphs on the CPU for the tests only).  Supported today: Conv (groups 1: 1x1 / 3x3; depthwise: 3x3 / 5x5; stride 1 / 2,
"same" padding), ConvTranspose 2x2 / stride 2, BatchNormalization (folded), Relu / HardSwish / Sigmoid / HardSigmoid /
Relu6, a BatchNormalization that stands alone, Add, Mul by a per-channel gate, MaxPool(3, 2, 1) and k x k / stride k,
AveragePool k x k / stride k, GlobalAveragePool, Resize / Upsample (nearest, integer factor, by scales or sizes), Concat over
channels, Gemm / Flatten after a global pool.

Two arithmetic modes, like the dedicated launch graphs (``precision=`` of the constructor): ``"bf16"`` -- bf16 operands, fp32 accumulate, every
activation rounded to bf16 (the throughput mode) -- and ``"bf16x3"``, the tolerance mode: every activation is a (hi | lo) pair of bf16 halves
(``_Act.t[..., :Cp]`` and ``[..., Cp:]``, value = hi + lo: 16 significant bits), convolutions / GEMMs run as three MFMA passes over (hi, lo) weight
tiles, every other operator computes on hi + lo in fp32 and splits its result again (``split`` of the pt_op_* entry points, ABI 12).  This is the
mode whose outputs agree with an fp32 execution of the graph (oracle/onnx_ref.py; onnxruntime in the reference) to 1e-3.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import os
import threading

import numpy as np
import torch

from . import lib as L
from .engine import HipEngine
from .onnx_import import Layer, OnnxGraph, UnsupportedOnnxGraph, load_onnx
from .weights import split_bf16, tile_conv_weight, tile_conv_weight_x3

__all__ = ["HipGraphExecutor"]

_ACT_CODE = {None: 0, "relu": 1, "hardswish": 2}             # fused into the conv / depthwise epilogue
_ACT_KIND = {"relu": 1, "hardswish": 2, "sigmoid": 4, "hardsigmoid": 5, "relu6": 6, "gelu": 7, "swish": 8}    # pt_op_act


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


@dataclass
class _Act:
    """an activation: bf16 NHWC tensor with padded channels, and its real channel count"""
    t: torch.Tensor
    c: int
    flat: bool = False      # the ONNX tensor is [B, C] (after Flatten / Gemm), held here as [B, 1, 1, C]
    seq: bool = False       # the ONNX tensor is [B, T, C] (token rows), held here as [B, 1, T, Cpad]

    def shape(self):
        """the ONNX (logical) shape"""
        B, H, W = self.t.shape[:3]
        if self.flat:
            return (B, self.c)
        if self.seq:
            return (B, W, self.c)
        return (B, self.c, H, W)


@dataclass
class _View:
    """a logical tensor that is a re-indexing of a materialised one: idx has the logical shape and holds, per element, the flat position
    in ``base.shape()`` (row-major); scale: a scalar the values are multiplied with (q * d ** -0.5 in front of an attention)"""
    base: _Act
    idx: np.ndarray
    scale: float = 1.0


@dataclass
class _Scores:
    """q k^T of a recognised fused-qkv attention (never materialised): soft-max and the product with v follow"""
    base: _Act
    heads: int
    d: int
    scale: float
    soft: bool = False


_CAPTURE_LOCK = threading.Lock()      # graph capture is serialised across executors and host threads (see HipGraphExecutor._graphed)


class HipGraphExecutor:
    def __init__(self, src, engine: Optional[HipEngine] = None, device: int = 0, precision: str = "bf16"):
        if precision not in ("bf16", "bf16x3", "f16"):
            raise ValueError(f"precision '{precision}': 'bf16' (throughput), 'f16' (single-pass IEEE half, the reference's fp16) or 'bf16x3' "
                             "(tolerance mode, (hi | lo) activations)")
        self.precision = precision
        self.split = precision == "bf16x3"
        self.m = 2 if self.split else 1                   # halves per activation: .t[..., :Cp] = hi, .t[..., Cp:] = lo
        self.fmt = "f16" if precision == "f16" else "bf16"       # 16-bit storage format of activations and weight tiles (csrc/act16.h)
        self.adt = torch.float16 if precision == "f16" else torch.bfloat16
        self.graph: OnnxGraph = src if isinstance(src, OnnxGraph) else load_onnx(src)
        want = {"bf16": L.PT_PRECISION_BF16, "bf16x3": L.PT_PRECISION_BF16X3, "f16": L.PT_PRECISION_F16}[precision]
        if engine is None:
            engine = HipEngine(device)
            engine.set_precision(want)
        elif (engine.precision == L.PT_PRECISION_F16) != (precision == "f16"):
            # the operator entry points interpret 16-bit tensors in the ENGINE's storage format: a mismatch would read fp16 bits as bf16
            raise ValueError(f"executor precision '{precision}' on an engine whose precision is {engine.precision}: set the engine's precision "
                             "(HipEngine.set_precision) to the executor's before building it")
        self.eng = engine
        self.layers: List[Layer] = self.graph.layers()
        bad = [f"{l.name} ({l.attrs.get('onnx_op', l.op)})" for l in self.layers if l.op == "unsupported"]
        if bad:
            raise UnsupportedOnnxGraph("operators without an engine kernel: " + ", ".join(bad) + "\n" + self.graph.summary())
        self.inputs = [i for i in self.graph.model.inputs if i.name not in self.graph.init]
        self.outputs = [o.name for o in self.graph.model.outputs]
        if len(self.inputs) != 1:
            raise UnsupportedOnnxGraph(f"the executor takes graphs with one image input, this one has {[i.name for i in self.inputs]}")
        self._dev: Dict[int, Dict[str, torch.Tensor]] = {}      # layer index -> uploaded operands (filled on first use)
        self._graphs: Dict[tuple, object] = {}                  # input shape -> captured HIP graph (run_device_graphed), least recently used first
        self._seen: set = set()                                 # shapes that ran once eagerly (the next call captures)
        self._bad: set = set()                                  # shapes whose capture raised: eager from then on
        self._fuse = self._plan_add_fusion()

    def _plan_add_fusion(self) -> Dict[int, tuple]:
        """Residual adds folded into the producing convolution's epilogue at load time: conv k (groups 1, 1x1 / 3x3, no activation of its own)
        whose ONLY consumer is an Add of two computed tensors, the other one produced earlier -> {k: (index of the Add, name of the other operand,
        index of a following ReLU / hardswish that is the Add's only consumer or None)}.  One launch and one read + write of the map less per
        residual block; the arithmetic (fp32 sum of bias, product and residual, then the activation, then ONE rounding) is what the fused launch
        graphs do (ConvDesc.res_mode 1)."""
        uses: Dict[str, List[int]] = {}
        made: Dict[str, int] = {i.name: -1 for i in self.inputs}
        for j, lay in enumerate(self.layers):
            for nm in lay.inputs:
                uses.setdefault(nm, []).append(j)
            for o in lay.outputs:
                made[o] = j
        plan: Dict[int, tuple] = {}
        taken = set()
        for k, lay in enumerate(self.layers):
            if lay.op != "conv" or lay.act is not None or lay.attrs.get("group", 1) != 1 or lay.attrs["kernel"][0] not in (1, 3) or len(lay.outputs) != 1:
                continue
            out = lay.outputs[0]
            if out in self.outputs or len(uses.get(out, ())) != 1:
                continue
            j = uses[out][0]
            add = self.layers[j]
            if add.op != "add" or add.extra or len(add.inputs) != 2 or j in taken or len(add.outputs) != 1:
                continue
            other = add.inputs[0] if add.inputs[1] == out else add.inputs[1]
            if other == out or made.get(other, 1 << 30) >= k:
                continue
            act = None
            ao = add.outputs[0]
            if ao not in self.outputs and len(uses.get(ao, ())) == 1:
                nx = self.layers[uses[ao][0]]
                if nx.op == "act" and nx.attrs.get("kind") in ("relu", "hardswish") and len(nx.outputs) == 1:
                    act = uses[ao][0]
            plan[k] = (j, other, act)
            taken.add(j)
        return plan

    # ---- weights ---------------------------------------------------------------------------------------------------
    def _up(self, a: np.ndarray, dtype=None) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.eng._tdev)

    def _conv_operands(self, k: int, lay: Layer, cin_pad: int, cin: int):
        d = self._dev.get(k)
        if d is not None and d["cin_pad"] == cin_pad:
            return d
        w = torch.from_numpy(lay.weight)
        n = w.shape[0]
        if w.shape[1] != cin:
            raise UnsupportedOnnxGraph(f"{lay.name}: weight expects {w.shape[1]} input channels, the tensor has {cin}")
        wp = torch.zeros(_pad64(n), cin_pad, w.shape[2], w.shape[3])
        wp[:n, :cin] = w
        bp = torch.zeros(_pad64(n))
        if lay.bias is not None:
            bp[:n] = torch.from_numpy(lay.bias)
        d = {"cin_pad": cin_pad, "w": self._up(self._tile(wp).view(np.int16)), "b": bp.to(self.eng._tdev), "n": n}
        self._dev[k] = d
        return d

    def _tile(self, w: torch.Tensor) -> np.ndarray:
        return tile_conv_weight_x3(w) if self.split else tile_conv_weight(w, self.fmt)

    def _cp(self, x: _Act) -> int:
        """padded channels of ONE half"""
        return x.t.shape[-1] // self.m

    def _zeros(self, lead, cp: int) -> torch.Tensor:
        return torch.zeros(tuple(lead) + (cp * self.m,), dtype=self.adt, device=self.eng._tdev)

    def _copy(self, src: torch.Tensor, dst: torch.Tensor, n: int, src_coff: int = 0, dst_coff: int = 0):
        """channels [src_coff, src_coff + n) of src -> [dst_coff, ..) of dst, both halves in the tolerance mode"""
        self.eng.op_copy_channels(src, dst, n, src_coff=src_coff, dst_coff=dst_coff)
        if self.split:
            self.eng.op_copy_channels(src, dst, n, src_coff=src.shape[-1] // 2 + src_coff, dst_coff=dst.shape[-1] // 2 + dst_coff)

    def values(self, a: _Act) -> torch.Tensor:
        """the real channels of an activation as fp32 (hi + lo in the tolerance mode); a final fp32 Softmax is returned as it is"""
        if a.t.dtype == torch.float32:
            return a.t[..., :a.c]
        v = a.t[..., :a.c].float()
        if self.split:
            cp = a.t.shape[-1] // 2
            v = v + a.t[..., cp:cp + a.c].float()
        return v

    # ---- layers ----------------------------------------------------------------------------------------------------
    def _conv(self, k: int, lay: Layer, x: _Act, res: Optional[_Act] = None, res_act: int = 0) -> _Act:
        """res: the other operand of a residual Add folded into this convolution (plain group-1 convolutions without an activation of their own),
        res_act: the epilogue activation that follows the Add"""
        a = lay.attrs
        kh, kw = a["kernel"]
        sh, sw = a["strides"]
        if a["dilations"] != [1, 1] or kh != kw or sh != sw or sh not in (1, 2):
            raise UnsupportedOnnxGraph(f"{lay.name}: conv {a} (square kernels, stride 1 / 2, no dilation)")
        if list(a["pads"]) != [kh // 2] * 4:
            raise UnsupportedOnnxGraph(f"{lay.name}: conv padding {a['pads']} is not k // 2 on every side")
        fused = _ACT_CODE.get(lay.act, None) if lay.act in _ACT_CODE else 0
        post = None if lay.act in _ACT_CODE else lay.act
        if a["group"] == 1 and kh == 7 and sh == 2 and x.c == 3 and lay.weight.shape[0] <= 64 and lay.act == "relu":
            # the ResNet stem: 7x7 / stride 2 on the 3-channel image + ReLU -> the engine's stem kernel (K = [7][8][4], NHWC4 input)
            d = self._dev.get(k)
            if d is None:
                w = torch.from_numpy(lay.weight)
                n0 = w.shape[0]
                stem = torch.zeros(64, 7, 8, 4)
                stem[:n0, :, :7, :3] = w.permute(0, 2, 3, 1)
                bp = torch.zeros(64)
                if lay.bias is not None:
                    bp[:n0] = torch.from_numpy(lay.bias)
                from .weights import to_bf16_bits
                if self.split:
                    sh, sl = split_bf16(stem)
                    wst = np.stack([to_bf16_bits(sh).reshape(64, 224), to_bf16_bits(sl).reshape(64, 224)])
                else:
                    wst = to_bf16_bits(stem, self.fmt).reshape(64, 224)
                d = self._dev[k] = {"w": self._up(wst.view(np.int16)), "b": bp.to(self.eng._tdev), "n": n0}
            if x.t.shape[1] % 2 or x.t.shape[2] % 2:
                raise UnsupportedOnnxGraph(f"{lay.name}: the 7x7 / stride-2 stem kernel needs even image sizes")
            x4 = self._zeros(x.t.shape[:-1], 4)           # NHWC4 image ([hi rgb0 | lo rgb0] in the tolerance mode)
            self._copy(x.t, x4, 3)
            return _Act(self.eng.op_stem7x7(x4, d["w"], d["b"], split=self.split), d["n"])
        if a["group"] == 1:
            if kh not in (1, 3):
                raise UnsupportedOnnxGraph(f"{lay.name}: {kh}x{kw} convolution (the MFMA kernel covers 1x1 and 3x3)")
            d = self._conv_operands(k, lay, self._cp(x), x.c)
            y = _Act(self.eng.op_conv2d(x.t, d["w"], d["b"], kh, sh, relu=fused if res is None else res_act, res=None if res is None else res.t,
                                        res_mode=0 if res is None else 1, split=int(self.split)), d["n"])
        elif a["group"] == x.c and lay.weight.shape[0] == x.c and lay.weight.shape[1] == 1:
            if kh not in (3, 5):
                raise UnsupportedOnnxGraph(f"{lay.name}: depthwise {kh}x{kw} (3x3 and 5x5 are built)")
            d = self._dev.get(k)
            if d is None:
                cp = self._cp(x)
                wt = np.zeros((kh * kw, cp), np.float32)
                wt[:, :x.c] = lay.weight.reshape(x.c, kh * kw).T
                bt = np.zeros((cp,), np.float32)
                if lay.bias is not None:
                    bt[:x.c] = lay.bias
                d = self._dev[k] = {"w": self._up(wt), "b": self._up(bt)}
            y = _Act(self.eng.op_dwconv(x.t, d["w"], d["b"], kh, sh, fused, split=self.split), x.c)
        else:
            raise UnsupportedOnnxGraph(f"{lay.name}: grouped convolution (group {a['group']} of {x.c} channels)")
        return self._post_act(lay, y, post)

    def _convT(self, k: int, lay: Layer, x: _Act) -> _Act:
        a = lay.attrs
        if a["kernel"] != [2, 2] or a["strides"] != [2, 2] or a["group"] != 1 or any(a["pads"]):
            raise UnsupportedOnnxGraph(f"{lay.name}: ConvTranspose {a} (2x2 / stride 2 is built)")
        d = self._dev.get(k)
        if d is None:
            w = torch.from_numpy(lay.weight)                       # [ci, co, 2, 2]
            ci, co = w.shape[:2]
            cop, cip = _pad64(co), self._cp(x)
            wq = torch.zeros(2, 2, cop, cip)
            wq[:, :, :co, :ci] = w.permute(2, 3, 1, 0)              # N index = (dy * 2 + dx) * cop + co: the pixel-shuffle epilogue
            bq = torch.zeros(cop)
            if lay.bias is not None:
                bq[:co] = torch.from_numpy(lay.bias)
            d = self._dev[k] = {"w": self._up(self._tile(wq.reshape(4 * cop, cip, 1, 1)).view(np.int16)),
                                "b": bq.repeat(4).to(self.eng._tdev), "n": co, "cop": cop}
        fused = _ACT_CODE.get(lay.act, None) if lay.act in _ACT_CODE else 0
        post = None if lay.act in _ACT_CODE else lay.act
        y = _Act(self.eng.op_conv2d(x.t, d["w"], d["b"], 1, 1, relu=fused, shuffle_cout=d["cop"], split=int(self.split)), d["n"])
        return self._post_act(lay, y, post)

    def _post_act(self, lay: Layer, y: _Act, kind: Optional[str]) -> _Act:
        if kind is None:
            return y
        if kind not in _ACT_KIND:
            raise UnsupportedOnnxGraph(f"{lay.name}: activation '{kind}'")
        return _Act(self.eng.op_act(y.t, _ACT_KIND[kind], lay.attrs.get("act_alpha", 0.2), lay.attrs.get("act_beta", 0.5), split=self.split), y.c, y.flat, y.seq)

    # ---- the graph -------------------------------------------------------------------------------------------------
    def run(self, x) -> List[np.ndarray]:
        """x: float NCHW image batch (numpy / torch) -> the graph outputs as float32 NCHW arrays, in graph order"""
        xt = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(torch.float32)
        if xt.ndim != 4:
            raise ValueError(f"expected an NCHW batch, got shape {tuple(xt.shape)}")
        outs = []
        nhwc = xt.permute(0, 2, 3, 1).to(self.eng._tdev)
        for a in self.run_device(nhwc if self.split else nhwc.to(self.adt), xt.shape[1]):
            v = self.values(a)
            if a.seq:                                    # token rows: [B, T, C]
                outs.append(v[:, 0].contiguous().cpu().numpy())
                continue
            o = v.permute(0, 3, 1, 2).contiguous().cpu().numpy()
            outs.append(o.reshape(o.shape[0], -1) if a.flat else o)
        return outs

    # ---- views: shape plumbing on index arrays ---------------------------------------------------------------------------
    @staticmethod
    def _as_view(v) -> _View:
        if isinstance(v, _View):
            return v
        shp = v.shape()
        return _View(v, np.arange(int(np.prod(shp)), dtype=np.int64).reshape(shp))

    def _realize(self, v, what: str) -> _Act:
        """a view -> an activation a kernel can read, without arithmetic: the same memory under another logical shape where the index
        array says so, a channel slice through pt_op_copy_channels otherwise"""
        if isinstance(v, _Act):
            return v
        if not isinstance(v, _View):
            raise UnsupportedOnnxGraph(f"{what}: operand is a {type(v).__name__}, not a tensor on the device")
        if v.scale != 1.0:
            raise UnsupportedOnnxGraph(f"{what}: a scaled view is only consumed by the attention pattern")
        b, idx = v.base, v.idx
        bs = b.shape()
        ident = np.arange(int(np.prod(bs)), dtype=np.int64)
        if idx.size == ident.size and np.array_equal(idx.reshape(-1), ident):
            # same element order: a pure reshape.  Token rows / maps share the [.., Cpad] memory only if the channel axis is kept
            if idx.shape == bs:
                return b
            if len(idx.shape) == 3 and idx.shape[0] == bs[0] and idx.shape[2] == b.c and b.seq:
                return _Act(b.t.reshape(bs[0], 1, idx.shape[1], b.t.shape[-1]), b.c, seq=True)
            if len(idx.shape) == 2 and (b.flat or (b.seq and bs[1] == 1) or (not b.seq and b.t.shape[1] == 1 and b.t.shape[2] == 1)) and idx.shape[1] == b.c:
                return _Act(b.t.reshape(bs[0], 1, 1, b.t.shape[-1]), b.c, flat=True)
        if not b.seq and not b.flat and len(idx.shape) == 3 and len(bs) == 4:
            # map [B, C, H, W] -> tokens [B, H W, C'] (flatten(2).transpose(1, 2)), optionally of a channel slice
            B, C, H, W = bs
            if idx.shape[0] == B and idx.shape[1] == H * W:
                c0, cn = int(idx[0, 0, 0]) // (H * W), idx.shape[2]
                want = (np.arange(B)[:, None, None] * C + (c0 + np.arange(cn))[None, None, :]) * (H * W) + np.arange(H * W)[None, :, None]
                if c0 + cn <= C and np.array_equal(idx, want):
                    src = b if (c0 == 0 and cn == C) else self._slice_channels(b, c0, cn)
                    return _Act(src.t.reshape(B, 1, H * W, src.t.shape[-1]), cn, seq=True)
        if not b.seq and not b.flat and len(idx.shape) == 4 and len(bs) == 4 and idx.shape[0] == bs[0] and idx.shape[2:] == bs[2:]:
            B, C, H, W = bs                               # channel slice of a map
            c0, cn = int(idx[0, 0, 0, 0]) // (H * W), idx.shape[1]
            if c0 + cn <= C and np.array_equal(idx, np.arange(int(np.prod(bs))).reshape(bs)[:, c0:c0 + cn]):
                return self._slice_channels(b, c0, cn)
        if b.seq and len(idx.shape) == 4 and len(bs) == 3 and idx.shape[0] == bs[0] and idx.shape[1] == bs[2] and idx.shape[2] * idx.shape[3] == bs[1]:
            B, T, C = bs                                  # tokens [B, T, C] -> map [B, C, H, W] (transpose(1, 2).reshape)
            want = (np.arange(B)[:, None, None] * T + np.arange(T)[None, None, :]) * C + np.arange(C)[None, :, None]
            if np.array_equal(idx.reshape(B, C, T), want):
                return _Act(b.t.reshape(B, idx.shape[2], idx.shape[3], b.t.shape[-1]), C)
        if b.seq and len(idx.shape) == 3 and len(bs) == 3 and idx.shape[:2] == bs[:2]:
            B, T, C = bs                                  # channel slice of token rows
            c0, cn = int(idx[0, 0, 0]), idx.shape[2]
            if c0 + cn <= C and np.array_equal(idx, np.arange(int(np.prod(bs))).reshape(bs)[:, :, c0:c0 + cn]):
                return self._slice_channels(b, c0, cn)
        raise UnsupportedOnnxGraph(f"{what}: a tensor of logical shape {tuple(idx.shape)} re-indexing a {bs} tensor in a way no kernel reads "
                                   "(built: map <-> token rows, channel slices, the head split of fused q / k / v rows)")

    def _slice_channels(self, b: _Act, c0: int, cn: int) -> _Act:
        out = self._zeros(b.t.shape[:-1], _pad64(cn))
        self._copy(b.t, out, cn, src_coff=c0, dst_coff=0)
        return _Act(out, cn, flat=b.flat, seq=b.seq)

    def _glue(self, lay: Layer, env) -> list:
        """Shape / Reshape / Transpose / Flatten / Squeeze / Unsqueeze / Gather / Slice / Split / Concat / Cast / Expand on host constants and
        on views (index arrays): -> one value per output"""
        op = lay.attrs.get("onnx_op", "Concat" if lay.op == "concat" else lay.op)
        at = lay.attrs.get("node_attrs", {})
        names = lay.attrs.get("all_inputs", lay.inputs)

        def val(i):
            nm = names[i] if i < len(names) else ""
            if nm == "":
                return None
            return env[nm] if nm in env else self.graph.init.get(nm)
        x = val(0)
        is_c = isinstance(x, np.ndarray)

        def ints(v):
            return [int(q) for q in np.asarray(v).reshape(-1)]
        if op == "Shape":
            if is_c:
                return [np.asarray(x.shape, np.int64)]
            return [np.asarray(x.idx.shape if isinstance(x, _View) else x.shape(), np.int64)]
        if op in ("Cast", "Identity"):
            return [x]
        if op == "Concat" and all(isinstance(val(i), np.ndarray) for i in range(len(names))):
            return [np.concatenate([np.atleast_1d(val(i)) for i in range(len(names))], axis=int(at.get("axis", lay.attrs.get("axis", 0))))]
        if op == "ConstantOfShape":
            v = at.get("value")
            return [np.full(ints(x), 0 if v is None else np.asarray(v).reshape(-1)[0])]
        arr = x if is_c else self._as_view(x).idx

        def wrap(a):
            return a if is_c else _View(self._as_view(x).base, np.ascontiguousarray(a), self._as_view(x).scale)
        if op == "Reshape":
            shp = ints(val(1))
            shp = [arr.shape[i] if d == 0 and not int(at.get("allowzero", 0)) else d for i, d in enumerate(shp)]
            return [wrap(arr.reshape(shp))]
        if op == "Flatten":
            ax = int(at.get("axis", 1))
            return [wrap(arr.reshape(int(np.prod(arr.shape[:ax])), -1))]
        if op == "Transpose":
            return [wrap(arr.transpose(at.get("perm", list(range(arr.ndim))[::-1])))]
        if op in ("Squeeze", "Unsqueeze"):
            axes = ints(at["axes"]) if "axes" in at else (ints(val(1)) if val(1) is not None else None)
            if op == "Squeeze":
                return [wrap(np.squeeze(arr, axis=None if axes is None else tuple(axes)))]
            out = arr
            for a_ in sorted(a_ if a_ >= 0 else a_ + arr.ndim + len(axes) for a_ in axes):
                out = np.expand_dims(out, a_)
            return [wrap(out)]
        if op == "Gather":
            ind = val(1)
            if not isinstance(ind, np.ndarray):
                raise UnsupportedOnnxGraph(f"{lay.name}: Gather with computed indices")
            return [wrap(np.take(arr, ind.astype(np.int64), axis=int(at.get("axis", 0))))]
        if op == "Slice":
            if "starts" in at:
                st, en, ax, sp = ints(at["starts"]), ints(at["ends"]), ints(at.get("axes", range(len(at["starts"])))), [1] * len(at["starts"])
            else:
                st, en = ints(val(1)), ints(val(2))
                ax = ints(val(3)) if val(3) is not None else list(range(len(st)))
                sp = ints(val(4)) if val(4) is not None else [1] * len(st)
            sl = [slice(None)] * arr.ndim
            for s_, e_, a_, p_ in zip(st, en, ax, sp):
                sl[a_] = slice(s_, None if e_ > (1 << 60) else e_, p_)
            return [wrap(arr[tuple(sl)])]
        if op == "Split":
            ax = int(at.get("axis", 0))
            sizes = ints(at["split"]) if "split" in at else (ints(val(1)) if val(1) is not None else None)
            parts = np.split(arr, len(lay.outputs), axis=ax) if sizes is None else np.split(arr, np.cumsum(sizes)[:-1], axis=ax)
            return [wrap(p_) for p_ in parts]
        if op == "Expand":
            return [wrap(np.broadcast_to(arr, np.broadcast_shapes(arr.shape, tuple(ints(val(1))))))]
        raise UnsupportedOnnxGraph(f"{lay.name}: {op} has no executor")

    def _attention(self, lay: Layer, a, b):
        """MatMul of two views: q k^T of a fused-qkv attention (-> _Scores), or soft-max(scores) v (-> the attention kernel)"""
        if isinstance(a, _Scores) and a.soft and isinstance(b, _View):
            base = a.base
            B, T, C3 = base.shape()
            C = a.heads * a.d
            want = ((np.arange(B)[:, None, None, None] * T + np.arange(T)[None, None, :, None]) * C3 + 2 * C
                    + np.arange(a.heads)[None, :, None, None] * a.d + np.arange(a.d)[None, None, None, :])
            if b.base is not base or b.idx.shape != want.shape or not np.array_equal(b.idx, want) or b.scale != 1.0:
                raise UnsupportedOnnxGraph(f"{lay.name}: the value operand is not the v part of the fused q / k / v rows")
            out = _Act(self.eng.op_attention(base.t, a.heads, a.d, a.scale, _pad64(C), split=self.split), C, seq=True)
            idx = ((np.arange(B)[:, None, None, None] * T + np.arange(T)[None, None, :, None]) * C
                   + np.arange(a.heads)[None, :, None, None] * a.d + np.arange(a.d)[None, None, None, :])
            return _View(out, idx)
        if isinstance(a, _View) and isinstance(b, _View) and a.base is b.base and a.base.seq and a.idx.ndim == 4 and b.idx.ndim == 4:
            base = a.base
            B, T, C3 = base.shape()
            _, h, Tq, d = a.idx.shape
            if Tq == T and C3 == 3 * h * d and b.idx.shape == (B, h, d, T):
                C = h * d
                q_want = ((np.arange(B)[:, None, None, None] * T + np.arange(T)[None, None, :, None]) * C3
                          + np.arange(h)[None, :, None, None] * d + np.arange(d)[None, None, None, :])
                if np.array_equal(a.idx, q_want) and np.array_equal(b.idx.transpose(0, 1, 3, 2), q_want + C):
                    return _Scores(base, h, d, a.scale * b.scale)
        raise UnsupportedOnnxGraph(f"{lay.name}: MatMul of two computed tensors outside the fused-qkv attention pattern "
                                   "(q, k, v = qkv.reshape(B, T, 3, heads, d).permute(2, 0, 3, 1, 4))")

    # ---- captured HIP graphs -------------------------------------------------------------------------------------------
    MAX_GRAPHS = 8          # captured graphs kept per executor (each pins a private pool of its peak activations): least recently used goes first

    def _graphed(self, key, nhwc: torch.Tensor, walk):
        """walk(static_input) replayed from a captured HIP graph, one per key.  A shape's first call runs eagerly (weights are uploaded, the engine's
        arenas sized) and is only remembered (`_seen`, bounded); the second captures; later calls copy the input into the captured buffer and replay.
        Only CAPTURED graphs count against MAX_GRAPHS, and the least recently used one is dropped for a new shape -- a dynamic-shape detector does not
        run eagerly for the rest of the process after its first eight page sizes.  Capture is serialised (one lock for all executors:
        capture is a property of the device's allocator) in thread-local capture mode, so allocations of other host threads (the pipeline's
        enqueue / collect threads) do not invalidate it; a capture that raises anyway falls back to the eager walk for that shape."""
        if os.environ.get("PT_ONNX_GRAPH", "1") == "0":
            return walk(nhwc)
        ent = self._graphs.get(key)
        if ent is None:
            if key in self._bad or key not in self._seen:
                if len(self._seen) >= 64:
                    self._seen.clear()
                self._seen.add(key)
                return walk(nhwc)
            torch.cuda.synchronize(self.eng._tdev)
            static_in = nhwc.clone()
            g = torch.cuda.CUDAGraph()
            try:
                with _CAPTURE_LOCK:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        outs = walk(static_in)
            except Exception:      # noqa: BLE001 -- the capture is an optimisation: the eager walk is always available
                torch.cuda.synchronize(self.eng._tdev)
                self._bad.add(key)
                return walk(nhwc)
            while len(self._graphs) >= self.MAX_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))          # dicts keep insertion order: the first key is the least recently used
            ent = (g, static_in, outs)
        else:
            self._graphs.pop(key)                                   # re-inserted below as the most recently used
        self._graphs[key] = ent
        g, static_in, outs = ent
        static_in.copy_(nhwc)
        g.replay()
        return outs

    def run_device_graphed(self, nhwc: torch.Tensor, c: int) -> List[_Act]:
        """run_device() replayed from a captured HIP graph, one per input shape.  The layer list is walked from Python -- tens of launches of a
        few microseconds each with the interpreter between them: a 52-layer recogniser spends 1.8 ms per line that way, most of it on the
        host.  Same kernels, same arguments: same bits as run_device().  The returned activations are the graph's own buffers -- read them
        before the next call.  PT_ONNX_GRAPH=0 runs eagerly (see _graphed for the cache's rules)."""
        return self._graphed((tuple(nhwc.shape), nhwc.dtype, int(c)), nhwc, lambda x: self.run_device(x, c))

    def run_lines_graphed(self, nhwc: torch.Tensor, c: int) -> List[List[_Act]]:
        """a batch [n, H, W, c] through a graph whose batch size is baked in as 1 (static exports): the n single-image walks are captured
        into ONE HIP graph, so that a mini-batch costs one copy and one replay -> per image the outputs of run_device().  Same rules as
        run_device_graphed."""
        n = int(nhwc.shape[0])
        return self._graphed(("lines",) + tuple(nhwc.shape) + (nhwc.dtype, int(c)), nhwc, lambda x: [self.run_device(x[i:i + 1], c) for i in range(n)])

    def run_device(self, nhwc: torch.Tensor, c: int) -> List[_Act]:
        """bf16 NHWC batch on the device whose first ``c`` channels are the image (what pt_det_preprocess / pt_cls_preprocess
        write) -> the graph outputs as device activations (bf16 NHWC, ``.t[..., :.c]`` are the real channels; ``values()`` gives them as fp32 in
        either mode): no host trip.  Tolerance mode: an fp32 batch is split into its (hi, lo) halves here (a cast: plumbing), a bf16 batch has lo = 0."""
        cp = (c + 31) // 32 * 32                         # the image itself: 32 channels are enough for the first GEMM's K
        first = self._zeros(nhwc.shape[:-1], cp)
        if nhwc.dtype == torch.float32:
            hi = nhwc[..., :c].to(self.adt)
            self.eng.op_copy_channels(hi.contiguous(), first, c)
            if self.split:
                lo = (nhwc[..., :c] - hi.float()).to(self.adt)
                self.eng.op_copy_channels(lo.contiguous(), first, c, dst_coff=cp)
        else:
            if nhwc.dtype != self.adt:
                raise ValueError(f"run_device: a {nhwc.dtype} batch for a '{self.precision}' executor (expected {self.adt} or float32)")
            nhwc = nhwc.contiguous()
            self.eng.op_copy_channels(nhwc, first, c)
            if self.split:
                # the engine's pre-processing kernels in PT_PRECISION_BF16X3 write [hi (c4) | lo (c4)]: the lo half travels too (without it the
                # normalised image is rounded to 8 significant bits before the first conv and the 1e-3 contract of this mode is not met --
                # ADVICE r04); a plain 16-bit batch (c4 channels, an engine in PT_PRECISION_BF16) has lo = 0
                half = nhwc.shape[-1] // 2
                if nhwc.shape[-1] % 2 == 0 and half >= c and self.eng.split:
                    self.eng.op_copy_channels(nhwc, first, c, src_coff=half, dst_coff=cp)
        env: Dict[str, object] = {self.inputs[0].name: _Act(first, c)}
        R = self._realize
        skip = set()                                     # Add / activation layers folded into a convolution's epilogue (self._fuse)
        for k, lay in enumerate(self.layers):
            op = lay.op
            if op == "glue" or (op == "concat" and all((i in self.graph.init or isinstance(env.get(i), np.ndarray)) for i in lay.inputs)):
                for o, v in zip(lay.outputs, self._glue(lay, env)):
                    env[o] = np.asarray(v) if isinstance(v, (np.generic, int, float)) else v      # host constants stay ndarrays (0-d included)
                continue
            raw = [env[i] for i in lay.inputs if i in env]
            if op in ("add", "mul", "sub", "div") and lay.extra and all(np.asarray(v).size == 1 for v in lay.extra.values()) and len(raw) == 1 \
                    and isinstance(raw[0], (_View, _Scores, np.ndarray)):
                cst = float(np.asarray(next(iter(lay.extra.values()))).reshape(-1)[0])
                v = raw[0]
                if isinstance(v, np.ndarray):            # integer shape arithmetic
                    first_is_const = lay.attrs["all_inputs"][0] not in env
                    a_, b_ = (cst, v) if first_is_const else (v, cst)
                    env[lay.outputs[0]] = np.asarray({"add": np.add, "mul": np.multiply, "sub": np.subtract, "div": np.floor_divide if v.dtype.kind in "iu" else np.divide}[op](a_, b_))
                    continue
                if op in ("mul", "div"):                 # the 1 / sqrt(d) of an attention, on q or on the scores
                    f = cst if op == "mul" else 1.0 / cst
                    env[lay.outputs[0]] = _View(v.base, v.idx, v.scale * f) if isinstance(v, _View) else _Scores(v.base, v.heads, v.d, v.scale * f, v.soft)
                    continue
            if op == "matmul":
                env[lay.outputs[0]] = self._attention(lay, raw[0], raw[1])
                continue
            if op == "act" and lay.attrs["kind"] == "softmax" and raw and isinstance(raw[0], _Scores):
                if lay.attrs.get("axis", -1) not in (-1, 3):
                    raise UnsupportedOnnxGraph(f"{lay.name}: attention soft-max over axis {lay.attrs.get('axis')}")
                env[lay.outputs[0]] = _Scores(raw[0].base, raw[0].heads, raw[0].d, raw[0].scale, True)
                continue
            if k in skip:
                continue
            ins = [R(v, lay.name) for v in raw]
            if op == "conv":
                fz = self._fuse.get(k)
                y = None
                if fz is not None and isinstance(env.get(fz[1]), _Act):
                    j_add, other, j_act = fz
                    r_ = env[other]
                    sh_ = lay.attrs["strides"][0]
                    ho, wo = (ins[0].t.shape[1] - 1) // sh_ + 1, (ins[0].t.shape[2] - 1) // sh_ + 1
                    if not (r_.seq or r_.flat or ins[0].seq or ins[0].flat) and r_.c == lay.weight.shape[0] and tuple(r_.t.shape[1:3]) == (ho, wo) \
                            and r_.t.shape[-1] == _pad64(r_.c) * self.m:
                        code = _ACT_CODE[self.layers[j_act].attrs["kind"]] if j_act is not None else 0
                        y = self._conv(k, lay, ins[0], res=r_, res_act=code)
                        skip.add(j_add)
                        for o in self.layers[j_add].outputs:
                            env[o] = y
                        if j_act is not None:
                            skip.add(j_act)
                            for o in self.layers[j_act].outputs:
                                env[o] = y
                if y is None:
                    y = self._conv(k, lay, ins[0])
            elif op == "convT":
                y = self._convT(k, lay, ins[0])
            elif op == "maxpool":
                a = lay.attrs
                kk, st, pd = a["kernel"], a["strides"], a["pads"]
                if kk[0] != kk[1] or st[0] != st[1] or len(set(pd)) != 1 or a.get("ceil_mode"):
                    raise UnsupportedOnnxGraph(f"{lay.name}: MaxPool {a}")
                y = _Act(self.eng.op_maxpool(ins[0].t, kk[0], st[0], pd[0], split=self.split), ins[0].c)
            elif op == "avgpool":
                a = lay.attrs
                kk, st, pd = a["kernel"], a["strides"], a["pads"]
                if kk[0] != kk[1] or st != kk or any(pd) or a.get("ceil_mode") or ins[0].t.shape[1] % kk[0] or ins[0].t.shape[2] % kk[0]:
                    raise UnsupportedOnnxGraph(f"{lay.name}: AveragePool {a} (k x k / stride k without padding is built)")
                y = _Act(self.eng.op_avgpool(ins[0].t, kk[0], split=self.split), ins[0].c)
            elif op == "bn":
                # a BatchNormalization that could not be folded into a convolution: per-channel affine = a depthwise 3x3 whose only
                # non-zero tap is the centre one
                d = self._dev.get(k)
                if d is None:
                    e_ = lay.extra
                    sc = e_["gamma"].astype(np.float64) / np.sqrt(e_["var"].astype(np.float64) + lay.attrs["epsilon"])
                    cp = self._cp(ins[0])
                    wt = np.zeros((9, cp), np.float32)
                    wt[4, :ins[0].c] = sc
                    bt = np.zeros((cp,), np.float32)
                    bt[:ins[0].c] = e_["beta"].astype(np.float64) - e_["mean"].astype(np.float64) * sc
                    d = self._dev[k] = {"w": self._up(wt), "b": self._up(bt)}
                y = _Act(self.eng.op_dwconv(ins[0].t, d["w"], d["b"], 3, 1, 0, split=self.split), ins[0].c)
            elif op == "gap":
                y = _Act(self.eng.op_chan_mean(ins[0].t, split=self.split), ins[0].c)
            elif op == "layernorm":
                x = ins[0]
                if not (x.seq or x.flat):
                    raise UnsupportedOnnxGraph(f"{lay.name}: LayerNormalization of a feature map (token rows [B, T, C] are built)")
                d = self._dev.get(k)
                if d is None:
                    g_, b_ = lay.extra.get("gamma"), lay.extra.get("beta")
                    d = self._dev[k] = {"g": self._up(np.ones(x.c, np.float32) if g_ is None else g_.astype(np.float32)),
                                        "b": self._up(np.zeros(x.c, np.float32) if b_ is None else b_.astype(np.float32))}
                if d["g"].numel() != x.c:
                    raise UnsupportedOnnxGraph(f"{lay.name}: LayerNormalization scale has {d['g'].numel()} entries, the rows {x.c} channels")
                y = _Act(self.eng.op_layernorm(x.t, x.c, d["g"], d["b"], lay.attrs["epsilon"], split=self.split), x.c, x.flat, x.seq)
            elif op == "add":
                if len(ins) != 2 or lay.extra:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Add with a constant operand")
                if ins[0].t.shape != ins[1].t.shape:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Add of {tuple(ins[0].t.shape)} and {tuple(ins[1].t.shape)} (broadcasting is not built)")
                y = _Act(self.eng.op_add(ins[0].t, ins[1].t, split=self.split), ins[0].c, ins[0].flat, ins[0].seq)
            elif op == "mul":
                if len(ins) != 2 or lay.extra:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Mul with a constant operand")
                if ins[0].t.shape == ins[1].t.shape:
                    y = _Act(self.eng.op_mul(ins[0].t, ins[1].t, split=self.split), ins[0].c, ins[0].flat, ins[0].seq)
                else:
                    big, gate = (ins[0], ins[1]) if ins[0].t.shape[1] * ins[0].t.shape[2] >= ins[1].t.shape[1] * ins[1].t.shape[2] else (ins[1], ins[0])
                    if gate.t.shape[1] != 1 or gate.t.shape[2] != 1:
                        raise UnsupportedOnnxGraph(f"{lay.name}: Mul of two differently shaped feature maps (equal shapes and a per-channel gate [B, C, 1, 1] are built)")
                    y = _Act(self.eng.op_scale_channels(big.t, gate.t, split=self.split), big.c)
            elif op == "act":
                if lay.attrs["kind"] == "softmax":
                    x = ins[0]
                    if not (x.seq or x.flat) or lay.attrs.get("axis", -1) not in (-1, len(x.shape()) - 1):
                        raise UnsupportedOnnxGraph(f"{lay.name}: Softmax over axis {lay.attrs.get('axis')} of a {x.shape()} tensor (the channel axis of token rows is built)")
                    # a Softmax that IS a graph output (the probabilities a CTC decoder / classifier reads) stays fp32 [.., c]: bf16 probabilities have 8
                    # significant bits, so near-equal classes tie and the arg-max / confidence would differ from onnxruntime's fp32 ones
                    y = _Act(self.eng.op_softmax(x.t, x.c, f32=lay.outputs[0] in self.outputs, split=self.split), x.c, x.flat, x.seq)
                else:
                    y = self._post_act(lay, ins[0], lay.attrs["kind"])
            elif op == "resize":
                sc = lay.attrs.get("scale")
                sz = lay.attrs.get("sizes")
                if not sc and sz and len(sz) == 4:       # Resize given by target sizes (how some exporters write a x2 up-sampling)
                    hh, ww = ins[0].t.shape[1], ins[0].t.shape[2]
                    sc = [1.0, 1.0, sz[2] / hh, sz[3] / ww]
                if lay.attrs.get("mode", "nearest") != "nearest" or not sc or sc[0] != 1 or sc[1] != 1 or sc[2] != sc[3] or sc[2] != int(sc[2]):
                    raise UnsupportedOnnxGraph(f"{lay.name}: Resize {lay.attrs} (nearest, integer factor)")
                y = _Act(self.eng.op_upsample(ins[0].t, int(sc[2])), ins[0].c)
            elif op == "concat":
                if lay.attrs["axis"] != 1 or any(i.seq or i.flat for i in ins):
                    raise UnsupportedOnnxGraph(f"{lay.name}: Concat over axis {lay.attrs['axis']}")
                cc = sum(i.c for i in ins)
                out = self._zeros(ins[0].t.shape[:-1], _pad64(cc))
                o_ = 0
                for i in ins:
                    self._copy(i.t, out, i.c, dst_coff=o_)
                    o_ += i.c
                y = _Act(out, cc)
            elif op == "gemm":
                src = ins[0]
                if not src.seq and (src.t.shape[1] != 1 or src.t.shape[2] != 1):
                    raise UnsupportedOnnxGraph(f"{lay.name}: Gemm on a {tuple(src.t.shape)} feature map (token rows and pooled vectors are built)")
                lay2 = Layer("conv", lay.name, lay.inputs, lay.outputs, {}, weight=lay.weight.reshape(lay.weight.shape[0], -1, 1, 1), bias=lay.bias)
                d = self._conv_operands(k, lay2, self._cp(src), src.c)
                y = _Act(self.eng.op_conv2d(src.t, d["w"], d["b"], 1, 1, split=int(self.split)), d["n"], not src.seq, src.seq)
            else:
                raise UnsupportedOnnxGraph(f"{lay.name}: layer kind '{op}' has no executor")
            for o in lay.outputs:
                env[o] = y
        for name in self.outputs:
            if name not in env:
                raise UnsupportedOnnxGraph(f"graph output '{name}' was not produced")
        return [R(env[name], f"graph output '{name}'") for name in self.outputs]
