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

There is no CPU path: an operator outside this set raises ``UnsupportedOnnxGraph`` naming it (oracle/onnx_ref.py executes
graphs on the CPU for the tests only).  Supported today: Conv (groups 1: 1x1 / 3x3; depthwise: 3x3 / 5x5; stride 1 / 2,
"same" padding), ConvTranspose 2x2 / stride 2, BatchNormalization (folded), Relu / HardSwish / Sigmoid / HardSigmoid /
Relu6, a BatchNormalization that stands alone, Add, Mul by a per-channel gate, MaxPool(3, 2, 1) and k x k / stride k,
AveragePool k x k / stride k, GlobalAveragePool, Resize / Upsample (nearest, integer factor, by scales or sizes), Concat over
channels, Gemm / Flatten after a global pool.  Arithmetic is PT_PRECISION_BF16 (bf16
operands, fp32 accumulate); the hi/lo mode of the dedicated graphs is not wired here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import HipEngine
from .onnx_import import Layer, OnnxGraph, UnsupportedOnnxGraph, load_onnx
from .weights import tile_conv_weight

__all__ = ["HipGraphExecutor"]

_ACT_CODE = {None: 0, "relu": 1, "hardswish": 2}             # fused into the conv / depthwise epilogue
_ACT_KIND = {"relu": 1, "hardswish": 2, "sigmoid": 4, "hardsigmoid": 5, "relu6": 6}    # pt_op_act


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


@dataclass
class _Act:
    """an activation: bf16 NHWC tensor with padded channels, and its real channel count"""
    t: torch.Tensor
    c: int
    flat: bool = False      # the ONNX tensor is [B, C] (after Flatten / Gemm), held here as [B, 1, 1, C]


class HipGraphExecutor:
    def __init__(self, src, engine: Optional[HipEngine] = None, device: int = 0):
        self.graph: OnnxGraph = src if isinstance(src, OnnxGraph) else load_onnx(src)
        self.eng = engine or HipEngine(device)
        self.layers: List[Layer] = self.graph.layers()
        bad = [f"{l.name} ({l.attrs.get('onnx_op', l.op)})" for l in self.layers if l.op == "unsupported"]
        if bad:
            raise UnsupportedOnnxGraph("operators without an engine kernel: " + ", ".join(bad) + "\n" + self.graph.summary())
        self.inputs = [i for i in self.graph.model.inputs if i.name not in self.graph.init]
        self.outputs = [o.name for o in self.graph.model.outputs]
        if len(self.inputs) != 1:
            raise UnsupportedOnnxGraph(f"the executor takes graphs with one image input, this one has {[i.name for i in self.inputs]}")
        self._dev: Dict[int, Dict[str, torch.Tensor]] = {}      # layer index -> uploaded operands (filled on first use)

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
        d = {"cin_pad": cin_pad, "w": self._up(tile_conv_weight(wp).view(np.int16)), "b": bp.to(self.eng._tdev), "n": n}
        self._dev[k] = d
        return d

    # ---- layers ----------------------------------------------------------------------------------------------------
    def _conv(self, k: int, lay: Layer, x: _Act) -> _Act:
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
                d = self._dev[k] = {"w": self._up(to_bf16_bits(stem).reshape(64, 224).view(np.int16)), "b": bp.to(self.eng._tdev), "n": n0}
            if x.t.shape[1] % 2 or x.t.shape[2] % 2:
                raise UnsupportedOnnxGraph(f"{lay.name}: the 7x7 / stride-2 stem kernel needs even image sizes")
            return _Act(self.eng.op_stem7x7(x.t[..., :4].contiguous(), d["w"], d["b"]), d["n"])
        if a["group"] == 1:
            if kh not in (1, 3):
                raise UnsupportedOnnxGraph(f"{lay.name}: {kh}x{kw} convolution (the MFMA kernel covers 1x1 and 3x3)")
            d = self._conv_operands(k, lay, x.t.shape[-1], x.c)
            y = _Act(self.eng.op_conv2d(x.t, d["w"], d["b"], kh, sh, relu=fused), d["n"])
        elif a["group"] == x.c and lay.weight.shape[0] == x.c and lay.weight.shape[1] == 1:
            if kh not in (3, 5):
                raise UnsupportedOnnxGraph(f"{lay.name}: depthwise {kh}x{kw} (3x3 and 5x5 are built)")
            d = self._dev.get(k)
            if d is None:
                cp = x.t.shape[-1]
                wt = np.zeros((kh * kw, cp), np.float32)
                wt[:, :x.c] = lay.weight.reshape(x.c, kh * kw).T
                bt = np.zeros((cp,), np.float32)
                if lay.bias is not None:
                    bt[:x.c] = lay.bias
                d = self._dev[k] = {"w": self._up(wt), "b": self._up(bt)}
            y = _Act(self.eng.op_dwconv(x.t, d["w"], d["b"], kh, sh, fused), x.c)
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
            cop, cip = _pad64(co), x.t.shape[-1]
            wq = torch.zeros(2, 2, cop, cip)
            wq[:, :, :co, :ci] = w.permute(2, 3, 1, 0)              # N index = (dy * 2 + dx) * cop + co: the pixel-shuffle epilogue
            bq = torch.zeros(cop)
            if lay.bias is not None:
                bq[:co] = torch.from_numpy(lay.bias)
            d = self._dev[k] = {"w": self._up(tile_conv_weight(wq.reshape(4 * cop, cip, 1, 1)).view(np.int16)),
                                "b": bq.repeat(4).to(self.eng._tdev), "n": co, "cop": cop}
        fused = _ACT_CODE.get(lay.act, None) if lay.act in _ACT_CODE else 0
        post = None if lay.act in _ACT_CODE else lay.act
        y = _Act(self.eng.op_conv2d(x.t, d["w"], d["b"], 1, 1, relu=fused, shuffle_cout=d["cop"]), d["n"])
        return self._post_act(lay, y, post)

    def _post_act(self, lay: Layer, y: _Act, kind: Optional[str]) -> _Act:
        if kind is None:
            return y
        if kind not in _ACT_KIND:
            raise UnsupportedOnnxGraph(f"{lay.name}: activation '{kind}'")
        return _Act(self.eng.op_act(y.t, _ACT_KIND[kind], lay.attrs.get("act_alpha", 0.2), lay.attrs.get("act_beta", 0.5)), y.c, y.flat)

    @staticmethod
    def _repad(t: torch.Tensor, c: int) -> torch.Tensor:
        cp = _pad64(c)
        if t.shape[-1] == cp:
            return t.contiguous()
        out = torch.zeros(t.shape[:-1] + (cp,), dtype=t.dtype, device=t.device)
        out[..., :c] = t[..., :c]
        return out

    # ---- the graph -------------------------------------------------------------------------------------------------
    def run(self, x) -> List[np.ndarray]:
        """x: float NCHW image batch (numpy / torch) -> the graph outputs as float32 NCHW arrays, in graph order"""
        xt = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(torch.float32)
        if xt.ndim != 4:
            raise ValueError(f"expected an NCHW batch, got shape {tuple(xt.shape)}")
        outs = []
        for a in self.run_device(xt.permute(0, 2, 3, 1).to(self.eng._tdev).to(torch.bfloat16), xt.shape[1]):
            o = a.t[..., :a.c].float().permute(0, 3, 1, 2).contiguous().cpu().numpy()
            outs.append(o.reshape(o.shape[0], -1) if a.flat else o)
        return outs

    def run_device(self, nhwc: torch.Tensor, c: int) -> List[_Act]:
        """bf16 NHWC batch on the device whose first ``c`` channels are the image (what pt_det_preprocess / pt_cls_preprocess
        write) -> the graph outputs as device activations (bf16 NHWC, ``.t[..., :.c]`` are the real channels): no host trip"""
        cp = (c + 31) // 32 * 32                         # the image itself: 32 channels are enough for the first GEMM's K
        first = torch.zeros(nhwc.shape[:-1] + (cp,), dtype=torch.bfloat16, device=self.eng._tdev)
        first[..., :c] = nhwc[..., :c]
        env: Dict[str, _Act] = {self.inputs[0].name: _Act(first, c)}
        for k, lay in enumerate(self.layers):
            ins = [env[i] for i in lay.inputs if i in env]
            op = lay.op
            if op == "conv":
                y = self._conv(k, lay, ins[0])
            elif op == "convT":
                y = self._convT(k, lay, ins[0])
            elif op == "maxpool":
                a = lay.attrs
                kk, st, pd = a["kernel"], a["strides"], a["pads"]
                if kk[0] != kk[1] or st[0] != st[1] or len(set(pd)) != 1 or a.get("ceil_mode"):
                    raise UnsupportedOnnxGraph(f"{lay.name}: MaxPool {a}")
                y = _Act(self.eng.op_maxpool(ins[0].t, kk[0], st[0], pd[0]), ins[0].c)
            elif op == "avgpool":
                a = lay.attrs
                kk, st, pd = a["kernel"], a["strides"], a["pads"]
                if kk[0] != kk[1] or st != kk or any(pd) or a.get("ceil_mode") or ins[0].t.shape[1] % kk[0] or ins[0].t.shape[2] % kk[0]:
                    raise UnsupportedOnnxGraph(f"{lay.name}: AveragePool {a} (k x k / stride k without padding is built)")
                y = _Act(self.eng.op_avgpool(ins[0].t, kk[0]), ins[0].c)
            elif op == "bn":
                # a BatchNormalization that could not be folded into a convolution: per-channel affine = a depthwise 3x3 whose only
                # non-zero tap is the centre one
                d = self._dev.get(k)
                if d is None:
                    e_ = lay.extra
                    sc = e_["gamma"].astype(np.float64) / np.sqrt(e_["var"].astype(np.float64) + lay.attrs["epsilon"])
                    cp = ins[0].t.shape[-1]
                    wt = np.zeros((9, cp), np.float32)
                    wt[4, :ins[0].c] = sc
                    bt = np.zeros((cp,), np.float32)
                    bt[:ins[0].c] = e_["beta"].astype(np.float64) - e_["mean"].astype(np.float64) * sc
                    d = self._dev[k] = {"w": self._up(wt), "b": self._up(bt)}
                y = _Act(self.eng.op_dwconv(ins[0].t, d["w"], d["b"], 3, 1, 0), ins[0].c)
            elif op == "gap":
                y = _Act(self.eng.op_chan_mean(ins[0].t), ins[0].c)
            elif op == "add":
                if len(ins) != 2 or lay.extra:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Add with a constant operand")
                y = _Act(self.eng.op_add(ins[0].t, ins[1].t), ins[0].c)
            elif op == "mul":
                if len(ins) != 2 or lay.extra:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Mul with a constant operand")
                big, gate = (ins[0], ins[1]) if ins[0].t.shape[1] * ins[0].t.shape[2] >= ins[1].t.shape[1] * ins[1].t.shape[2] else (ins[1], ins[0])
                if gate.t.shape[1] != 1 or gate.t.shape[2] != 1:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Mul of two feature maps (a per-channel gate [B, C, 1, 1] is built)")
                y = _Act(self.eng.op_scale_channels(big.t, gate.t), big.c)
            elif op == "act":
                y = self._post_act(lay, ins[0], lay.attrs["kind"])
            elif op == "resize":
                sc = lay.attrs.get("scale")
                sz = lay.attrs.get("sizes")
                if not sc and sz and len(sz) == 4:       # Resize given by target sizes (how some exporters write a x2 up-sampling)
                    hh, ww = ins[0].t.shape[1], ins[0].t.shape[2]
                    sc = [1.0, 1.0, sz[2] / hh, sz[3] / ww]
                if lay.attrs.get("mode", "nearest") != "nearest" or not sc or sc[0] != 1 or sc[1] != 1 or sc[2] != sc[3] or sc[2] != int(sc[2]):
                    raise UnsupportedOnnxGraph(f"{lay.name}: Resize {lay.attrs} (nearest, integer factor)")
                f = int(sc[2])
                y = _Act(ins[0].t.repeat_interleave(f, dim=1).repeat_interleave(f, dim=2).contiguous(), ins[0].c)      # data movement only
            elif op == "concat":
                if lay.attrs["axis"] != 1:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Concat over axis {lay.attrs['axis']}")
                cat = torch.cat([i.t[..., :i.c] for i in ins], dim=-1)
                cc = sum(i.c for i in ins)
                y = _Act(self._repad(cat, cc), cc)
            elif op == "gemm":
                src = ins[0]
                if src.t.shape[1] != 1 or src.t.shape[2] != 1:
                    raise UnsupportedOnnxGraph(f"{lay.name}: Gemm on a {tuple(src.t.shape)} tensor (after a global pool only)")
                lay2 = Layer("conv", lay.name, lay.inputs, lay.outputs, {}, weight=lay.weight.reshape(lay.weight.shape[0], -1, 1, 1), bias=lay.bias)
                d = self._conv_operands(k, lay2, src.t.shape[-1], src.c)
                y = _Act(self.eng.op_conv2d(src.t, d["w"], d["b"], 1, 1), d["n"], True)
            elif op == "glue":
                if lay.attrs.get("onnx_op") in ("Flatten", "Reshape", "Squeeze", "Unsqueeze", "Identity") and ins:
                    y = ins[0]                               # [B, C, 1, 1] <-> [B, C]: the same NHWC tensor here
                    if y.t.shape[1] != 1 or y.t.shape[2] != 1:
                        raise UnsupportedOnnxGraph(f"{lay.name}: {lay.attrs.get('onnx_op')} of a feature map")
                    y = _Act(y.t, y.c, lay.attrs.get("onnx_op") in ("Flatten", "Squeeze", "Reshape"))
                elif not ins:
                    continue                                 # shape arithmetic on constants
                else:
                    raise UnsupportedOnnxGraph(f"{lay.name}: {lay.attrs.get('onnx_op')} has no executor")
            else:
                raise UnsupportedOnnxGraph(f"{lay.name}: layer kind '{op}' has no executor")
            for o in lay.outputs:
                env[o] = y
        for name in self.outputs:
            if name not in env:
                raise UnsupportedOnnxGraph(f"graph output '{name}' was not produced")
        return [env[name] for name in self.outputs]
