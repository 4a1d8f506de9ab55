"""ONNX graph importer: ``model.onnx`` -> engine layer list -> a checkpoint the HIP engine serves.

The reference resolves its default detection / recognition / layout models to ONNX files and runs them through
onnxruntime (``BaseInferTask._prepare_onnx_mode`` model/ocr_pdf/base_infer_task.py:139-144 -> ``DeployUtils.
prepare_onnx_model`` utils/deploy_utils.py:243-280; ``infer`` :366-370 calls ``predictor.run(None, input_dict)``).  Here:

  * ``pdf_table_amd.onnx_proto`` decodes the file without the ``onnx`` package;
  * ``OnnxGraph.layers()`` normalises the node list into the ENGINE LAYER LIST -- one record per kernel-sized unit:
    Conv / ConvTranspose with the following BatchNormalization folded (float64) and the following activation attached,
    pooling, Resize, Add, Concat, Mul, GlobalAveragePool, LSTM, Gemm / MatMul, shape glue -- and names every operator the
    engine has no kernel for (``unsupported_ops``);
  * ``recognise()`` matches the layer list against the network architectures the engine has launch graphs for
    (DB-ResNet18 ``DBModel``, ``CRNN``, PP-LCNet) and rebuilds the reference-layout ``state_dict``, which the ordinary
    packers (pdf_table_amd/weights.py) turn into an engine blob -- so an imported model and a ``.pt`` checkpoint of the same
    weights run through the SAME kernels and give the same numbers;
  * ``HipOnnxSession`` gives the imported detector the ``predictor.run(None, {"x": ...})`` surface.

A graph of any other architecture is executed layer by layer by ``pdf_table_amd.onnx_exec.HipGraphExecutor`` -- one engine
call per layer of the list -- as long as it is made of the operators that executor lists (convolutional networks:
detection / layout / classification backbones, necks and heads); ``recognise`` still raises ``UnsupportedOnnxGraph`` with
the layer inventory for such a graph, and the executor names the first operator it has no kernel for (sequence models:
attention / LayerNorm blocks of SVTR-type recognisers).  There is no CPU execution path in the product (oracle/onnx_ref.py
executes graphs on the CPU for the tests only).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .onnx_proto import OnnxModel, OnnxNode, parse_model

__all__ = ["OnnxGraph", "Layer", "UnsupportedOnnxGraph", "load_onnx", "recognise", "HipOnnxSession", "ENGINE_OPS"]

# operators the engine has kernels for (conv_igemm / dwconv / stem / pool / epilogue fusions / lstm / gemm rows ...)
ENGINE_OPS = {"Conv", "ConvTranspose", "BatchNormalization", "Relu", "Sigmoid", "HardSwish", "HardSigmoid", "PRelu", "Clip",
              "MaxPool", "AveragePool", "GlobalAveragePool", "ReduceMean", "Add", "Mul", "Concat", "Resize", "Upsample",
              "LSTM", "Gemm", "MatMul", "Softmax", "Flatten", "Reshape", "Transpose", "Squeeze", "Unsqueeze", "Identity",
              "Constant", "ConstantOfShape", "Expand", "Shape", "Gather", "Slice", "Cast", "Div", "Sub",
              # sequence blocks (SVTR-type recognisers): native nodes, and the pieces LayerNorm / GELU decompose into below opset 17 / 20
              "LayerNormalization", "Gelu", "Erf", "Pow", "Sqrt", "Split"}
_ACTS = {"Relu": "relu", "HardSwish": "hardswish", "Sigmoid": "sigmoid", "HardSigmoid": "hardsigmoid", "PRelu": "prelu"}


class UnsupportedOnnxGraph(RuntimeError):
    pass


@dataclass
class Layer:
    """one entry of the engine layer list"""
    op: str                               # conv | convT | maxpool | avgpool | gap | resize | add | mul | concat | lstm | gemm | act | glue
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    weight: Optional[np.ndarray] = None    # conv: [Cout, Cin/g, kh, kw]; convT: [Cin, Cout/g, kh, kw]; gemm: [out, in]
    bias: Optional[np.ndarray] = None
    act: Optional[str] = None
    bn_folded: bool = False
    extra: Dict[str, np.ndarray] = field(default_factory=dict)    # LSTM: W, R, B ; prelu: slope

    def describe(self) -> str:
        w = "" if self.weight is None else f" w{list(self.weight.shape)}"
        a = f" +{self.act}" if self.act else ""
        at = {k: v for k, v in self.attrs.items() if k in ("kernel", "strides", "pads", "group", "scale", "axis")}
        return f"{self.op}{w}{' +bn' if self.bn_folded else ''}{a} {at if at else ''}".rstrip()


class OnnxGraph:
    def __init__(self, model: OnnxModel):
        self.model = model
        self.init: Dict[str, np.ndarray] = dict(model.initializers)
        self.nodes: List[OnnxNode] = []
        for n in model.nodes:                      # constants become initializers; identities are short-circuited
            if n.op_type == "Constant":
                v = n.attrs.get("value")
                if v is None:
                    v = np.asarray(n.attrs.get("value_float", n.attrs.get("value_int", n.attrs.get("value_floats", n.attrs.get("value_ints")))))
                self.init[n.outputs[0]] = np.asarray(v)
            else:
                self.nodes.append(n)
        alias: Dict[str, str] = {}
        kept = []
        for n in self.nodes:
            n.inputs = [alias.get(i, i) for i in n.inputs]
            if n.op_type == "Identity":
                if n.inputs[0] in self.init:
                    self.init[n.outputs[0]] = self.init[n.inputs[0]]
                else:
                    alias[n.outputs[0]] = n.inputs[0]
            else:
                kept.append(n)
        self.nodes = kept
        self.consumers: Dict[str, List[int]] = {}
        for k, n in enumerate(self.nodes):
            for i in n.inputs:
                self.consumers.setdefault(i, []).append(k)
        self.graph_outputs = [alias.get(o.name, o.name) for o in model.outputs]

    # ---- inventory --------------------------------------------------------------------------------------------------
    def op_histogram(self) -> Dict[str, int]:
        h: Dict[str, int] = {}
        for n in self.nodes:
            h[n.op_type] = h.get(n.op_type, 0) + 1
        return h

    def unsupported_ops(self) -> List[str]:
        return sorted({n.op_type for n in self.nodes if n.op_type not in ENGINE_OPS})

    def _sole_consumer(self, tensor: str, op_types: Sequence[str]) -> Optional[int]:
        c = self.consumers.get(tensor, [])
        if len(c) == 1 and tensor not in self.graph_outputs and self.nodes[c[0]].op_type in op_types:
            return c[0]
        return None

    def _hardswish_pair(self, tensor: str):
        """tensor -> HardSigmoid(alpha 1/6, beta 0.5) -> Mul(tensor, .) and nothing else: (hsig node, mul node, output)"""
        c = self.consumers.get(tensor, [])
        if len(c) != 2 or tensor in self.graph_outputs:
            return None
        a, b = (self.nodes[c[0]], self.nodes[c[1]])
        if a.op_type != "HardSigmoid" or b.op_type != "Mul" or sorted(b.inputs) != sorted([tensor, a.outputs[0]]):
            return None
        if abs(float(a.attrs.get("alpha", 0.2)) - 1.0 / 6.0) > 1e-6 or abs(float(a.attrs.get("beta", 0.5)) - 0.5) > 1e-6:
            return None
        if len(self.consumers.get(a.outputs[0], [])) != 1:
            return None
        return c[0], c[1], b.outputs[0]

    def _const_scalar(self, name: str) -> Optional[float]:
        v = self.init.get(name)
        return float(np.asarray(v).reshape(-1)[0]) if v is not None and np.asarray(v).size == 1 else None

    def _only(self, tensor: str, op: str) -> Optional[int]:
        """index of the ONLY consumer of `tensor` if it is an `op` node"""
        return self._sole_consumer(tensor, (op,))

    def _match_layernorm(self, k: int):
        """ReduceMean(x, -1) -> Sub(x, .) -> Pow(., 2) -> ReduceMean(-1) -> Add(eps) -> Sqrt -> Div(sub, .) -> Mul(gamma) -> Add(beta): how
        nn.LayerNorm exports below opset 17 (and from Paddle).  -> (used node indices, x, gamma, beta, eps, output) or None"""
        n = self.nodes[k]
        ax = n.attrs.get("axes")
        if ax is None and len(n.inputs) > 1:
            ax = self.init.get(n.inputs[1])
        if n.op_type != "ReduceMean" or ax is None or [int(a) for a in np.asarray(ax).reshape(-1)] != [-1]:
            return None
        x = n.inputs[0]
        j_sub = self._only(n.outputs[0], "Sub")
        if j_sub is None or self.nodes[j_sub].inputs[0] != x:
            return None
        d = self.nodes[j_sub].outputs[0]
        cons = self.consumers.get(d, [])
        if len(cons) != 2:
            return None
        j_pow = next((c for c in cons if self.nodes[c].op_type in ("Pow", "Mul")), None)
        j_div = next((c for c in cons if self.nodes[c].op_type == "Div"), None)
        if j_pow is None or j_div is None:
            return None
        pw = self.nodes[j_pow]
        if pw.op_type == "Pow" and self._const_scalar(pw.inputs[1]) != 2.0:
            return None
        if pw.op_type == "Mul" and list(pw.inputs) != [d, d]:
            return None
        j_m2 = self._only(pw.outputs[0], "ReduceMean")
        if j_m2 is None:
            return None
        j_add = self._only(self.nodes[j_m2].outputs[0], "Add")
        if j_add is None:
            return None
        eps = next((self._const_scalar(i) for i in self.nodes[j_add].inputs if i in self.init), None)
        j_sqrt = self._only(self.nodes[j_add].outputs[0], "Sqrt")
        if eps is None or j_sqrt is None or self.nodes[j_div].inputs != [d, self.nodes[j_sqrt].outputs[0]]:
            return None
        used = [k, j_sub, j_pow, j_m2, j_add, j_sqrt, j_div]
        cur = self.nodes[j_div].outputs[0]
        gamma = beta = None
        j_mul = self._only(cur, "Mul")
        if j_mul is not None:
            g = [i for i in self.nodes[j_mul].inputs if i in self.init]
            if len(g) == 1 and self.init[g[0]].ndim == 1:
                gamma, cur = np.asarray(self.init[g[0]], np.float32), self.nodes[j_mul].outputs[0]
                used.append(j_mul)
                j_b = self._only(cur, "Add")
                if j_b is not None:
                    bb = [i for i in self.nodes[j_b].inputs if i in self.init]
                    if len(bb) == 1 and self.init[bb[0]].ndim == 1:
                        beta, cur = np.asarray(self.init[bb[0]], np.float32), self.nodes[j_b].outputs[0]
                        used.append(j_b)
        return used, x, gamma, beta, eps, cur

    def _match_gelu(self, k: int):
        """Div(x, sqrt 2) [or Mul(x, 1 / sqrt 2)] -> Erf -> Add(1) -> Mul(x, .) -> Mul(0.5): nn.GELU() below opset 20"""
        n = self.nodes[k]
        if n.op_type not in ("Div", "Mul") or len(n.inputs) != 2:
            return None
        c = self._const_scalar(n.inputs[1])
        x = n.inputs[0]
        if c is None or abs((c if n.op_type == "Div" else 1.0 / c) - 2.0 ** 0.5) > 1e-3:
            return None
        j_erf = self._only(n.outputs[0], "Erf")
        if j_erf is None:
            return None
        j_add = self._only(self.nodes[j_erf].outputs[0], "Add")
        if j_add is None or next((self._const_scalar(i) for i in self.nodes[j_add].inputs if i in self.init), None) != 1.0:
            return None
        j_mx = self._only(self.nodes[j_add].outputs[0], "Mul")
        if j_mx is None or sorted(self.nodes[j_mx].inputs) != sorted([x, self.nodes[j_add].outputs[0]]):
            return None
        j_half = self._only(self.nodes[j_mx].outputs[0], "Mul")
        if j_half is None or next((self._const_scalar(i) for i in self.nodes[j_half].inputs if i in self.init), None) != 0.5:
            return None
        return [k, j_erf, j_add, j_mx, j_half], x, self.nodes[j_half].outputs[0]

    def _match_swish(self, k: int):
        """Sigmoid(x) -> Mul(x, .): nn.SiLU / PaddleOCR's swish"""
        n = self.nodes[k]
        j = self._only(n.outputs[0], "Mul") if n.op_type == "Sigmoid" else None
        if j is None or sorted(self.nodes[j].inputs) != sorted([n.inputs[0], n.outputs[0]]):
            return None
        return [k, j], n.inputs[0], self.nodes[j].outputs[0]

    # ---- the engine layer list ----------------------------------------------------------------------------------------
    def layers(self) -> List[Layer]:
        out: List[Layer] = []
        used = set()
        for k, n in enumerate(self.nodes):
            if k in used:
                continue
            t = n.op_type
            m = self._match_layernorm(k) if t == "ReduceMean" else None
            if m is not None:
                used.update(m[0])
                out.append(Layer("layernorm", n.name or m[5], [m[1]], [m[5]], {"epsilon": float(m[4])}, extra={"gamma": m[2], "beta": m[3]}))
                continue
            if t == "LayerNormalization":
                if int(n.attrs.get("axis", -1)) != -1:
                    raise UnsupportedOnnxGraph(f"LayerNormalization '{n.name}' over axis {n.attrs.get('axis')} (the last axis is built)")
                out.append(Layer("layernorm", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]], {"epsilon": float(n.attrs.get("epsilon", 1e-5))},
                                 extra={"gamma": np.asarray(self.init[n.inputs[1]], np.float32),
                                        "beta": np.asarray(self.init[n.inputs[2]], np.float32) if len(n.inputs) > 2 and n.inputs[2] in self.init else None}))
                continue
            m = self._match_gelu(k) if t in ("Div", "Mul") else (self._match_swish(k) if t == "Sigmoid" else None)
            if m is not None:
                used.update(m[0])
                out.append(Layer("act", n.name or m[2], [m[1]], [m[2]], {"kind": "gelu" if t != "Sigmoid" else "swish", "axis": -1}))
                continue
            if t == "Gelu":
                out.append(Layer("act", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]], {"kind": "gelu", "axis": -1}))
                continue
            if t in ("Conv", "ConvTranspose"):
                w = self.init.get(n.inputs[1])
                if w is None:
                    raise UnsupportedOnnxGraph(f"{t} '{n.name}': weights are not an initializer (dynamic weights)")
                w = np.asarray(w, np.float32) if w.dtype != np.float32 else w
                cout = w.shape[0] if t == "Conv" else w.shape[1] * int(n.attrs.get("group", 1))
                b = self.init.get(n.inputs[2]) if len(n.inputs) > 2 and n.inputs[2] else None
                lay = Layer("conv" if t == "Conv" else "convT", n.name or n.outputs[0], [n.inputs[0]], list(n.outputs),
                            {"kernel": list(n.attrs.get("kernel_shape", w.shape[2:])), "strides": list(n.attrs.get("strides", [1, 1])),
                             "pads": list(n.attrs.get("pads", [0, 0, 0, 0])), "group": int(n.attrs.get("group", 1)),
                             "dilations": list(n.attrs.get("dilations", [1, 1]))},
                            weight=w, bias=None if b is None else np.asarray(b, np.float32))
                cur = n.outputs[0]
                j = self._sole_consumer(cur, ("BatchNormalization",))
                if j is not None:                   # fold the BN into the conv: float64, like the weight packers do
                    bn = self.nodes[j]
                    g, be, mu, var = (np.asarray(self.init[x], np.float64) for x in bn.inputs[1:5])
                    s = g / np.sqrt(var + float(bn.attrs.get("epsilon", 1e-5)))
                    shp = (-1, 1, 1, 1) if t == "Conv" else (1, -1, 1, 1)
                    lay.weight = (lay.weight.astype(np.float64) * s.reshape(shp)).astype(np.float32)
                    b0 = np.zeros(cout) if lay.bias is None else lay.bias.astype(np.float64)
                    lay.bias = ((b0 - mu) * s + be).astype(np.float32)
                    lay.bn_folded = True
                    used.add(j)
                    cur = bn.outputs[0]
                hs = self._hardswish_pair(cur)
                if hs is not None:                  # x * HardSigmoid(x): how hardswish exports below opset 14 (and from Paddle)
                    lay.act = "hardswish"
                    used.update(hs[:2])
                    cur = hs[2]
                j = None if hs is not None else self._sole_consumer(cur, tuple(_ACTS) + ("Clip",))
                if j is not None:
                    a = self.nodes[j]
                    if a.op_type == "Clip":
                        lo = float(self.init[a.inputs[1]]) if len(a.inputs) > 1 and a.inputs[1] in self.init else float(a.attrs.get("min", 0.0))
                        hi = float(self.init[a.inputs[2]]) if len(a.inputs) > 2 and a.inputs[2] in self.init else float(a.attrs.get("max", 6.0))
                        lay.act = "relu6" if (lo, hi) == (0.0, 6.0) else f"clip({lo},{hi})"
                    else:
                        lay.act = _ACTS[a.op_type]
                        if a.op_type == "HardSigmoid":      # ONNX defaults: alpha 0.2, beta 0.5 (torch exports 1/6, 0.5)
                            lay.attrs["act_alpha"] = float(a.attrs.get("alpha", 0.2))
                            lay.attrs["act_beta"] = float(a.attrs.get("beta", 0.5))
                        if a.op_type == "PRelu":
                            lay.extra["slope"] = np.asarray(self.init[a.inputs[1]], np.float32)
                    used.add(j)
                    cur = a.outputs[0]
                lay.outputs = [cur]
                out.append(lay)
            elif t == "BatchNormalization":
                out.append(Layer("bn", n.name or n.outputs[0], [n.inputs[0]], list(n.outputs), {"epsilon": float(n.attrs.get("epsilon", 1e-5))},
                                 extra={k2: np.asarray(self.init[x], np.float32) for k2, x in zip(("gamma", "beta", "mean", "var"), n.inputs[1:5])}))
            elif t in ("MaxPool", "AveragePool"):
                out.append(Layer("maxpool" if t == "MaxPool" else "avgpool", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]],
                                 {"kernel": list(n.attrs.get("kernel_shape", [])), "strides": list(n.attrs.get("strides", [1, 1])),
                                  "pads": list(n.attrs.get("pads", [0, 0, 0, 0])), "ceil_mode": int(n.attrs.get("ceil_mode", 0))}))
            elif t == "GlobalAveragePool" or (t == "ReduceMean" and sorted(n.attrs.get("axes", [])) in ([2, 3], [-2, -1])):
                out.append(Layer("gap", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]]))
            elif t in ("Resize", "Upsample"):
                scale = sizes = None
                for i in n.inputs[1:]:
                    v = self.init.get(i)
                    if v is not None and v.size == 4 and v.dtype.kind == "f":
                        scale = [float(x) for x in np.asarray(v).reshape(-1)]
                    elif v is not None and v.size == 4 and v.dtype.kind in "iu":
                        sizes = [int(x) for x in np.asarray(v).reshape(-1)]
                out.append(Layer("resize", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]],
                                 {"mode": n.attrs.get("mode", "nearest"), "scale": scale, "sizes": sizes,
                                  "coordinate_transformation_mode": n.attrs.get("coordinate_transformation_mode", "")}))
            elif t in ("Add", "Mul", "Sub", "Div"):
                consts = {i: self.init[i] for i in n.inputs if i in self.init}
                out.append(Layer(t.lower(), n.name or n.outputs[0], [i for i in n.inputs if i not in consts], [n.outputs[0]], {"all_inputs": list(n.inputs)},
                                 extra={f"const{q}": np.asarray(v) for q, v in enumerate(consts.values())}))
            elif t == "Concat":
                out.append(Layer("concat", n.name or n.outputs[0], list(n.inputs), [n.outputs[0]], {"axis": int(n.attrs.get("axis", 1))}))
            elif t in _ACTS or t in ("Softmax", "Clip"):
                out.append(Layer("act", n.name or n.outputs[0], [n.inputs[0]], [n.outputs[0]],
                                 {"kind": _ACTS.get(t, t.lower()), "axis": int(n.attrs.get("axis", -1)),
                                  "act_alpha": float(n.attrs.get("alpha", 0.2)), "act_beta": float(n.attrs.get("beta", 0.5))}))
            elif t == "LSTM":
                out.append(Layer("lstm", n.name or n.outputs[0], [n.inputs[0]], [o for o in n.outputs if o],
                                 {"hidden_size": int(n.attrs["hidden_size"]), "direction": n.attrs.get("direction", "forward")},
                                 extra={"W": np.asarray(self.init[n.inputs[1]], np.float32), "R": np.asarray(self.init[n.inputs[2]], np.float32),
                                        "B": np.asarray(self.init[n.inputs[3]], np.float32) if len(n.inputs) > 3 and n.inputs[3] in self.init else None}))
            elif t in ("Gemm", "MatMul"):
                wname = next((i for i in n.inputs if i in self.init and self.init[i].ndim == 2), None)
                if wname is None:
                    out.append(Layer("matmul", n.name or n.outputs[0], list(n.inputs), [n.outputs[0]], {"all_inputs": list(n.inputs)}))
                    continue
                w = np.asarray(self.init[wname], np.float32)
                if t == "MatMul" or not int(n.attrs.get("transB", 0)):
                    w = w.T                                    # -> [out, in] like nn.Linear.weight
                b = next((np.asarray(self.init[i], np.float32) for i in n.inputs[2:] if i in self.init), None)
                lay = Layer("gemm", n.name or n.outputs[0], [i for i in n.inputs if i not in self.init], [n.outputs[0]], weight=np.ascontiguousarray(w), bias=b)
                if t == "MatMul":                              # MatMul + Add(bias) is how nn.Linear on 3-D input exports
                    j = self._sole_consumer(n.outputs[0], ("Add",))
                    if j is not None:
                        cb = [i for i in self.nodes[j].inputs if i in self.init]
                        if len(cb) == 1 and self.init[cb[0]].ndim == 1:
                            lay.bias = np.asarray(self.init[cb[0]], np.float32)
                            lay.outputs = [self.nodes[j].outputs[0]]
                            used.add(j)
                out.append(lay)
            else:
                out.append(Layer("glue" if t in ENGINE_OPS else "unsupported", n.name or n.outputs[0], [i for i in n.inputs if i not in self.init],
                                 list(n.outputs), {"onnx_op": t, "all_inputs": list(n.inputs), "node_attrs": dict(n.attrs)}))
        return out

    def summary(self) -> str:
        lines = [f"ONNX graph '{self.model.graph_name}' (opset {self.model.opset}, producer '{self.model.producer}'): "
                 f"{len(self.nodes)} nodes, {len(self.init)} initializers, "
                 f"{sum(int(v.size) for v in self.init.values()) / 1e6:.2f} M parameters"]
        lines.append("operators: " + ", ".join(f"{k} x{v}" for k, v in sorted(self.op_histogram().items())))
        u = self.unsupported_ops()
        lines.append("operators without an engine kernel: " + (", ".join(u) if u else "none"))
        return "\n".join(lines)


def load_onnx(src: Union[str, bytes, os.PathLike]) -> OnnxGraph:
    """a path to model.onnx, a directory holding fp16_model.onnx / model.onnx (prepare_onnx_model's layout,
    deploy_utils.py:257-263), or the bytes themselves"""
    if isinstance(src, (bytes, bytearray, memoryview)):
        data = bytes(src)
    else:
        p = os.fspath(src)
        if os.path.isdir(p):
            for cand in ("model.onnx", "fp16_model.onnx", "inference.onnx"):      # the fp32 graph first: the engine makes its own precision
                if os.path.exists(os.path.join(p, cand)):
                    p = os.path.join(p, cand)
                    break
            else:
                raise FileNotFoundError(f"no model.onnx under {p}")
        with open(p, "rb") as f:
            data = f.read()
    return OnnxGraph(parse_model(data))


# ---------------------------------------------------------------------------------------------------------------------
# architecture recognisers: engine layer list -> reference-layout state_dict
# ---------------------------------------------------------------------------------------------------------------------
def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _put_conv_bn(sd, conv: str, bn: Optional[str], lay: Layer, conv_has_bias: bool):
    """a conv whose BN the exporter (or layers()) folded away: the folded weights go to conv.weight and the BN of the
    reference layout becomes an affine identity carrying the folded bias -- the packers' own fold then reproduces W, b"""
    c = lay.weight.shape[0] if lay.op == "conv" else lay.weight.shape[1]
    sd[conv + ".weight"] = _t(lay.weight)
    b = lay.bias if lay.bias is not None else np.zeros(c, np.float32)
    if bn is None:
        if conv_has_bias:
            sd[conv + ".bias"] = _t(b)
        elif np.abs(b).max() > 0:
            raise UnsupportedOnnxGraph(f"{conv}: the reference layer has no bias but the graph's has one")
        return
    if conv_has_bias:
        sd[conv + ".bias"] = _t(np.zeros(c, np.float32))
    sd[bn + ".weight"] = torch.ones(c)
    sd[bn + ".bias"] = _t(b)
    sd[bn + ".running_mean"] = torch.zeros(c)
    sd[bn + ".running_var"] = torch.full((c,), 1.0 - 1e-5)     # sqrt(var + eps) == 1 in the packers' float64 fold
    sd[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _expect(lay: Layer, op: str, wshape: Tuple[int, ...], what: str):
    if lay.op != op or lay.weight is None or tuple(lay.weight.shape) != tuple(wshape):
        got = f"{lay.op} {None if lay.weight is None else list(lay.weight.shape)}"
        raise UnsupportedOnnxGraph(f"not this architecture: {what} should be {op} {list(wshape)}, the graph has {got}")


def _as_db_resnet18(layers: List[Layer]) -> Dict[str, torch.Tensor]:
    """DBModel = ResNet-18 + SegDetector in forward order (db_net/dbnet.py:324-335 backbone, :140-172 BasicBlock: conv1,
    conv2, then the downsample branch, :615-638 decoder: in5..in2, out5..out2, binarize)"""
    convs = [l for l in layers if l.op in ("conv", "convT")]
    if len(convs) != 31:
        raise UnsupportedOnnxGraph(f"not DB-ResNet18: {len(convs)} convolutions instead of 31")
    sd: Dict[str, torch.Tensor] = {}
    it = iter(convs)
    lay = next(it)
    _expect(lay, "conv", (64, 3, 7, 7), "backbone.conv1")
    _put_conv_bn(sd, "backbone.conv1", "backbone.bn1", lay, False)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            l1, l2 = next(it), next(it)
            _expect(l1, "conv", (planes, inpl, 3, 3), p + ".conv1")
            _expect(l2, "conv", (planes, planes, 3, 3), p + ".conv2")
            _put_conv_bn(sd, p + ".conv1", p + ".bn1", l1, False)
            _put_conv_bn(sd, p + ".conv2", p + ".bn2", l2, False)
            if bi == 0 and li > 1:
                ld = next(it)
                _expect(ld, "conv", (planes, inpl, 1, 1), p + ".downsample.0")
                _put_conv_bn(sd, p + ".downsample.0", p + ".downsample.1", ld, False)
            inpl = planes
    for name, cin in (("in5", 512), ("in4", 256), ("in3", 128), ("in2", 64)):
        lay = next(it)
        _expect(lay, "conv", (256, cin, 1, 1), "decoder." + name)
        _put_conv_bn(sd, "decoder." + name, None, lay, False)
    for name in ("out5.0", "out4.0", "out3.0", "out2"):
        lay = next(it)
        _expect(lay, "conv", (64, 256, 3, 3), "decoder." + name)
        _put_conv_bn(sd, "decoder." + name, None, lay, False)
    lay = next(it)
    _expect(lay, "conv", (64, 256, 3, 3), "decoder.binarize.0")
    _put_conv_bn(sd, "decoder.binarize.0", "decoder.binarize.1", lay, False)
    lay = next(it)
    _expect(lay, "convT", (64, 64, 2, 2), "decoder.binarize.3")
    _put_conv_bn(sd, "decoder.binarize.3", "decoder.binarize.4", lay, True)
    lay = next(it)
    _expect(lay, "convT", (64, 1, 2, 2), "decoder.binarize.6")
    if lay.act != "sigmoid":
        raise UnsupportedOnnxGraph("not DB-ResNet18: the last transposed conv is not followed by a Sigmoid")
    _put_conv_bn(sd, "decoder.binarize.6", None, lay, True)
    return sd


def _as_crnn(layers: List[Layer]) -> Dict[str, torch.Tensor]:
    """CRNN (crnn/modeling_crnn.py:40-113): 7 convs (+BN, ReLU), two BiLSTM + Linear, classifier without bias.
    ONNX LSTM tensors: W [dirs, 4H, in], R [dirs, 4H, H], B [dirs, 8H] with gate order i, o, f, c; torch keeps i, f, g, o."""
    convs = [l for l in layers if l.op == "conv"]
    lstms = [l for l in layers if l.op == "lstm"]
    gemms = [l for l in layers if l.op == "gemm"]
    if len(convs) != 7 or len(lstms) != 2 or len(gemms) != 3:
        raise UnsupportedOnnxGraph(f"not CRNN: {len(convs)} convs / {len(lstms)} LSTMs / {len(gemms)} linear layers instead of 7 / 2 / 3")
    sd: Dict[str, torch.Tensor] = {}
    spec = [("conv0.0", "conv0.1", 64, 1, 3, 3), ("conv1.0", "conv1.1", 128, 64, 3, 3), ("conv2.0", "conv2.1", 256, 128, 3, 3),
            ("conv2.3", "conv2.4", 256, 256, 3, 3), ("conv3.0", "conv3.1", 512, 256, 3, 3), ("conv3.3", "conv3.4", 512, 512, 3, 3),
            ("conv4.0", "conv4.1", 512, 512, 2, 1)]
    for lay, (c, bn, co, ci, kh, kw) in zip(convs, spec):
        _expect(lay, "conv", (co, ci, kh, kw), c)
        _put_conv_bn(sd, c, bn, lay, True)

    def regate(a, H):      # ONNX i, o, f, c  ->  torch i, f, g(c), o   along the 4H axis
        i, o, f, c = a[0 * H:1 * H], a[1 * H:2 * H], a[2 * H:3 * H], a[3 * H:4 * H]
        return np.concatenate([i, f, c, o], 0)

    for k, (lay, lin) in enumerate(zip(lstms, gemms[:2])):
        H = int(lay.attrs["hidden_size"])
        if lay.attrs.get("direction") != "bidirectional" or H != 256:
            raise UnsupportedOnnxGraph("not CRNN: LSTM is not bidirectional with 256 hidden units")
        W, R, B = lay.extra["W"], lay.extra["R"], lay.extra["B"]
        for d, sfx in enumerate(("", "_reverse")):
            sd[f"rnn.{k}.rnn.weight_ih_l0{sfx}"] = _t(regate(W[d], H))
            sd[f"rnn.{k}.rnn.weight_hh_l0{sfx}"] = _t(regate(R[d], H))
            b = np.zeros(8 * H, np.float32) if B is None else B[d]
            sd[f"rnn.{k}.rnn.bias_ih_l0{sfx}"] = _t(regate(b[:4 * H], H))
            sd[f"rnn.{k}.rnn.bias_hh_l0{sfx}"] = _t(regate(b[4 * H:], H))
        sd[f"rnn.{k}.embedding.weight"] = _t(lin.weight)
        sd[f"rnn.{k}.embedding.bias"] = _t(lin.bias if lin.bias is not None else np.zeros(lin.weight.shape[0], np.float32))
    if gemms[2].bias is not None and np.abs(gemms[2].bias).max() > 0:
        raise UnsupportedOnnxGraph("not CRNN: the classifier has a bias")
    sd["cls.weight"] = _t(gemms[2].weight)
    return sd


_RECOGNISERS = (("db_resnet18", _as_db_resnet18), ("crnn", _as_crnn))


def recognise(graph: OnnxGraph) -> Tuple[str, Dict[str, torch.Tensor]]:
    """-> (architecture name, reference-layout state_dict) or UnsupportedOnnxGraph naming what the graph is made of"""
    layers = graph.layers()
    errors = []
    for name, fn in _RECOGNISERS:
        try:
            return name, fn(layers)
        except UnsupportedOnnxGraph as e:
            errors.append(f"  {name}: {e}")
    inv: Dict[str, int] = {}
    for l in layers:
        d = l.describe()
        inv[d] = inv.get(d, 0) + 1
    top = sorted(inv.items(), key=lambda kv: -kv[1])[:12]
    raise UnsupportedOnnxGraph(
        "the graph parses, but it is none of the architectures the engine has a launch graph for:\n" + "\n".join(errors)
        + "\n" + graph.summary() + "\nlayer list (most frequent): " + "; ".join(f"{k} x{v}" for k, v in top))


# ---------------------------------------------------------------------------------------------------------------------
# predictor.run(None, input_dict) for an imported detector
# ---------------------------------------------------------------------------------------------------------------------
class _IO:
    def __init__(self, name, shape):
        self.name, self.shape = name, shape


class HipOnnxSession:
    """the ``ort.InferenceSession`` surface the reference's ``infer`` uses (``run`` / ``get_inputs`` / ``get_providers``) over
    the HIP engine: a graph ``recognise`` maps to DB-ResNet18 runs on that launch graph (input float [B, 3, H, W] with H, W
    multiples of 32 -> probability map [B, 1, H, W]); any other convolutional graph runs layer by layer through
    ``pdf_table_amd.onnx_exec.HipGraphExecutor`` (``arch == "generic"``)."""

    def __init__(self, src, engine=None, device: int = 0):
        from . import lib as L
        from .engine import HipEngine
        from .weights import pack_db_resnet18
        self.graph = load_onnx(src)
        self._L = L
        self._exec = None
        try:
            self.arch, self.state_dict = recognise(self.graph)
        except UnsupportedOnnxGraph:
            self.arch, self.state_dict = "generic", None
        if self.arch == "db_resnet18":
            self.engine = engine or HipEngine(device)
            self.engine.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(self.state_dict, fmt=self.engine.weight_fmt))
            return
        if self.arch != "generic" and any(l.op == "lstm" for l in self.graph.layers()):
            raise UnsupportedOnnxGraph(f"'{self.arch}' imports as weights (see the task classes) but has no session surface: its "
                                       "engine path does not materialise the logits the ONNX graph returns")
        # no dedicated launch graph: the layer list runs operator by operator (convolutional graphs; anything else is named
        # by the executor when it is reached -- there is no CPU path)
        from .onnx_exec import HipGraphExecutor
        self.arch = "generic"
        self.engine = engine or HipEngine(device)
        self._exec = HipGraphExecutor(self.graph, engine=self.engine)

    def get_providers(self):
        return ["HipExecutionProvider"]

    def get_inputs(self):
        return [_IO(i.name, list(i.shape)) for i in self.graph.model.inputs]

    def get_outputs(self):
        return [_IO(o.name, list(o.shape)) for o in self.graph.model.outputs]

    def run(self, output_names, input_dict):
        (x,) = [np.asarray(v) for v in input_dict.values()]
        if self._exec is not None:
            outs = self._exec.run(x.astype(np.float32))
            if output_names:
                outs = [outs[self._exec.outputs.index(o)] for o in output_names]
            return [o.astype(np.float16) for o in outs] if x.dtype == np.float16 else outs
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"input {x.shape}: expected [B, 3, H, W] with H, W multiples of 32")
        dev = self.engine._tdev
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).permute(0, 2, 3, 1)
        n, h, w, _ = t.shape
        adt = self.engine.act_dtype                  # torch.float16 under PT_PRECISION_F16, else torch.bfloat16
        if self.engine.split:
            hi = t.to(adt)
            lo = (t - hi.float()).to(adt)
            x8 = torch.zeros((n, h, w, 8), dtype=adt, device=dev)
            x8[..., :3], x8[..., 4:7] = hi, lo
            xin = x8
        else:
            xin = torch.zeros((n, h, w, 4), dtype=adt, device=dev)
            xin[..., :3] = t.to(adt)
        prob = self.engine.det_forward_net(xin.contiguous())
        out = prob.cpu().numpy()[:, None]
        return [out.astype(x.dtype) if x.dtype == np.float16 else out]
