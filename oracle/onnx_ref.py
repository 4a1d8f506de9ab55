"""CPU executor for parsed ONNX graphs (TEST INFRASTRUCTURE, see oracle/__init__.py): runs an
``pdf_table_amd.onnx_proto.OnnxModel`` node by node with torch ops, following the operator definitions of the public ONNX
operator schema (opset 13).  It exists so that the dependency-free protobuf reader can be checked end to end -- parse a
file PyTorch's exporter wrote, execute it here, compare with the module that was exported -- and is never imported by
pdf_table_amd/ (the product executes recognised architectures on the GPU only)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def run(model, feeds):
    """model: OnnxModel; feeds: {input name: np.ndarray} -> list of np.ndarray for model.outputs"""
    def tt(v):      # np.ascontiguousarray would turn a 0-d array into a 1-d one
        v = np.asarray(v)
        return torch.from_numpy(v.copy() if v.ndim else v.reshape(1).copy()).reshape(v.shape)

    env = {k: tt(v) for k, v in model.initializers.items()}
    env.update({k: tt(v) for k, v in feeds.items()})
    env[""] = None

    def ints(t):
        return [int(v) for v in t.reshape(-1).tolist()]

    for n in model.nodes:
        a = n.attrs
        x = [env[i] for i in n.inputs]
        t = n.op_type
        if t == "Constant":
            y = tt(a["value"])
        elif t == "Identity":
            y = x[0]
        elif t == "Conv":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = F.conv2d(x[0], x[1], x[2] if len(x) > 2 else None, a.get("strides", [1, 1]), (p[0], p[1]), a.get("dilations", [1, 1]),
                         a.get("group", 1))
        elif t == "ConvTranspose":
            p = a.get("pads", [0, 0, 0, 0])
            y = F.conv_transpose2d(x[0], x[1], x[2] if len(x) > 2 else None, a.get("strides", [1, 1]), (p[0], p[1]),
                                   groups=a.get("group", 1))
        elif t == "BatchNormalization":
            y = F.batch_norm(x[0], x[3], x[4], x[1], x[2], False, 0.0, a.get("epsilon", 1e-5))
        elif t == "Relu":
            y = torch.relu(x[0])
        elif t == "HardSigmoid":      # max(0, min(1, alpha x + beta)); ONNX defaults alpha 0.2, beta 0.5
            y = torch.clamp(a.get("alpha", 0.2) * x[0] + a.get("beta", 0.5), 0.0, 1.0)
        elif t == "HardSwish":
            y = x[0] * torch.clamp(x[0] / 6.0 + 0.5, 0.0, 1.0)
        elif t == "GlobalAveragePool":
            y = x[0].mean(dim=(2, 3), keepdim=True)
        elif t == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])) if ax else 1, -1)
        elif t == "Softmax":
            y = torch.softmax(x[0], dim=a.get("axis", -1))
        elif t == "Clip":
            lo = float(x[1]) if len(x) > 1 and x[1] is not None else a.get("min", None)
            hi = float(x[2]) if len(x) > 2 and x[2] is not None else a.get("max", None)
            y = torch.clamp(x[0], min=lo, max=hi)
        elif t == "AveragePool":
            y = torch.nn.functional.avg_pool2d(x[0], tuple(a["kernel_shape"]), tuple(a.get("strides", a["kernel_shape"])),
                                               tuple(a.get("pads", [0, 0, 0, 0])[:2]), ceil_mode=bool(a.get("ceil_mode", 0)),
                                               count_include_pad=bool(a.get("count_include_pad", 0)))
        elif t == "Sigmoid":
            y = torch.sigmoid(x[0])
        elif t == "MaxPool":
            p = a.get("pads", [0, 0, 0, 0])
            y = F.max_pool2d(x[0], a["kernel_shape"], a.get("strides", [1, 1]), (p[0], p[1]), ceil_mode=bool(a.get("ceil_mode", 0)))
        elif t in ("Add", "Mul", "Sub", "Div"):
            y = {"Add": torch.add, "Mul": torch.mul, "Sub": torch.sub, "Div": torch.div}[t](x[0], x[1])
        elif t == "Concat":
            y = torch.cat(x, a["axis"])
        elif t == "Resize":
            sc = next(([float(v) for v in t_.tolist()] for t_ in x[1:] if t_ is not None and t_.numel() == 4 and t_.is_floating_point()), None)
            assert a.get("mode", "nearest") == "nearest" and sc is not None
            y = F.interpolate(x[0], scale_factor=(sc[2], sc[3]), mode="nearest")
        elif t == "Transpose":
            y = x[0].permute(a["perm"])
        elif t == "Reshape":
            shp = ints(x[1])
            shp = [x[0].shape[i] if s == 0 else s for i, s in enumerate(shp)]
            y = x[0].reshape(shp)
        elif t == "Squeeze":
            ax = ints(x[1]) if len(x) > 1 else a.get("axes")
            y = x[0]
            for d in sorted(ax, reverse=True):
                y = y.squeeze(d)
        elif t == "Unsqueeze":
            ax = ints(x[1]) if len(x) > 1 else a.get("axes")
            y = x[0]
            for d in sorted(ax):
                y = y.unsqueeze(d)
        elif t == "Shape":
            y = torch.tensor(list(x[0].shape), dtype=torch.int64)
        elif t == "Gather":
            y = torch.index_select(x[0], a.get("axis", 0), x[1].reshape(-1).long())
            if x[1].dim() == 0:
                y = y.squeeze(a.get("axis", 0))
        elif t == "Slice":
            st, en = ints(x[1]), ints(x[2])
            axes = ints(x[3]) if len(x) > 3 and x[3] is not None else list(range(len(st)))
            steps = ints(x[4]) if len(x) > 4 and x[4] is not None else [1] * len(st)
            y = x[0]
            for s_, e_, ax, sp in zip(st, en, axes, steps):
                assert sp == 1
                dim = y.shape[ax]
                s_ = max(0, s_ + dim if s_ < 0 else s_)
                e_ = min(dim, e_ + dim if e_ < 0 else e_)
                y = y.narrow(ax, s_, max(0, e_ - s_))
        elif t == "Expand":
            y = x[0].expand(torch.broadcast_shapes(tuple(x[0].shape), tuple(ints(x[1])))).clone()
        elif t == "ConstantOfShape":
            v = a.get("value")
            y = torch.full(ints(x[0]), float(np.asarray(v).reshape(-1)[0]) if v is not None else 0.0)
        elif t == "Cast":
            y = x[0].to({1: torch.float32, 6: torch.int32, 7: torch.int64, 10: torch.float16, 11: torch.float64}[a["to"]])
        elif t == "MatMul":
            y = x[0] @ x[1]
        elif t == "Gemm":
            A = x[0].t() if a.get("transA", 0) else x[0]
            B = x[1].t() if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ B) + (a.get("beta", 1.0) * x[2] if len(x) > 2 else 0)
        elif t == "LSTM":
            y = _lstm(x, a)
            for name, v in zip(n.outputs, y):
                if name:
                    env[name] = v
            continue
        else:
            raise NotImplementedError(f"oracle/onnx_ref.py: operator {t}")
        env[n.outputs[0]] = y
    return [env[o.name].numpy() for o in model.outputs]


def _lstm(x, a):
    """ONNX LSTM: X [T, B, I], W [D, 4H, I], R [D, 4H, H], B [D, 8H]; gates i, o, f, c; Y [T, D, B, H]"""
    X, W, R = x[0], x[1], x[2]
    Bv = x[3] if len(x) > 3 and x[3] is not None else None
    h0 = x[5] if len(x) > 5 and x[5] is not None else None
    c0 = x[6] if len(x) > 6 and x[6] is not None else None
    H = a["hidden_size"]
    T, Bn, _ = X.shape
    D = W.shape[0]
    ys, hs, cs = [], [], []
    for d in range(D):
        h = h0[d] if h0 is not None else torch.zeros(Bn, H)
        c = c0[d] if c0 is not None else torch.zeros(Bn, H)
        b = (Bv[d, :4 * H] + Bv[d, 4 * H:]) if Bv is not None else torch.zeros(4 * H)
        out = [None] * T
        steps = range(T - 1, -1, -1) if (d == 1 or a.get("direction") == "reverse") else range(T)
        for t in steps:
            g = X[t] @ W[d].t() + h @ R[d].t() + b
            i, o, f, cc = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(cc)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[t] = h
        ys.append(torch.stack(out))
        hs.append(h)
        cs.append(c)
    return torch.stack(ys, 1), torch.stack(hs), torch.stack(cs)
