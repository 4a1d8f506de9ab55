"""ONNX files for validating the importer (pdf_table_amd/onnx_import.py) -- test / tooling code, not product.

Two independent producers:

  * ``torch_export(module, example)``: PyTorch's own TorchScript ONNX exporter -- what the reference's export path calls
    (``DeployUtils.export_onnx`` -> ``torch.onnx.export``, utils/deploy_utils.py:197-224) -- driven through its internal
    entry points because ``torch.onnx.export`` insists on the ``onnx`` Python package, which is only used for a
    post-export check.  The serialiser itself is C++ inside torch.  Eval-mode export folds BatchNorm into the preceding
    Conv, turns nn.LSTM into an ONNX LSTM (gate order i, o, f, c) and nn.Upsample into Resize: the graphs the importer must
    digest in practice.
  * ``write_db_resnet18(sd)``: a hand-built, UNFOLDED graph (Conv + BatchNormalization + Relu ... with the state_dict's own
    tensor names) through pdf_table_amd.onnx_proto.serialize_model -- importing it must give back the state_dict bit for
    bit.

    python tools/onnx_export.py db_resnet18 out.onnx [--unfolded]
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pdf_table_amd.onnx_proto import OnnxModel, OnnxNode, OnnxValueInfo, serialize_model  # noqa: E402


def torch_export(module: torch.nn.Module, example: torch.Tensor, input_name="x", output_name="y", opset=13, dynamic_batch=False) -> bytes:
    """dynamic_batch: axis 0 of the input and the output is symbolic (as in the shipped PP-OCR exports): the graph then carries its
    Shape -> Gather -> Concat arithmetic instead of baked Reshape constants"""
    import torch.onnx
    from torch.onnx._internal.torchscript_exporter import utils as TU
    from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
    module = module.eval()
    dyn = {input_name: {0: "batch"}, output_name: {0: "batch"}} if dynamic_batch else {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        GLOBALS.export_onnx_opset_version = opset
        with torch.no_grad():
            graph, params, _ = TU._model_to_graph(module, (example,), False, [input_name], [output_name],
                                                  torch.onnx.OperatorExportTypes.ONNX, True, dynamic_axes=dyn)
        proto = graph._export_onnx(params, opset, dyn, False, torch.onnx.OperatorExportTypes.ONNX, True, True, {}, True, "", {})[0]
    return bytes(proto)


class _DbModule(torch.nn.Module):
    """DBModel as real nn modules (the same graph torch.onnx.export sees for the reference's class): built here from a
    state_dict so that tests need neither the reference tree nor the oracle's functional restatement"""

    def __init__(self, sd):
        super().__init__()
        nn = torch.nn

        def bn(c):
            return nn.BatchNorm2d(c)
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), bn(64)
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.blocks = nn.ModuleList()
        inpl = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):
            for bi in range(2):
                stride = 2 if (bi == 0 and li > 1) else 1
                blk = nn.ModuleDict({"conv1": nn.Conv2d(inpl, planes, 3, stride, 1, bias=False), "bn1": bn(planes),
                                     "conv2": nn.Conv2d(planes, planes, 3, 1, 1, bias=False), "bn2": bn(planes)})
                if stride != 1 or inpl != planes:
                    blk["down"] = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), bn(planes))
                self.blocks.append(blk)
                inpl = planes
        self.inl = nn.ModuleList([nn.Conv2d(c, 256, 1, bias=False) for c in (512, 256, 128, 64)])       # in5, in4, in3, in2
        self.outl = nn.ModuleList([nn.Conv2d(256, 64, 3, padding=1, bias=False) for _ in range(4)])      # out5 .. out2
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.ups = nn.ModuleList([nn.Upsample(scale_factor=f, mode="nearest") for f in (8, 4, 2)])
        self.binarize = nn.Sequential(nn.Conv2d(256, 64, 3, padding=1, bias=False), bn(64), nn.ReLU(inplace=True),
                                      nn.ConvTranspose2d(64, 64, 2, 2), bn(64), nn.ReLU(inplace=True),
                                      nn.ConvTranspose2d(64, 1, 2, 2), nn.Sigmoid())
        m = {"backbone.conv1": self.conv1, "backbone.bn1": self.bn1}
        k = 0
        for li in range(1, 5):
            for bi in range(2):
                p = f"backbone.layer{li}.{bi}"
                blk = self.blocks[k]
                k += 1
                m.update({p + ".conv1": blk["conv1"], p + ".bn1": blk["bn1"], p + ".conv2": blk["conv2"], p + ".bn2": blk["bn2"]})
                if "down" in blk:
                    m.update({p + ".downsample.0": blk["down"][0], p + ".downsample.1": blk["down"][1]})
        for name, mod in zip(("in5", "in4", "in3", "in2"), self.inl):
            m["decoder." + name] = mod
        for name, mod in zip(("out5.0", "out4.0", "out3.0", "out2"), self.outl):
            m["decoder." + name] = mod
        for i in (0, 1, 3, 4, 6):
            m[f"decoder.binarize.{i}"] = self.binarize[i]
        with torch.no_grad():
            for prefix, mod in m.items():
                for pn, par in list(mod.named_parameters(recurse=False)) + list(mod.named_buffers(recurse=False)):
                    par.copy_(sd[prefix + "." + pn])

    def forward(self, x):
        x = self.pool(torch.relu(self.bn1(self.conv1(x))))
        feats = []
        for k, blk in enumerate(self.blocks):
            out = torch.relu(blk["bn1"](blk["conv1"](x)))
            out = blk["bn2"](blk["conv2"](out))
            res = blk["down"](x) if "down" in blk else x
            x = torch.relu(out + res)
            if k % 2 == 1:
                feats.append(x)
        c2, c3, c4, c5 = feats
        in5, in4, in3, in2 = self.inl[0](c5), self.inl[1](c4), self.inl[2](c3), self.inl[3](c2)
        out4 = self.up(in5) + in4
        out3 = self.up(out4) + in3
        out2 = self.up(out3) + in2
        p5 = self.ups[0](self.outl[0](in5))
        p4 = self.ups[1](self.outl[1](out4))
        p3 = self.ups[2](self.outl[2](out3))
        p2 = self.outl[3](out2)
        return self.binarize(torch.cat((p5, p4, p3, p2), 1))


class _CrnnModule(torch.nn.Module):
    """CRNN (crnn/modeling_crnn.py:40-113) as nn modules, from a state_dict"""

    def __init__(self, sd):
        super().__init__()
        nn = torch.nn
        spec = [(1, 64, (3, 3), 1, 1), (64, 128, (3, 3), 1, 1), (128, 256, (3, 3), 1, 1), (256, 256, (3, 3), 1, 1),
                (256, 512, (3, 3), 1, 1), (512, 512, (3, 3), 1, 1), (512, 512, (2, 1), (2, 1), 0)]
        self.convs = nn.ModuleList([nn.Sequential(nn.Conv2d(ci, co, k, s, p), nn.BatchNorm2d(co), nn.ReLU(inplace=True))
                                    for ci, co, k, s, p in spec])
        self.rnn0, self.emb0 = nn.LSTM(512, 256, bidirectional=True), nn.Linear(512, 256)
        self.rnn1, self.emb1 = nn.LSTM(256, 256, bidirectional=True), nn.Linear(512, 512)
        self.cls = nn.Linear(512, sd["cls.weight"].shape[0], bias=False)
        names = ["conv0.0", "conv1.0", "conv2.0", "conv2.3", "conv3.0", "conv3.3", "conv4.0"]
        bns = ["conv0.1", "conv1.1", "conv2.1", "conv2.4", "conv3.1", "conv3.4", "conv4.1"]
        with torch.no_grad():
            for seq, cn, bn in zip(self.convs, names, bns):
                seq[0].weight.copy_(sd[cn + ".weight"]); seq[0].bias.copy_(sd[cn + ".bias"])
                for pn in ("weight", "bias", "running_mean", "running_var"):
                    getattr(seq[1], pn).copy_(sd[f"{bn}.{pn}"])
            for k, (rnn, emb) in enumerate(((self.rnn0, self.emb0), (self.rnn1, self.emb1))):
                rnn.load_state_dict({kk[len(f"rnn.{k}.rnn."):]: v for kk, v in sd.items() if kk.startswith(f"rnn.{k}.rnn.")})
                emb.weight.copy_(sd[f"rnn.{k}.embedding.weight"]); emb.bias.copy_(sd[f"rnn.{k}.embedding.bias"])
            self.cls.weight.copy_(sd["cls.weight"])

    def forward(self, x):
        F = torch.nn.functional
        g = x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140
        f = F.max_pool2d(self.convs[0](g), (2, 2), (2, 2))
        f = F.max_pool2d(self.convs[1](f), (2, 2), (2, 2))
        f = F.max_pool2d(self.convs[3](self.convs[2](f)), (2, 1), (2, 1))
        f = F.max_pool2d(self.convs[5](self.convs[4](f)), (2, 1), (2, 1))
        f = self.convs[6](f).squeeze(2).permute(2, 0, 1)
        r, _ = self.rnn0(f)
        r = self.emb0(r)
        r, _ = self.rnn1(r)
        r = self.emb1(r)
        return self.cls(r).permute(1, 0, 2)


def export_db_resnet18(sd, h=64, w=64) -> bytes:
    return torch_export(_DbModule(sd), torch.zeros(1, 3, h, w))


def export_crnn(sd, w=640) -> bytes:
    return torch_export(_CrnnModule(sd), torch.zeros(1, 3, 32, w))


def write_db_resnet18(sd, h=64, w=64) -> bytes:
    """the unfolded graph, node by node, with the state_dict's own tensor names"""
    m = OnnxModel(producer="pdf_table_amd.tools.onnx_export", graph_name="db_resnet18")
    init = m.initializers
    cnt = [0]

    def T(name):
        init[name] = sd[name].numpy()
        return name

    def node(op, ins, attrs=None):
        cnt[0] += 1
        out = f"t{cnt[0]}"
        m.nodes.append(OnnxNode(op, list(ins), [out], dict(attrs or {}), name=f"{op}_{cnt[0]}"))
        return out

    def conv(x, p, k, s, pad):
        ins = [x, T(p + ".weight")] + ([T(p + ".bias")] if (p + ".bias") in sd else [])
        return node("Conv", ins, {"kernel_shape": [k, k], "strides": [s, s], "pads": [pad] * 4, "group": 1, "dilations": [1, 1]})

    def bn(x, p):
        return node("BatchNormalization", [x, T(p + ".weight"), T(p + ".bias"), T(p + ".running_mean"), T(p + ".running_var")],
                    {"epsilon": 1e-5, "momentum": 0.9})

    def up(x, f):
        init[f"scales{f}"] = np.array([1, 1, f, f], np.float32)
        init["roi"] = np.zeros((0,), np.float32)
        return node("Resize", [x, "roi", f"scales{f}"], {"mode": "nearest", "coordinate_transformation_mode": "asymmetric",
                                                         "nearest_mode": "floor"})

    x = node("Relu", [bn(conv("x", "backbone.conv1", 7, 2, 3), "backbone.bn1")])
    x = node("MaxPool", [x], {"kernel_shape": [3, 3], "strides": [2, 2], "pads": [1, 1, 1, 1]})
    feats = []
    for li in range(1, 5):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            s = 2 if (bi == 0 and li > 1) else 1
            o = node("Relu", [bn(conv(x, p + ".conv1", 3, s, 1), p + ".bn1")])
            o = bn(conv(o, p + ".conv2", 3, 1, 1), p + ".bn2")
            r = bn(conv(x, p + ".downsample.0", 1, s, 0), p + ".downsample.1") if (p + ".downsample.0.weight") in sd else x
            x = node("Relu", [node("Add", [o, r])])
        feats.append(x)
    c2, c3, c4, c5 = feats
    in5, in4 = conv(c5, "decoder.in5", 1, 1, 0), conv(c4, "decoder.in4", 1, 1, 0)
    in3, in2 = conv(c3, "decoder.in3", 1, 1, 0), conv(c2, "decoder.in2", 1, 1, 0)
    out4 = node("Add", [up(in5, 2), in4])
    out3 = node("Add", [up(out4, 2), in3])
    out2 = node("Add", [up(out3, 2), in2])
    p5 = up(conv(in5, "decoder.out5.0", 3, 1, 1), 8)
    p4 = up(conv(out4, "decoder.out4.0", 3, 1, 1), 4)
    p3 = up(conv(out3, "decoder.out3.0", 3, 1, 1), 2)
    p2 = conv(out2, "decoder.out2", 3, 1, 1)
    y = node("Concat", [p5, p4, p3, p2], {"axis": 1})
    y = node("Relu", [bn(conv(y, "decoder.binarize.0", 3, 1, 1), "decoder.binarize.1")])

    def convT(x, p):
        return node("ConvTranspose", [x, T(p + ".weight"), T(p + ".bias")], {"kernel_shape": [2, 2], "strides": [2, 2], "pads": [0, 0, 0, 0],
                                                                             "group": 1, "dilations": [1, 1]})
    y = node("Relu", [bn(convT(y, "decoder.binarize.3"), "decoder.binarize.4")])
    y = node("Sigmoid", [convT(y, "decoder.binarize.6")])
    m.nodes[-1].outputs = ["y"]
    m.inputs = [OnnxValueInfo("x", 1, (1, 3, h, w))]
    m.outputs = [OnnxValueInfo("y", 1, (1, 1, h, w))]
    return serialize_model(m)


if __name__ == "__main__":
    from pdf_table_amd.synth_weights import crnn_state_dict, db_resnet18_state_dict
    which, path = sys.argv[1], sys.argv[2]
    if which == "db_resnet18":
        sd = db_resnet18_state_dict(seed=0)
        data = write_db_resnet18(sd) if "--unfolded" in sys.argv else export_db_resnet18(sd)
    elif which == "crnn":
        data = export_crnn(crnn_state_dict(seed=1))
    else:
        raise SystemExit("db_resnet18 | crnn")
    with open(path, "wb") as f:
        f.write(data)
    print(path, len(data), "bytes")


# ---- sequence / multi-output stand-ins for the executor tests (tests/test_gpu_onnx_seq.py) --------------------------------------------
class SvtrBlock(torch.nn.Module):
    """one global-mixing block of an SVTR-type recogniser (PaddleOCR ppocr/modeling/backbones/rec_svtrnet.py: fused qkv, (B, heads, T, d)
    attention, MLP), pre-norm"""

    def __init__(self, dim=64, heads=4, mlp=2.0, act="gelu"):
        super().__init__()
        self.heads, self.hd = heads, dim // heads
        self.norm1, self.norm2 = torch.nn.LayerNorm(dim, eps=1e-6), torch.nn.LayerNorm(dim, eps=1e-6)
        self.qkv, self.proj = torch.nn.Linear(dim, 3 * dim), torch.nn.Linear(dim, dim)
        self.fc1, self.fc2 = torch.nn.Linear(dim, int(dim * mlp)), torch.nn.Linear(int(dim * mlp), dim)
        self.act = torch.nn.GELU() if act == "gelu" else torch.nn.SiLU()

    def forward(self, x):
        B, T, C = x.shape
        qkv = self.qkv(self.norm1(x)).reshape(B, T, 3, self.heads, self.hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (self.hd ** -0.5), qkv[1], qkv[2]
        a = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
        x = x + self.proj((a @ v).transpose(1, 2).reshape(B, T, C))
        return x + self.fc2(self.act(self.fc1(self.norm2(x))))


class SvtrTiny(torch.nn.Module):
    """conv stem -> tokens -> two SVTR blocks -> LayerNorm -> CTC head with its Softmax: the operator set of a PP-OCRv4-type recogniser"""

    def __init__(self, dim=64, classes=97, act="gelu"):
        super().__init__()
        self.c1, self.b1 = torch.nn.Conv2d(3, 32, 3, 2, 1, bias=False), torch.nn.BatchNorm2d(32)
        self.c2, self.b2 = torch.nn.Conv2d(32, dim, 3, 2, 1, bias=False), torch.nn.BatchNorm2d(dim)
        self.blocks = torch.nn.ModuleList([SvtrBlock(dim, 4, 2.0, act), SvtrBlock(dim, 4, 2.0, act)])
        self.norm = torch.nn.LayerNorm(dim, eps=1e-6)
        self.head = torch.nn.Linear(dim, classes)

    def forward(self, x):
        x = torch.relu(self.b2(self.c2(torch.relu(self.b1(self.c1(x))))))
        x = x.flatten(2).transpose(1, 2)                         # [B, H W, C]
        for b in self.blocks:
            x = b(x)
        return torch.softmax(self.head(self.norm(x)), dim=-1)


class PicoLike(torch.nn.Module):
    """backbone -> levels -> per level one head conv whose channels are split into class scores (sigmoid) and box distributions, each
    flattened to [B, anchors, C]: the output convention of the PicoDet export the reference's layout stage consumes
    (ocr_layout_task.py:159-175: first half of the outputs = scores, second half = distributions).  levels = 3: strides 4 / 8 / 16 with a
    top-down path (nearest x2 Resize + Add); levels = 4: strides 8 / 16 / 32 / 64 like picodet_lcnet_x1_0_layout, no top-down path (odd sizes)"""

    def __init__(self, ncls=5, reg=32, levels=3):
        super().__init__()
        self.ncls, self.levels = ncls, levels
        cbr = lambda i, o, s: torch.nn.Sequential(torch.nn.Conv2d(i, o, 3, s, 1, bias=False), torch.nn.BatchNorm2d(o), torch.nn.Hardswish())
        self.s1, self.s2, self.s3, self.s4 = cbr(3, 32, 2), cbr(32, 64, 2), cbr(64, 96, 2), cbr(96, 128, 2)
        chans = (64, 96, 128)
        if levels == 4:
            self.s5, self.s6 = cbr(128, 128, 2), cbr(128, 128, 2)
            chans = (96, 128, 128, 128)
        self.lat = torch.nn.ModuleList([torch.nn.Conv2d(c, 64, 1) for c in chans])
        self.heads = torch.nn.ModuleList([torch.nn.Conv2d(64, ncls + reg, 1) for _ in chans])

    def forward(self, x):
        c2 = self.s2(self.s1(x))
        c3 = self.s3(c2)
        c4 = self.s4(c3)
        if self.levels == 4:
            c5 = self.s5(c4)
            feats = [l(c) for l, c in zip(self.lat, (c3, c4, c5, self.s6(c5)))]
        else:
            p4 = self.lat[2](c4)
            p3 = self.lat[1](c3) + torch.nn.functional.interpolate(p4, scale_factor=2.0, mode="nearest")
            feats = [self.lat[0](c2) + torch.nn.functional.interpolate(p3, scale_factor=2.0, mode="nearest"), p3, p4]
        scores, dists = [], []
        for p, h in zip(feats, self.heads):
            o = h(p)
            scores.append(torch.sigmoid(o[:, :self.ncls]).flatten(2).permute(0, 2, 1))
            dists.append(o[:, self.ncls:].flatten(2).permute(0, 2, 1))
        return tuple(scores + dists)


def seeded(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """seeded parameters incl. non-trivial BatchNorm statistics"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.ndim == 1 else (2.0 / max(1, p[0].numel())) ** 0.5))
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("norm.weight") or (p.ndim == 1 and "b" in n.split(".")[-2][:1] and n.endswith("weight")):
                p.add_(1.0)
        for n, b in module.named_buffers():
            if n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
            elif n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
    return module.eval()
