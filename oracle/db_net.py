"""Oracle for the DB-ResNet18 text-detection network (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``DBModel.forward`` = ``SegDetector(ResNet(BasicBlock,[2,2,2,2]))`` in eval mode:
  model/db_net/dbnet.py:715-728  (DBModel)
  model/db_net/dbnet.py:260-336  (ResNet: 7x7 s2 stem, BN, ReLU, maxpool 3x3 s2 p1, 4 stages)
  model/db_net/dbnet.py:102-171  (BasicBlock: conv-bn-relu, conv-bn, [downsample], add, relu)
  model/db_net/dbnet.py:615-638  (SegDetector.forward, eval branch returns ``binary`` only)
  model/db_net/dbnet.py:533-539  (binarize head: conv3x3-BN-ReLU-ConvT2x2-BN-ReLU-ConvT2x2-Sigmoid)

Two arithmetic modes over the same graph:

``db_forward_fp32``  -- the reference's own op sequence in fp32 (conv, then BatchNorm with running
    stats, then ReLU).  Pinned against the reference module by tests/golden/db_resnet18_*.npz.

``db_forward_bf16``  -- the arithmetic contract of the HIP engine (DESIGN.md "numerics"): BatchNorm
    folded into the conv (float64 fold, weights rounded to bf16 RNE, bias kept fp32), bf16
    activations between layers, fp32 accumulation, fp32 logits + sigmoid at the end.  Rounding
    happens at exactly the points where the engine stores an activation to HBM.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, dbnet.py:14


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=BN_EPS)


def _basic_block_fp32(sd, p, x, stride):
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, p + ".bn1", out))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, p + ".bn2", out)
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
    else:
        res = x
    return F.relu(out + res)


def db_backbone_fp32(sd, x):
    """ResNet.forward dbnet.py:324-335 -> (c2, c3, c4, c5)."""
    x = F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3)
    x = F.relu(_bn(sd, "backbone.bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(1, 5):
        for bi in range(2):
            x = _basic_block_fp32(sd, f"backbone.layer{li}.{bi}", x, 2 if (li > 1 and bi == 0) else 1)
        feats.append(x)
    return feats


def _up(x, f):
    return F.interpolate(x, scale_factor=f, mode="nearest")


def db_forward_fp32(sd, x, return_logits=False):
    """x: fp32 [B,3,H,W] (H,W multiples of 32) -> prob map fp32 [B,1,H,W]."""
    c2, c3, c4, c5 = db_backbone_fp32(sd, x)
    in5 = F.conv2d(c5, sd["decoder.in5.weight"])
    in4 = F.conv2d(c4, sd["decoder.in4.weight"])
    in3 = F.conv2d(c3, sd["decoder.in3.weight"])
    in2 = F.conv2d(c2, sd["decoder.in2.weight"])
    out4 = _up(in5, 2) + in4
    out3 = _up(out4, 2) + in3
    out2 = _up(out3, 2) + in2
    p5 = _up(F.conv2d(in5, sd["decoder.out5.0.weight"], None, 1, 1), 8)
    p4 = _up(F.conv2d(out4, sd["decoder.out4.0.weight"], None, 1, 1), 4)
    p3 = _up(F.conv2d(out3, sd["decoder.out3.0.weight"], None, 1, 1), 2)
    p2 = F.conv2d(out2, sd["decoder.out2.weight"], None, 1, 1)
    fuse = torch.cat((p5, p4, p3, p2), 1)
    y = F.conv2d(fuse, sd["decoder.binarize.0.weight"], None, 1, 1)
    y = F.relu(_bn(sd, "decoder.binarize.1", y))
    y = F.conv_transpose2d(y, sd["decoder.binarize.3.weight"], sd["decoder.binarize.3.bias"], 2)
    y = F.relu(_bn(sd, "decoder.binarize.4", y))
    y = F.conv_transpose2d(y, sd["decoder.binarize.6.weight"], sd["decoder.binarize.6.bias"], 2)
    if return_logits:
        return y
    return torch.sigmoid(y)


# --------------------------------------------------------------------------------------------
# bf16 engine-contract emulation
# --------------------------------------------------------------------------------------------

def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16, returned as fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


def fold_bn(sd, conv, bn, transposed=False):
    """Fold eval-mode BN into the preceding conv.  float64 fold -> (bf16-rounded weight as fp32, fp32 bias)."""
    w = sd[conv + ".weight"].double()
    b = sd[conv + ".bias"].double() if (conv + ".bias") in sd else None
    if bn is not None:
        s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
        shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * s
        if transposed:  # [Cin, Cout, kh, kw]
            w = w * s.view(1, -1, 1, 1)
        else:           # [Cout, Cin, kh, kw]
            w = w * s.view(-1, 1, 1, 1)
        b = shift if b is None else b * s + shift
    if b is None:
        b = torch.zeros(w.shape[1] if transposed else w.shape[0], dtype=torch.float64)
    return bf16_round(w.float()), b.float()


def _conv_bf16(sd, x, conv, bn, stride, pad, relu, res=None):
    w, b = fold_bn(sd, conv, bn)
    y = F.conv2d(x, w, b, stride, pad)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    return bf16_round(y)


def db_forward_bf16(sd, x, return_logits=False, return_features=False):
    """x: fp32 tensor holding bf16-representable values [B,3,H,W] -> prob fp32 [B,1,H,W]."""
    x = bf16_round(x)
    x = _conv_bf16(sd, x, "backbone.conv1", "backbone.bn1", 2, 3, True)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(1, 5):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            stride = 2 if (li > 1 and bi == 0) else 1
            out = _conv_bf16(sd, x, p + ".conv1", p + ".bn1", stride, 1, True)
            if (p + ".downsample.0.weight") in sd:
                res = _conv_bf16(sd, x, p + ".downsample.0", p + ".downsample.1", stride, 0, False)
            else:
                res = x
            x = _conv_bf16(sd, out, p + ".conv2", p + ".bn2", 1, 1, True, res=res)
        feats.append(x)
    c2, c3, c4, c5 = feats
    in5 = _conv_bf16(sd, c5, "decoder.in5", None, 1, 0, False)
    out4 = _conv_bf16(sd, c4, "decoder.in4", None, 1, 0, False, res=_up(in5, 2))
    out3 = _conv_bf16(sd, c3, "decoder.in3", None, 1, 0, False, res=_up(out4, 2))
    out2 = _conv_bf16(sd, c2, "decoder.in2", None, 1, 0, False, res=_up(out3, 2))
    p5 = _up(_conv_bf16(sd, in5, "decoder.out5.0", None, 1, 1, False), 8)
    p4 = _up(_conv_bf16(sd, out4, "decoder.out4.0", None, 1, 1, False), 4)
    p3 = _up(_conv_bf16(sd, out3, "decoder.out3.0", None, 1, 1, False), 2)
    p2 = _conv_bf16(sd, out2, "decoder.out2", None, 1, 1, False)
    fuse = torch.cat((p5, p4, p3, p2), 1)
    y = _conv_bf16(sd, fuse, "decoder.binarize.0", "decoder.binarize.1", 1, 1, True)
    w, b = fold_bn(sd, "decoder.binarize.3", "decoder.binarize.4", transposed=True)
    y = bf16_round(F.relu(F.conv_transpose2d(y, w, b, 2)))
    w, b = fold_bn(sd, "decoder.binarize.6", None, transposed=True)
    logits = F.conv_transpose2d(y, w, b, 2)
    out = logits if return_logits else torch.sigmoid(logits)
    if return_features:
        return out, dict(c2=c2, c3=c3, c4=c4, c5=c5, fuse=fuse)
    return out
