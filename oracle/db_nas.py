"""Oracle for the DB-ProxylessNAS text-detection network (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``DBNasModel.forward`` (model/db_net/dbnet.py:693-712) in eval mode, fp32, as functional torch ops over a
``state_dict``:
  model/db_net/proxyless.py:14-31,92-178   CompactDetBackbone: 3x3 s2 stem, 24 blocks, outputs after blocks 5/11/17/23
  model/db_net/layers.py:42-58             MobileInvertedResidualBlock: conv(x) [+ x when it has the identity shortcut]
  model/db_net/layers.py:669-745           MBInvertedRepConvLayer (sum of depthwise branches)
  model/db_net/layers.py:93-160            MBInvertedConvLayer
  model/db_net/layers.py:469-490           SELayer (scale * input)
  model/db_net/dbnet.py:338-473            LightSegDetector.forward, eval branch returns ``binary`` only
  model/db_net/dbnet.py:35-99              DwPwConv / DwPwConvTranspose
Pinned against the reference module by tests/golden/db_nas.npz (tests/test_oracle_db_net.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from pdf_table_amd.dbnas_arch import OUTPUT_BLOCKS, dbnas_blocks

BN_EPS = 1e-5   # nn.BatchNorm2d default: CompactDetBackbone.__init__ never calls set_bn_param (proxyless.py:92-178)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=BN_EPS)


def _prelu(sd, p, x):
    return F.prelu(x, sd[p + ".weight"])


def _conv_block(sd, p, b, x):
    """the `mobile_inverted_conv` of a 'rep' or 'mb' block"""
    mid = b["mid"]
    y = F.conv2d(x, sd[p + ".inverted_bottleneck.conv.weight"])
    y = _prelu(sd, p + ".inverted_bottleneck.act", _bn(sd, p + ".inverted_bottleneck.bn", y))
    if b["kind"] == "rep":          # layers.py:732-745 (deploy False): out[0] + out[1] + ...
        acc = None
        for ri, k in enumerate(b["sizes"]):
            t = F.conv2d(y, sd[f"{p}.rep_conv.{ri}.conv.weight"], None, b["stride"], k // 2, 1, mid)
            t = _bn(sd, f"{p}.rep_conv.{ri}.bn", t)
            acc = t if acc is None else acc + t
        y = _prelu(sd, p + ".act", acc)
    else:                           # layers.py:146-151
        k = b["sizes"][0]
        y = F.conv2d(y, sd[p + ".depth_conv.conv.weight"], None, b["stride"], k // 2, 1, mid)
        y = _prelu(sd, p + ".depth_conv.act", _bn(sd, p + ".depth_conv.bn", y))
    return _bn(sd, p + ".point_conv.bn", F.conv2d(y, sd[p + ".point_conv.conv.weight"]))


def dbnas_backbone_fp32(sd, x: torch.Tensor):
    """x f32 [n,3,H,W] -> (c2, c3, c4, c5) at strides 4, 8, 16, 32"""
    x = F.conv2d(x, sd["backbone.first_conv.0.weight"], None, 2, 1)
    x = F.relu(_bn(sd, "backbone.first_conv.1", x))
    outs = []
    for bi, b in enumerate(dbnas_blocks()):
        p = f"backbone.blocks.{bi}.mobile_inverted_conv"
        if b["kind"] == "se":       # residual block with an identity shortcut around SELayer: x + scale * x
            s = F.adaptive_avg_pool2d(x, 1)
            s = F.relu(F.conv2d(s, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]))
            s = torch.sigmoid(F.conv2d(s, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"]))
            x = x + s * x
        else:
            y = _conv_block(sd, p, b, x)
            x = x + y if b["shortcut"] else y
        if bi in OUTPUT_BLOCKS:
            outs.append(x)
    return tuple(outs)


def _dwpw_t(sd, p, x):
    """DwPwConvTranspose (dbnet.py:75-99)"""
    c = x.shape[1]
    y = F.conv_transpose2d(x, sd[p + ".depthwise.weight"], sd[p + ".depthwise.bias"], 2, 0, 0, c)
    y = F.relu(_bn(sd, p + ".bn1", y))
    return F.conv2d(y, sd[p + ".pointwise.weight"], sd[p + ".pointwise.bias"])


def dbnas_decoder_fp32(sd, feats, return_logits: bool = False):
    c2, c3, c4, c5 = feats
    up = lambda t, f: F.interpolate(t, scale_factor=f, mode="nearest")
    p5 = up(F.conv2d(c5, sd["decoder.in5.weight"]), 8)
    p4 = up(F.conv2d(c4, sd["decoder.in4.weight"]), 4)
    p3 = up(F.conv2d(c3, sd["decoder.in3.weight"]), 2)
    p2 = F.conv2d(c2, sd["decoder.in2.weight"])
    fuse = p5 + p4 + p3 + p2
    c = fuse.shape[1]
    k = sd["decoder.binarize.0.depthwise.weight"].shape[-1]
    y = F.conv2d(fuse, sd["decoder.binarize.0.depthwise.weight"], None, 1, k // 2, 1, c)
    y = F.relu(_bn(sd, "decoder.binarize.0.bn1", y))
    y = F.conv2d(y, sd["decoder.binarize.0.pointwise.weight"])
    y = F.relu(_bn(sd, "decoder.binarize.1", y))
    y = _dwpw_t(sd, "decoder.binarize.3", y)
    y = F.relu(_bn(sd, "decoder.binarize.4", y))
    logits = _dwpw_t(sd, "decoder.binarize.6", y)
    return logits if return_logits else torch.sigmoid(logits)


def dbnas_forward_fp32(sd, x: torch.Tensor, return_logits: bool = False):
    """x f32 [n,3,H,W] (H, W multiples of 32) -> probability map f32 [n,1,H,W]"""
    with torch.no_grad():
        return dbnas_decoder_fp32(sd, dbnas_backbone_fp32(sd, x), return_logits)
