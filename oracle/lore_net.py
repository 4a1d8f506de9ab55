"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of Lore's DLA-34 + DCN detector.

Follows, function by function:
  * ``deform_conv2d``  -- modulated deformable 3x3 convolution = ``torchvision.ops.deform_conv2d`` as called from
    /root/reference/src/pdftable/model/lore/dcnv2.py:71-86.  torchvision is not installed here; its algorithm is the
    DCNv2 im2col the reference vendors at lore/DCNv2_latest/src/cpu/dcn_v2_im2col_cpu.cpp:26-55 (bilinear rule) and
    :123-190 (sampling grid, zero outside (-1, H) x (-1, W), value * mask).  PINNED against that very source, compiled
    from the reference tree into oracle/_ref/libdcnv2_ref.so (oracle/Makefile) -- tests/test_oracle_lore.py.
  * ``dla34_forward`` -- DLA.forward, center_net/modeling_centernet.py:382-402 with Tree.forward :259-271,
    Root.forward :167-175, BasicBlock.forward :58-72.
  * ``dlaseg_forward`` -- DLASeg.forward lore/lore_dla_34.py:184-196 with DLAUp.forward :128-134, IDAUp.forward :106-112,
    DeformConv.forward :75-83.  PINNED by tests/golden/lore_dla34.npz (outputs of the reference DLASeg itself with
    this file's ``deform_conv2d`` standing in for the missing torchvision op).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

HEADS = {"hm": 2, "st": 8, "wh": 8, "ax": 256, "cr": 256, "reg": 2}      # modeling_lore.py:89
DLA34_LEVELS = [1, 1, 1, 2, 2, 1]                                        # modeling_centernet.py:405-409
DLA34_CHANNELS = [16, 32, 64, 128, 256, 512]


def deform_conv2d(x: torch.Tensor, offset: torch.Tensor, mask: torch.Tensor, weight: torch.Tensor, bias,
                  stride: int = 1, pad: int = 1, dil: int = 1, return_cols: bool = False):
    """x [B,C,H,W], offset [B,2*K,Ho,Wo] (channel 2k = dy, 2k+1 = dx of tap k), mask [B,K,Ho,Wo] -> [B,O,Ho,Wo]."""
    B, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    K = kh * kw
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    f32 = torch.float32
    ys = (torch.arange(Ho, dtype=f32) * stride - pad).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=f32) * stride - pad).view(1, 1, Wo)
    xf = x.reshape(B, C, H * W)
    cols = x.new_zeros((B, C, K, Ho, Wo))
    for k in range(K):
        i, j = divmod(k, kw)
        h_im = ys + float(i * dil) + offset[:, 2 * k]               # [B,Ho,Wo]
        w_im = xs + float(j * dil) + offset[:, 2 * k + 1]
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh = h_im - h_low
        lw = w_im - w_low
        hh = 1 - lh
        hw = 1 - lw
        hl = h_low.long()
        wl = w_low.long()
        hh_i = hl + 1
        wh_i = wl + 1

        def tap(hi, wi, ok):
            ok = ok & inside
            idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
            v = torch.gather(xf, 2, idx).view(B, C, Ho, Wo)
            return v * ok.view(B, 1, Ho, Wo).to(x.dtype)
        v1 = tap(hl, wl, (hl >= 0) & (wl >= 0))
        v2 = tap(hl, wh_i, (hl >= 0) & (wh_i <= W - 1))
        v3 = tap(hh_i, wl, (hh_i <= H - 1) & (wl >= 0))
        v4 = tap(hh_i, wh_i, (hh_i <= H - 1) & (wh_i <= W - 1))
        w1 = (hh * hw).unsqueeze(1)
        w2 = (hh * lw).unsqueeze(1)
        w3 = (lh * hw).unsqueeze(1)
        w4 = (lh * lw).unsqueeze(1)
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4                 # same association as the C source
        cols[:, :, k] = val * mask[:, k].unsqueeze(1)
    if return_cols:
        return cols
    out = torch.einsum("ock,bckhw->bohw", weight.reshape(O, C, K), cols)
    if bias is not None:
        out = out + bias.view(1, O, 1, 1)
    return out


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _basic_block(sd, p, x, residual, stride):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1))
    return F.relu(out + residual)


def _root(sd, p, xs):
    return F.relu(_bn(sd, p + ".bn", F.conv2d(torch.cat(xs, 1), sd[p + ".conv.weight"])))


def _tree(sd, p, levels, x, cin, cout, stride, level_root, children=None):
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if level_root:
        children.append(bottom)
    if levels == 1:
        residual = bottom
        if cin != cout:
            residual = _bn(sd, p + ".project.1", F.conv2d(bottom, sd[p + ".project.0.weight"]))
        x1 = _basic_block(sd, p + ".tree1", x, residual, stride)
        x2 = _basic_block(sd, p + ".tree2", x1, x1, 1)
        return _root(sd, p + ".root", [x2, x1] + children)
    # levels > 1: the outer project() result is passed to tree1 and overwritten there (Tree.forward :262-265) -- skipped
    x1 = _tree(sd, p + ".tree1", levels - 1, x, cin, cout, stride, False)
    children.append(x1)
    return _tree(sd, p + ".tree2", levels - 1, x1, cout, cout, 1, False, children)


def dla34_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, prefix: str = "base") -> List[torch.Tensor]:
    ch = DLA34_CHANNELS
    p = prefix
    x = F.relu(_bn(sd, p + ".base_layer.1", F.conv2d(x, sd[p + ".base_layer.0.weight"], None, 1, 3)))
    ys = []
    x = F.relu(_bn(sd, p + ".level0.1", F.conv2d(x, sd[p + ".level0.0.weight"], None, 1, 1)))
    ys.append(x)
    x = F.relu(_bn(sd, p + ".level1.1", F.conv2d(x, sd[p + ".level1.0.weight"], None, 2, 1)))
    ys.append(x)
    for lvl in range(2, 6):
        x = _tree(sd, f"{p}.level{lvl}", DLA34_LEVELS[lvl], x, ch[lvl - 1], ch[lvl], 2, lvl > 2)
        ys.append(x)
    return ys


def dcn_block(sd, p, x):
    """DeformConv.forward (lore_dla_34.py:75-83): DCN (dcnv2.py:71-86) -> BN -> ReLU."""
    om = F.conv2d(x, sd[p + ".conv.conv_offset_mask.weight"], sd[p + ".conv.conv_offset_mask.bias"], 1, 1)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), 1)
    y = deform_conv2d(x, offset, torch.sigmoid(m), sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    return F.relu(_bn(sd, p + ".actf.0", y))


def _ida_up(sd, p, layers, startp, endp):
    for i in range(startp + 1, endp):
        j = i - startp
        w = sd[f"{p}.up_{j}.weight"]
        f = w.shape[2] // 2
        y = dcn_block(sd, f"{p}.proj_{j}", layers[i])
        y = F.conv_transpose2d(y, w, None, stride=f, padding=f // 2, groups=w.shape[0])
        layers[i] = dcn_block(sd, f"{p}.node_{j}", y + layers[i - 1])


def dlaseg_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, return_feature: bool = False):
    """x f32 [B,3,H,W] -> {'hm','st','wh','ax','cr','reg'} at H/4 x W/4 (first_level 2, last_level 5)."""
    first, last = 2, 5
    layers = dla34_forward(sd, x)
    # DLAUp.forward
    out = [layers[-1]]
    for i in range(len(layers) - first - 1):
        _ida_up(sd, f"dla_up.ida_{i}", layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    y = [out[i].clone() for i in range(last - first)]
    _ida_up(sd, "ida_up", y, 0, len(y))
    feat = y[-1]
    z = {}
    for h in HEADS:
        t = F.relu(F.conv2d(feat, sd[f"{h}.0.weight"], sd[f"{h}.0.bias"], 1, 1))
        z[h] = F.conv2d(t, sd[f"{h}.2.weight"], sd[f"{h}.2.bias"])
    if return_feature:
        return z, feat
    return z


# --------------------------------------------------------------------------------------------------------------------
# door to the compiled reference (oracle/_ref), used by tests only
# --------------------------------------------------------------------------------------------------------------------
def ref_dcn_im2col(x: np.ndarray, offset: np.ndarray, mask: np.ndarray, k: int = 3, pad: int = 1, stride: int = 1,
                   dil: int = 1) -> np.ndarray:
    """modulated_deformable_im2col_cpu of the reference's vendored DCNv2 -> columns [B, C*k*k, Ho, Wo]."""
    import ctypes
    import os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libdcnv2_ref.so"))
    B, C, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = np.ascontiguousarray(x, np.float32)
    offset = np.ascontiguousarray(offset, np.float32)
    mask = np.ascontiguousarray(mask, np.float32)
    col = np.zeros((B, C * k * k, Ho, Wo), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    ci = ctypes.c_int
    lib.modulated_deformable_im2col_cpu.argtypes = [fp, fp, fp] + [ci] * 15 + [fp]
    lib.modulated_deformable_im2col_cpu.restype = None
    lib.modulated_deformable_im2col_cpu(x.ctypes.data_as(fp), offset.ctypes.data_as(fp), mask.ctypes.data_as(fp),
                                        B, C, H, W, Ho, Wo, k, k, pad, pad, stride, stride, dil, dil, 1,
                                        col.ctypes.data_as(fp))
    return col


# --------------------------------------------------------------------------------------------------------------------
# 'wireless' detector: LoreDetectModel.forward, lore/lore_detector.py:353-389 (BasicBlock :57-88)
# --------------------------------------------------------------------------------------------------------------------
def lore_wireless_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor):
    """x f32 [B,3,H,W] -> the six head maps at H/4 x W/4.  PINNED by tests/golden/lore_wireless.npz."""
    def block(p, x, stride):
        out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride, 1)))
        out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, 1))
        res = x
        if (p + ".downsample.0.weight") in sd:
            res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
        return F.relu(out + res)

    def deconv(i, x):
        y = F.conv_transpose2d(x, sd[f"deconv_layers{i}.0.weight"], None, stride=2, padding=1)
        return F.relu(_bn(sd, f"deconv_layers{i}.1", y))
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], None, 2, 3)))
    x0 = F.max_pool2d(x, 3, 2, 1)
    feats = []
    cur = x0
    for li in range(1, 5):
        cur = block(f"layer{li}.0", cur, 2)
        cur = block(f"layer{li}.1", cur, 1)
        feats.append(cur)
    x1, x2, x3, x4 = feats
    x3_ = F.conv2d(x3, sd["adaption3.weight"]) + deconv(1, x4)
    x2_ = F.conv2d(x2, sd["adaption2.weight"]) + deconv(2, x3_)
    x1_ = F.conv2d(x1, sd["adaption1.weight"]) + deconv(3, x2_)
    x0_ = deconv(4, x1_) + F.conv2d(x0, sd["adaption0.weight"])
    x0_ = F.conv2d(x0_, sd["adaptionU1.weight"])
    z = {}
    for h in HEADS:
        n3 = 1 if h == "reg" else 4
        t = x0_
        for j in range(n3):
            t = F.relu(F.conv2d(t, sd[f"{h}.{2 * j}.weight"], sd[f"{h}.{2 * j}.bias"], 1, 1))
        z[h] = F.conv2d(t, sd[f"{h}.{2 * n3}.weight"], sd[f"{h}.{2 * n3}.bias"])
    return z
