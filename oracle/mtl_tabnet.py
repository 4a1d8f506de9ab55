"""CPU fp32 restatement of the MtlTabNet backbone (second half) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

What it restates (reference file:line):
  TableResNetExtra.forward     model/table/mtl_tabnet/table_resnet_extra.py:205-318 (conv1/conv2 + max-pool, four stages of
                               BasicBlocks [1, 2, 5, 3] with a 3x3 conv + BN + ReLU and a max-pool between them; three
                               feature maps at 1/2, 1/4 and 1/8 of the input)
  BasicBlock.forward           :164-202 (conv-bn-relu-conv-bn, global-context block, + residual / 1x1 down-sample, ReLU)
  ContextBlock                 :36-161 (attention pooling: 1x1 conv mask, soft-max over H*W, weighted sum of the map;
                               ``channel_add``: 1x1 -> LayerNorm([planes, 1, 1]) -> ReLU -> 1x1, broadcast add)
  configuration                model/table/mtl_tabnet/mtl_tabnet_config.py:41-53 (gcb ratio 1/16, one header, blocks with a
                               context block: the FIRST block of stages 2-4)
Pinned by ``tests/golden/mtl_tabnet_backbone.npz`` (the reference's own ``TableResNetExtra`` module on seeded weights,
``tests/golden/make_golden.py::gen_mtl_tabnet_backbone``).  The decoders (master_decoder.py:194-531) are not restated yet:
PARITY UNPINNED for everything of MtlTabNet beyond the backbone; no engine path exists for it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = (1, 2, 5, 3)
GCB_LAYERS = (False, True, True, True)
GCB_RATIO = 0.0625
BN_EPS = 1e-5


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def context_block(sd, p, x):
    """ContextBlock.forward, pooling 'att', one header, fusion 'channel_add' (table_resnet_extra.py:89-141)."""
    n, c, h, w = x.shape
    mask = F.conv2d(x, sd[p + ".conv_mask.weight"], sd[p + ".conv_mask.bias"]).view(n, 1, h * w)
    mask = torch.softmax(mask, dim=2).unsqueeze(-1)                                  # [n, 1, hw, 1]
    ctx = torch.matmul(x.view(n, c, h * w).unsqueeze(1), mask).view(n, c, 1, 1)      # [n, c, 1, 1]
    q = p + ".channel_add_conv"
    t = F.conv2d(ctx, sd[q + ".0.weight"], sd[q + ".0.bias"])
    t = F.layer_norm(t, list(t.shape[1:]), sd[q + ".1.weight"], sd[q + ".1.bias"])
    t = F.conv2d(F.relu(t), sd[q + ".3.weight"], sd[q + ".3.bias"])
    return x + t


def basic_block(sd, p, x, gcb):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], padding=1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1))
    if gcb:
        out = context_block(sd, p + ".context_block", out)
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"]))
    return F.relu(out + res)


def backbone_forward_fp32(sd, x):
    """x fp32 [n, 3, H, W] -> [f1 (256, H/2), f2 (256, H/4), f3 (512, H/8)] like TableResNetExtra.forward."""
    def cbr(i, t):
        return F.relu(_bn(sd, f"bn{i}", F.conv2d(t, sd[f"conv{i}.weight"], padding=1)))

    def layer(i, t):
        for j in range(LAYERS[i - 1]):
            t = basic_block(sd, f"layer{i}.{j}", t, GCB_LAYERS[i - 1] and j == 0)
        return t
    feats = []
    x = cbr(2, cbr(1, x))
    x = cbr(3, layer(1, F.max_pool2d(x, 2, 2)))
    feats.append(x)
    x = cbr(4, layer(2, F.max_pool2d(x, 2, 2)))
    feats.append(x)
    x = cbr(5, layer(3, F.max_pool2d(x, 2, 2)))
    x = cbr(6, layer(4, x))
    feats.append(x)
    return feats
