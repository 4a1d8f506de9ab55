"""ORACLE (test infrastructure only -- see oracle/__init__.py): CPU restatement of Lore's logical-location processor.

Follows /root/reference/src/pdftable/model/lore/lore_processor.py: ``norm`` = Norm.forward :126-131 (unbiased std,
eps added to the std), ``mha`` = MultiHeadAttention.forward :205-226 with attention :134-163, ``encoder_layer`` =
EncoderLayer.forward :296-313 (pre-norm; the attention map of the last layer is computed and dropped in eval),
``transformer`` = Transformer.forward :92-114 (Encoder.forward :48-61 applies neither positional encoding nor the
final Norm), Decoder :64-78, ``stacker`` = Stacker.forward :369-396, ``processor_forward`` =
LoreProcessModel.forward :465-514 (evaluation branch).

PINNED by tests/golden/lore_processor.npz (outputs of the reference LoreProcessModel with seeded weights).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

HEADS, D_MODEL = 8, 256


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def norm(sd, p, x, eps=1e-6):
    return sd[p + ".alpha"] * (x - x.mean(dim=-1, keepdim=True)) / (x.std(dim=-1, keepdim=True) + eps) + sd[p + ".bias"]


def mha(sd, p, x):
    bs, n, _ = x.shape
    dk = D_MODEL // HEADS
    k = _lin(sd, p + ".k_linear", x).view(bs, n, HEADS, dk).transpose(1, 2)
    q = _lin(sd, p + ".q_linear", x).view(bs, n, HEADS, dk).transpose(1, 2)
    v = _lin(sd, p + ".v_linear", x).view(bs, n, HEADS, dk).transpose(1, 2)
    att = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).contiguous().view(bs, n, D_MODEL)
    return _lin(sd, p + ".out", o)


def encoder_layer(sd, p, x):
    x = x + mha(sd, p + ".attn", norm(sd, p + ".norm_1", x))
    x2 = norm(sd, p + ".norm_2", x)
    return x + _lin(sd, p + ".ff.linear_2", F.relu(_lin(sd, p + ".ff.linear_1", x2)))


def transformer(sd, p, x, n_layers):
    x = _lin(sd, p + ".linear", x)
    for i in range(n_layers):
        x = encoder_layer(sd, f"{p}.encoder.layers.{i}", x)
    x = F.relu(_lin(sd, p + ".decoder.linear.0", x))
    return F.relu(_lin(sd, p + ".decoder.linear.2", x))


def processor_forward(sd: Dict[str, torch.Tensor], feat: torch.Tensor, dets: Optional[torch.Tensor] = None,
                      tsfm_layers: int = 4, stacking_layers: int = 4):
    """feat f32 [1,n,256] (+ dets int64 [1,n,8] when wiz_2dpe) -> (logic_axis [1,n,4], stacked_axis [1,n,4])."""
    if dets is not None:
        xe, ye = sd["x_position_embeddings.weight"], sd["y_position_embeddings.weight"]
        feat = feat + xe[dets[:, :, 0]] + ye[dets[:, :, 1]] + xe[dets[:, :, 2]] + ye[dets[:, :, 5]]
    logic = transformer(sd, "tsfm_axis", feat, tsfm_layers)
    le = F.relu(_lin(sd, "stacker.logi_encoder.2", F.relu(_lin(sd, "stacker.logi_encoder.0", logic))))
    stacked = transformer(sd, "stacker.tsfm", torch.cat((le, feat), dim=2), stacking_layers)
    return logic, stacked
