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
  MtlTabNetDecoder (test time)  model/table/mtl_tabnet/master_decoder.py:17-188 (Embeddings, SubLayerConnection, FeedForward,
                               self_attention, MultiHeadAttention(Cell), DecoderLayer(Cell), PositionalEncoding), :354-517
                               (decode_test, greedy_forward, forward_test: structure tokens, cell boxes, cell content)
Pinned by ``tests/golden/mtl_tabnet_backbone.npz`` / ``mtl_tabnet_decoder.npz`` (the reference's own ``TableResNetExtra`` and
``MtlTabNetDecoder`` modules on seeded weights, ``tests/golden/make_golden.py::gen_mtl_tabnet_*``).  Not restated: the label
convertor / post-processor (master_convertor.py, master_post_processor.py).
  TableMasterDecoder (test time) master_decoder.py:532-645: the same two shared layers + structure-token / box layers WITHOUT the cell-content
                               decoder; greedy_forward :599-608 always runs max_seq_len + 1 steps (no <EOS> stop).  ``table_master_decode``,
                               pinned by ``tests/golden/table_master_decoder.npz`` (the reference's own module, ``gen_table_master``).
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


# ---------------------------------------------------------------------------------------------------------------------
# Decoder (model/table/mtl_tabnet/master_decoder.py) -- restated as the reference runs it at test time: the whole prefix is
# decoded again at every step (greedy_forward :463-491); pinned by tests/golden/mtl_tabnet_decoder.npz, which the reference's
# own MtlTabNetDecoder produced (make_golden.py::gen_mtl_tabnet_decoder).  No engine path exists yet.
# ---------------------------------------------------------------------------------------------------------------------
import math

HEADS = 8


def positional_encoding(x):
    """PositionalEncoding.forward (:166-188) for [b, L, d] (or [b, c, h, w] feature maps, flattened to [b, h*w, c])."""
    if x.dim() > 3:
        b, c, h, w = x.shape
        x = x.view(b, c, h * w).permute(0, 2, 1)
    d = x.shape[-1]
    pos = torch.arange(0, x.shape[1]).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2).float() * -math.log(10000.0) / d)
    pe = torch.zeros(x.shape[1], d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return x + pe.unsqueeze(0)


def _mha(sd, p, q_in, kv_in, mask, shared_kv=False):
    """MultiHeadAttention.forward (:75-96); shared_kv: MultiHeadAttentionCell (:117-144: keys / values of ONE sample for all rows)."""
    nb, d = q_in.shape[0], q_in.shape[-1]
    dk = d // HEADS
    lin = lambda i, t: F.linear(t, sd[f"{p}.linears.{i}.weight"], sd[f"{p}.linears.{i}.bias"])
    q = lin(0, q_in).view(nb, -1, HEADS, dk).transpose(1, 2)
    kb = 1 if shared_kv else nb
    k = lin(1, kv_in).contiguous().view(kb, -1, HEADS, dk).transpose(1, 2)
    v = lin(2, kv_in).contiguous().view(kb, -1, HEADS, dk).transpose(1, 2)
    score = torch.matmul(q, k.transpose(-2, -1) / math.sqrt(dk))          # :65 (the KEYS are scaled)
    if mask is not None:
        score = score.masked_fill(mask == 0, -6.55e4)
    x = torch.matmul(F.softmax(score, dim=-1), v)
    x = x.transpose(1, 2).contiguous().view(nb, -1, d)
    return lin(3, x)


def decoder_layer(sd, p, x, feature, tgt_mask, cell=False):
    """DecoderLayer / DecoderLayerCell.forward (:99-114, :147-164): pre-norm residual blocks, eval mode (no dropout)."""
    ln = lambda i, t: F.layer_norm(t, (t.shape[-1],), sd[f"{p}.sublayer.{i}.norm.weight"], sd[f"{p}.sublayer.{i}.norm.bias"])
    h = ln(0, x)
    x = x + _mha(sd, p + ".self_attn", h, h, tgt_mask)
    x = x + _mha(sd, p + ".src_attn", ln(1, x), feature, None, shared_kv=cell)
    h = ln(2, x)
    return x + F.linear(F.relu(F.linear(h, sd[p + ".feed_forward.w_1.weight"], sd[p + ".feed_forward.w_1.bias"])),
                        sd[p + ".feed_forward.w_2.weight"], sd[p + ".feed_forward.w_2.bias"])


def _tgt_mask(tokens, pad):
    """make_mask (:264-278): padding mask & causal mask, [b, 1, L, L]."""
    L = tokens.shape[1]
    return (tokens != pad).unsqueeze(1).unsqueeze(3).byte() & torch.tril(torch.ones((L, L), dtype=torch.uint8))


def _embed(sd, p, tokens):
    w = sd[p + ".lut.weight"]
    return positional_encoding(F.embedding(tokens, w) * math.sqrt(w.shape[1]))


def decode_step(sd, tokens, feature, cfg, decode_cell):
    """decode_test (:354-461): structure logits [b, L, classes], boxes [b, L, 4] and, with decode_cell, the greedy cell-content
    decode of every sample (a list of [cells, steps, cell classes] logits; torch.zeros(1) for a sample without cell tags)."""
    norm = lambda t: F.layer_norm(t, (t.shape[-1],), sd["norm.weight"], sd["norm.bias"])
    x = _embed(sd, "embedding", tokens)
    mask = _tgt_mask(tokens, cfg["pad"])
    for i in range(cfg["N"] - 1):
        x = decoder_layer(sd, f"layers.{i}", x, feature, mask)
    tag = F.linear(norm(decoder_layer(sd, "cls_layer.0", x, feature, mask)), sd["cls_fc.weight"], sd["cls_fc.bias"])
    box = torch.sigmoid(F.linear(norm(decoder_layer(sd, "bbox_layer.0", x, feature, mask)), sd["bbox_fc.0.weight"], sd["bbox_fc.0.bias"]))
    cells = []
    if decode_cell:
        dec = torch.max(F.softmax(tag, dim=-1), dim=-1)[1]
        bmask = (dec == cfg["idx_tag_cell"][0]) | (dec == cfg["idx_tag_cell"][1])
        for b in range(tokens.shape[0]):
            if int(bmask[b].sum()) == 0:
                cells.append(torch.zeros(1))
                continue
            x_i = x[b, bmask[b]].unsqueeze(1)                                          # [cells, 1, d]
            inp = torch.full((x_i.shape[0], 1), cfg["sos_cell"], dtype=torch.long)
            eos = torch.full((x_i.shape[0], 1), cfg["eos_cell"], dtype=torch.long)
            out_cell = None
            for step in range(cfg["max_len_cell"] + 1):
                xc = _embed(sd, "embedding_cell", inp)
                cm = _tgt_mask(inp, cfg["pad_cell"])
                xi = F.linear(torch.cat((xc, x_i.expand(-1, step + 1, -1)), -1), sd["cell_input_fc.weight"], sd["cell_input_fc.bias"])
                parts = [norm(decoder_layer(sd, "cell_layer.0", xi[s:s + 64], feature[b:b + 1], cm[s:s + 64], cell=True))
                         for s in range(0, inp.shape[0], 64)]                                # batch_size_cell = 64 (:245)
                out_cell = F.linear(torch.cat(parts, 0), sd["cell_fc.weight"], sd["cell_fc.bias"])
                nxt = torch.max(F.softmax(out_cell, dim=-1), dim=-1)[1][:, -1:]
                if torch.equal(nxt, eos):
                    break
                inp = torch.cat([inp, nxt], dim=1)
            cells.append(out_cell)
    return tag, box, cells


def greedy_decode(sd, feature, cfg):
    """forward_test / greedy_forward (:463-517): feature [b, h*w, d] AFTER the positional encoding."""
    b = feature.shape[0]
    inp = torch.full((b, 1), cfg["sos"], dtype=torch.long)
    eos = torch.full((b, 1), cfg["eos"], dtype=torch.long)
    for i in range(cfg["max_len"] + 1):
        if i == cfg["max_len"]:
            return decode_step(sd, inp, feature, cfg, True)
        out, _, _ = decode_step(sd, inp, feature, cfg, False)
        nxt = torch.max(F.softmax(out, dim=-1), dim=-1)[1][:, -1:]
        if torch.equal(nxt, eos):
            return decode_step(sd, inp, feature, cfg, True)
        inp = torch.cat([inp, nxt], dim=1)


def table_master_decode(sd, feature, cfg):
    """TableMasterDecoder.forward_test / greedy_forward (master_decoder.py:599-632): max_len + 1 decode steps whatever is emitted (the convertor cuts
    at the first <EOS>), every step over the whole prefix; -> (structure logits [b, max_len + 1, classes], boxes [b, max_len + 1, 4])."""
    b = feature.shape[0]
    inp = torch.full((b, 1), cfg["sos"], dtype=torch.long)
    out = box = None
    for _ in range(cfg["max_len"] + 1):
        out, box, _ = decode_step(sd, inp, feature, cfg, False)
        nxt = torch.max(F.softmax(out, dim=-1), dim=-1)[1][:, -1:]
        inp = torch.cat([inp, nxt], dim=1)
    return out, box


def mtl_preprocess(img, size=480):
    """The test pipeline of mtl_tabnet_config.py:136-160 on one uint8 HxWx3 image (taken as it is: mmcv's loader hands an ndarray on):
    TableResize(keep_ratio, long_size) -- model/table/lgpma/lgpma_preprocess.py:1067-1093: the long side becomes `size`, the short side
    int(size / long * short) in Python floats, cv2.resize INTER_LINEAR (restated in oracle/db_pre.py, parity unpinned like every cv2
    call) -- TablePad(size x size, pad_val 0) :1128-1182, ToTensorOCR (/ 255, CHW) and NormalizeOCR(0.5, 0.5).
    -> (fp32 [3, size, size], img_meta with the keys the label convertor reads)."""
    import numpy as np
    from . import db_pre
    h, w = img.shape[:2]
    fw, fh = float(w), float(h)
    if fw < fh:
        fw, fh = size / fh * fw, size
    else:
        fh, fw = size / fw * fh, size
    nw, nh = int(fw), int(fh)
    r = db_pre.cv2_resize_linear_u8(img, nw, nh)
    pad = np.zeros((size, size, 3), dtype=np.uint8)
    pad[:nh, :nw] = r
    x = torch.from_numpy(pad).permute(2, 0, 1).float().div(255.0)
    x = (x - 0.5) / 0.5
    meta = {"scale_factor": (nh / h, nw / w), "pad_shape": (size, size, 3), "ori_shape": tuple(img.shape), "img_shape": (nh, nw, 3)}
    return x, meta
