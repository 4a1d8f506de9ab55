"""CPU fp32 restatement of the ConvNextViT text-line recogniser -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
path (``pdf_table_amd/``) never does.

What it restates (reference file:line):
  ConvNextViT.forward            model/convnext_vit/modeling_convnext_vit.py:38-45 (RGB -> gray, CNN, ViT head)
  ConvNextEncoder                model/convnext_vit/modeling_convnext.py:29-54 (HF ``ConvNextStage`` with a (2,1)
                                 kernel / (2,1) stride down-sampler for stages 1..3, none for stage 0)
  ConvNextModel.forward          model/convnext_vit/modeling_convnext.py:97-131 (``last_hidden_state`` is the raw encoder
                                 output: the final LayerNorm only feeds the unused pooled output)
  ViTForSTR.forward_features     model/convnext_vit/modeling_vit.py:31-103 (patch embedding, position embeddings WITHOUT
                                 the class token's slot, encoder, final LayerNorm)
  ViTForSTR.forward              model/convnext_vit/modeling_vit.py:131-143 (three 75-token chunks stitched to 201 tokens:
                                 [:69] of chunk 0, [6:-6] of chunk 1, [6:] of chunk 2; Linear(192 -> 7644))
  OCRRecognitionPreprocessor     model/ocr_recognition/processor_ocr_recognition.py:44-62,94-113 (keep-ratio resize to
                                 32 x 804, three 300-px chunks at 252-px steps)
  OCRRecognitionPostProcessor    model/ocr_recognition/processor_ocr_recognition.py:127-164 (arg-max, CTC collapse, the
                                 vocabulary starts at index 2 when do_chunking)
The HF building blocks are third-party (``transformers``, unpinned in the reference's requirements.txt:19; 5.15.0 in this
image): ConvNextEmbeddings / ConvNextLayer / ConvNextStage / ConvNextLayerNorm (models/convnext/modeling_convnext.py) and
ViTPatchEmbeddings / ViTLayer / ViTAttention / ViTMLP (models/vit/modeling_vit.py); their published arithmetic is
restated below.  Pinned by ``tests/golden/convnext_vit.npz``, which ``tests/golden/make_golden.py`` produced by driving the
reference's own ``ConvNextViT`` module's sub-modules in the order its forward does (the forward itself raises under
transformers 5.15: ``ViTModel.get_head_mask`` is gone).

State-dict names: the checkpoint era of the reference (transformers 4.x: ``vitstr.vit.encoder.layer.N.attention.attention
.query`` ...) is the canonical form; ``canonical_state_dict`` also accepts the 5.x names (``vitstr.vit.layers.N.attention
.q_proj`` ...) the installed modules declare.
"""
from __future__ import annotations

import re

import numpy as np
import torch
import torch.nn.functional as F

DEPTHS = (3, 3, 8, 3)
DIMS = (96, 192, 256, 512)
VIT_DIM, VIT_HEADS, VIT_LAYERS, VIT_FF, VIT_TOKENS = 192, 3, 12, 768, 75
NUM_LABELS = 7644
TARGET_H, TARGET_W, CHUNK_W, CHUNK_STEP = 32, 804, 300, 252

_V5_TO_V4 = [
    (r"vit\.layers\.(\d+)\.attention\.q_proj\.", r"vit.encoder.layer.\1.attention.attention.query."),
    (r"vit\.layers\.(\d+)\.attention\.k_proj\.", r"vit.encoder.layer.\1.attention.attention.key."),
    (r"vit\.layers\.(\d+)\.attention\.v_proj\.", r"vit.encoder.layer.\1.attention.attention.value."),
    (r"vit\.layers\.(\d+)\.attention\.o_proj\.", r"vit.encoder.layer.\1.attention.output.dense."),
    (r"vit\.layers\.(\d+)\.mlp\.fc1\.", r"vit.encoder.layer.\1.intermediate.dense."),
    (r"vit\.layers\.(\d+)\.mlp\.fc2\.", r"vit.encoder.layer.\1.output.dense."),
    (r"vit\.layers\.(\d+)\.layernorm_", r"vit.encoder.layer.\1.layernorm_"),
]


def canonical_state_dict(sd):
    out = {}
    for k, v in sd.items():
        k = k.replace("recognizer.", "").replace("module.", "")          # modeling_ocr_recognition.py:108-111
        for pat, rep in _V5_TO_V4:
            k = re.sub(pat, rep, k)
        out[k] = v
    return out


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def cnn_forward_fp32(sd, gray):
    """gray fp32 [B,1,32,W] -> NHWC features [B, 1, W/4, 512] (ConvNextModel.last_hidden_state, channels last here)."""
    p = "cnn_model."
    x = F.conv2d(gray, sd[p + "embeddings.patch_embeddings.weight"], sd[p + "embeddings.patch_embeddings.bias"], stride=4)
    x = _ln(x.permute(0, 2, 3, 1), sd, p + "embeddings.layernorm", 1e-6)            # NHWC from here on
    for i, (depth, dim) in enumerate(zip(DEPTHS, DIMS)):
        q = f"{p}encoder.stages.{i}."
        if i > 0:                                                                   # ConvNextStage.downsampling_layer
            x = _ln(x, sd, q + "downsampling_layer.0", 1e-6)
            x = F.conv2d(x.permute(0, 3, 1, 2), sd[q + "downsampling_layer.1.weight"], sd[q + "downsampling_layer.1.bias"],
                         stride=(2, 1)).permute(0, 2, 3, 1)
        for j in range(depth):                                                      # ConvNextLayer.forward
            lq = f"{q}layers.{j}."
            y = F.conv2d(x.permute(0, 3, 1, 2), sd[lq + "dwconv.weight"], sd[lq + "dwconv.bias"], padding=3, groups=dim)
            y = _ln(y.permute(0, 2, 3, 1), sd, lq + "layernorm", 1e-6)
            y = F.gelu(F.linear(y, sd[lq + "pwconv1.weight"], sd[lq + "pwconv1.bias"]))
            y = F.linear(y, sd[lq + "pwconv2.weight"], sd[lq + "pwconv2.bias"])
            x = x + sd[lq + "layer_scale_parameter"] * y
    return x


def vit_features_fp32(sd, feats):
    """feats NHWC [B,1,75,512] -> [B,75,192] (ViTForSTR.forward_features)."""
    p = "vitstr.vit."
    B = feats.shape[0]
    x = F.linear(feats.reshape(B, VIT_TOKENS, 512), sd[p + "embeddings.patch_embeddings.projection.weight"].reshape(VIT_DIM, 512),
                 sd[p + "embeddings.patch_embeddings.projection.bias"])
    x = x + sd[p + "embeddings.position_embeddings"][:, 1:, :]
    hd = VIT_DIM // VIT_HEADS
    for l in range(VIT_LAYERS):
        q = f"{p}encoder.layer.{l}."
        h = _ln(x, sd, q + "layernorm_before", 1e-12)
        qq = F.linear(h, sd[q + "attention.attention.query.weight"], sd[q + "attention.attention.query.bias"])
        kk = F.linear(h, sd[q + "attention.attention.key.weight"], sd[q + "attention.attention.key.bias"])
        vv = F.linear(h, sd[q + "attention.attention.value.weight"], sd[q + "attention.attention.value.bias"])
        sh = lambda t: t.view(B, VIT_TOKENS, VIT_HEADS, hd).transpose(1, 2)
        att = torch.softmax(torch.matmul(sh(qq), sh(kk).transpose(2, 3)) * hd ** -0.5, dim=-1)
        ctx = torch.matmul(att, sh(vv)).transpose(1, 2).reshape(B, VIT_TOKENS, VIT_DIM)
        x = F.linear(ctx, sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"]) + x
        h = _ln(x, sd, q + "layernorm_after", 1e-12)
        h = F.gelu(F.linear(h, sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"]))
        x = F.linear(h, sd[q + "output.dense.weight"], sd[q + "output.dense.bias"]) + x
    return _ln(x, sd, p + "layernorm", 1e-12)


def stitch_chunks(seq):
    """[3n, 75, E] -> [n, 201, E] (modeling_vit.py:133-138)."""
    b3, _, e = seq.shape
    ap = seq.view(b3 // 3, 3, VIT_TOKENS, e)
    out = torch.ones(b3 // 3, 201, e, dtype=seq.dtype)
    out[:, :69] = ap[:, 0, :69]
    out[:, 69:69 + 63] = ap[:, 1, 6:-6]
    out[:, 69 + 63:] = ap[:, 2, 6:]
    return out


def convnext_vit_forward_fp32(sd, x):
    """x fp32 [3n, 3, 32, 300] RGB in [0,1] (or [3n,1,32,300] gray) -> logits [n, 201, 7644]."""
    sd = canonical_state_dict(sd)
    if x.shape[1] == 3:
        x = x[:, 0:1] * 0.2989 + x[:, 1:2] * 0.5870 + x[:, 2:3] * 0.1140            # modeling_convnext_vit.py:40
    seq = vit_features_fp32(sd, cnn_forward_fp32(sd, x))
    feat = stitch_chunks(seq)
    return F.linear(feat, sd["vitstr.classifier.weight"], sd["vitstr.classifier.bias"])


def chunk_preprocess(crop):
    """crop uint8 HxWx3 RGB -> fp32 [3,3,32,300] (processor_ocr_recognition.py:94-113, do_chunking)."""
    from .crnn import keepratio_resize
    img = torch.FloatTensor(keepratio_resize(crop, TARGET_H, TARGET_W))
    chunks = [img[:, CHUNK_STEP * i:CHUNK_STEP * i + CHUNK_W] for i in range(3)]
    data = torch.cat(chunks, 0).view(3, TARGET_H, CHUNK_W, 3) / 255.
    return data.permute(0, 3, 1, 2)


def greedy_text(logits, vocab):
    """OCRRecognitionPostProcessor.__call__ (:147-164) for do_chunking: label_mapping[i + 2] = vocab[i]; class 1 has no
    entry (KeyError in the reference too)."""
    preds = np.asarray(torch.argmax(torch.softmax(torch.as_tensor(logits), dim=-1), -1))
    out = []
    for row in preds:
        last, s = 0, []
        for p in row.tolist():
            if p != last and p != 0:
                if p < 2:
                    raise KeyError(p)
                s.append(vocab[p - 2])
            last = p
        out.append("".join(s))
    return out
