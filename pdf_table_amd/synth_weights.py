"""Seeded synthetic checkpoints with the reference's ``state_dict`` layouts.

There is no network on the build or GPU boxes, so every net on the hot path is exercised with
deterministic random-init weights whose key names and shapes are exactly the ones the reference
modules declare (checked with ``load_state_dict(strict=True)`` by ``tests/golden/make_golden.py``):

* DB-ResNet18 ``DBModel``      -- /root/reference/src/pdftable/model/db_net/dbnet.py:715-728
* ``CRNN``                     -- /root/reference/src/pdftable/model/crnn/modeling_crnn.py:36-90

BatchNorm running statistics and affine terms are randomised (gamma, var in [0.5, 1.5]; beta, mean
in [-0.1, 0.1]) so that BN folding in the weight packer is really exercised.
numpy's Generator is used (not torch's RNG) so the stream does not depend on the torch build.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import numpy as np
import torch

__all__ = ["db_resnet18_state_dict", "crnn_state_dict", "CRNN_NUM_CLASSES", "lore_dla34_state_dict",
           "lore_processor_state_dict", "LORE_HEADS", "picodet_state_dict", "LCNET_CONFIG", "PICODET_STANDIN", "lore_wireless_state_dict", "db_nas_state_dict", "pplcnet_state_dict", "convnext_vit_state_dict", "mtl_tabnet_backbone_state_dict", "mtl_tabnet_decoder_state_dict"]

CRNN_NUM_CLASSES = 7644  # crnn/modeling_crnn.py:90


class _Gen:
    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)
        self.sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def put(self, name, arr):
        self.sd[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))

    def conv(self, name, cout, cin, kh, kw, bias=False, gain=2.0):
        fan_in = cin * kh * kw
        std = math.sqrt(gain / fan_in)
        self.put(name + ".weight", self.rng.standard_normal((cout, cin, kh, kw)) * std)
        if bias:
            self.put(name + ".bias", self.rng.uniform(-0.1, 0.1, (cout,)))

    def convT(self, name, cin, cout, kh, kw, gain=2.0, quad_noise=None, bias=None):
        # nn.ConvTranspose2d weight layout: [Cin, Cout, kh, kw]; always has a bias in the reference.
        # quad_noise: make the kh*kw sub-pixel kernels nearly equal (base * (1 + quad_noise * N(0,1))), like a
        # trained up-sampling head, so that the output is blob-like instead of per-pixel speckle.
        std = math.sqrt(gain / cin)
        if quad_noise is None:
            w = self.rng.standard_normal((cin, cout, kh, kw)) * std
        else:
            base = self.rng.standard_normal((cin, cout, 1, 1)) * std
            w = base * (1.0 + quad_noise * self.rng.standard_normal((cin, cout, kh, kw)))
        self.put(name + ".weight", w)
        b = self.rng.uniform(-0.1, 0.1, (cout,))
        self.put(name + ".bias", b if bias is None else np.full((cout,), bias))

    def bn(self, name, c):
        self.put(name + ".weight", self.rng.uniform(0.5, 1.5, (c,)))
        self.put(name + ".bias", self.rng.uniform(-0.1, 0.1, (c,)))
        self.put(name + ".running_mean", self.rng.uniform(-0.1, 0.1, (c,)))
        self.put(name + ".running_var", self.rng.uniform(0.5, 1.5, (c,)))
        self.sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def linear(self, name, cout, cin, bias=True, scale=1.0):
        k = scale / math.sqrt(cin)
        self.put(name + ".weight", self.rng.uniform(-k, k, (cout, cin)))
        if bias:
            self.put(name + ".bias", self.rng.uniform(-k, k, (cout,)))

    def lstm(self, name, nin, nh, scale=1.0):
        k = scale / math.sqrt(nh)
        for sfx in ("", "_reverse"):
            self.put(f"{name}.weight_ih_l0{sfx}", self.rng.uniform(-k, k, (4 * nh, nin)))
            self.put(f"{name}.weight_hh_l0{sfx}", self.rng.uniform(-k, k, (4 * nh, nh)))
            self.put(f"{name}.bias_ih_l0{sfx}", self.rng.uniform(-k, k, (4 * nh,)))
            self.put(f"{name}.bias_hh_l0{sfx}", self.rng.uniform(-k, k, (4 * nh,)))


def db_resnet18_state_dict(seed: int = 0, with_thresh_branch: bool = True, head_bias: float = 6.0, text_signal=False):
    """state_dict of ``DBModel`` (ResNet-18 backbone + ``SegDetector`` decoder, adaptive=True).

    Key order follows module registration order in dbnet.py:260-336 (backbone) and :488-586
    (decoder; the ``thresh`` branch exists in checkpoints but is never run in eval, :635-638).
    """
    g = _Gen(seed)
    g.conv("backbone.conv1", 64, 3, 7, 7)
    g.bn("backbone.bn1", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            stride = 2 if (bi == 0 and li > 1) else 1
            g.conv(p + ".conv1", planes, inpl, 3, 3)
            g.bn(p + ".bn1", planes)
            # residual branch scaled down a little so 8 blocks do not blow the dynamic range
            g.conv(p + ".conv2", planes, planes, 3, 3, gain=1.0)
            g.bn(p + ".bn2", planes)
            if stride != 1 or inpl != planes:
                g.conv(p + ".downsample.0", planes, inpl, 1, 1, gain=1.0)
                g.bn(p + ".downsample.1", planes)
            inpl = planes
    # decoder (bias=False for the plain convs: SegDetector(..., bias=False) dbnet.py:494)
    g.conv("decoder.in5", 256, 512, 1, 1)
    g.conv("decoder.in4", 256, 256, 1, 1)
    g.conv("decoder.in3", 256, 128, 1, 1)
    g.conv("decoder.in2", 256, 64, 1, 1)
    for n in ("out5.0", "out4.0", "out3.0", "out2"):
        g.conv("decoder." + n, 64, 256, 3, 3)
    g.conv("decoder.binarize.0", 64, 256, 3, 3)
    g.bn("decoder.binarize.1", 64)
    g.convT("decoder.binarize.3", 64, 64, 2, 2, quad_noise=0.05)
    g.bn("decoder.binarize.4", 64)
    # small gain on the last layer: logits with std ~5-8 (soft edges like a trained DB head) instead of ~50,
    # and a bias calibrated on the synthetic pages so that ~6 % of a page is "text" (~150 blob components)
    g.convT("decoder.binarize.6", 64, 1, 2, 2, gain=0.02, quad_noise=0.05, bias=head_bias)
    if with_thresh_branch:
        g.conv("decoder.thresh.0", 64, 256, 3, 3)
        g.bn("decoder.thresh.1", 64)
        g.convT("decoder.thresh.3", 64, 64, 2, 2)
        g.bn("decoder.thresh.4", 64)
        g.convT("decoder.thresh.6", 64, 1, 2, 2)
    if text_signal:
        _db_text_signal(g.sd, *text_signal) if isinstance(text_signal, tuple) else _db_text_signal(g.sd)
    return g.sd


def _db_text_signal(sd, gain: float = 24.0, level: float = 0.69, a1: float = 2.0, a2: float = 1.25):
    """Turn channel 0 of a random-init DBModel checkpoint into a hand-built text-line detector for the synthetic pages
    (dark glyph blobs on white), so that the detector's OWN boxes can feed the recogniser in bench.py and the box
    post-process sees a realistic number of boxes per page (~75 on the synthetic pages, whose generator draws ~76 lines).
    All other channels keep their random weights, so operand statistics, arithmetic and memory traffic of every layer are
    what they were; only the rows that produce channel 0 (plus helper channel 1 between the two convs of layer1's
    blocks), and the columns of the last layer, are set:

      conv1 (7x7 s2) ch 0   x0 = 1 - mean gray / 255 over the window ("darkness"), BN as an affine identity; max-pool dilates
      layer1.0              S1 = min(1, a1 * mean3x3(x0)): conv1 ch 0 -> u = relu(1 - a1 * mean3x3(x0)), ch 1 -> v = x0;
                            conv2 ch 0 = 1 - u - v, and the block's residual adds x0 back.  Glyph holes and word gaps
                            saturate to 1, 1-2 px table rules stay below 0.3
      layer1.1              S2 = min(1, a2 * mean1x3(S1)) the same way: closes gaps of <= 2 cells (8 px) along a line
      in2 -> out2 -> fuse[192] -> binarize   centre taps, BN identities, the two transposed convs replicate 2x2:
                            logit = gain * (S2 - level) - logit(0.3), i.e. the DB bitmap threshold 0.3 cuts at S2 = level
    """
    import math as _m

    def bn_identity(name, c, beta=0.0):
        sd[name + ".weight"][c] = 1.0
        sd[name + ".bias"][c] = beta
        sd[name + ".running_mean"][c] = 0.0
        sd[name + ".running_var"][c] = 1.0 - 1e-5      # sqrt(var + eps) == 1

    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    # sum over taps and channels of w * x with x = (pix/255 - mean_c) / std_c  ==  mean(mean_c) - gray/255
    sd["backbone.conv1.weight"][0] = (-(std / 3.0 / 49.0)).view(3, 1, 1).expand(3, 7, 7)
    bn_identity("backbone.bn1", 0, beta=1.0 - float(mean.mean()))
    for bi, (a, taps) in enumerate(((a1, [(dy, dx) for dy in range(3) for dx in range(3)]), (a2, [(1, dx) for dx in range(3)]))):
        p = f"backbone.layer1.{bi}"
        w1, w2 = sd[p + ".conv1.weight"], sd[p + ".conv2.weight"]
        w1[0] = 0.0
        w1[1] = 0.0
        for dy, dx in taps:
            w1[0, 0, dy, dx] = -a / len(taps)
        w1[1, 0, 1, 1] = 1.0
        bn_identity(p + ".bn1", 0, beta=1.0)
        bn_identity(p + ".bn1", 1)
        w2[0] = 0.0
        w2[0, 0, 1, 1] = -1.0
        w2[0, 1, 1, 1] = -1.0
        bn_identity(p + ".bn2", 0, beta=1.0)
    for n in ("in5", "in4", "in3", "in2"):
        sd[f"decoder.{n}.weight"][0] = 0.0
    sd["decoder.in2.weight"][0, 0, 0, 0] = 1.0
    for name, cin in (("decoder.out2.weight", 0), ("decoder.binarize.0.weight", 192)):
        sd[name][0] = 0.0
        sd[name][0, cin, 1, 1] = 1.0
    bn_identity("decoder.binarize.1", 0)
    sd["decoder.binarize.3.weight"][:, 0] = 0.0
    sd["decoder.binarize.3.weight"][0, 0] = 1.0
    sd["decoder.binarize.3.bias"][0] = 0.0
    bn_identity("decoder.binarize.4", 0)
    sd["decoder.binarize.6.weight"][:] = 0.0
    sd["decoder.binarize.6.weight"][0, 0] = gain
    sd["decoder.binarize.6.bias"][:] = -gain * level + _m.log(0.3 / 0.7)


def crnn_state_dict(seed: int = 0, num_classes: int = CRNN_NUM_CLASSES, conditioned: bool = False):
    """state_dict of ``CRNN`` (crnn/modeling_crnn.py:40-90); every conv has a bias.

    ``conditioned=True`` (seed 1: bench.py's and the end-to-end fixture's recogniser): the classifier ``cls.weight`` is replaced by the rows of
    ``data/crnn_synth_classifier.npz``, fitted by ``tools/fit_crnn_classifier.py`` on this very net's BiLSTM features of the generator's text lines
    (k-means clusters of the frames as classes, the largest one as CTC blank; every other row zero).  A random 7644-way classifier has near-tied
    logits on every frame -- any 16-bit arithmetic flips token ids there --, the fitted one has the margin distribution of a trained recogniser
    (the file keeps both distributions).  A workload device that memorises its pages, not a recogniser."""
    g = _Gen(seed)
    g.conv("conv0.0", 64, 1, 3, 3, bias=True)
    g.bn("conv0.1", 64)
    g.conv("conv1.0", 128, 64, 3, 3, bias=True)
    g.bn("conv1.1", 128)
    g.conv("conv2.0", 256, 128, 3, 3, bias=True)
    g.bn("conv2.1", 256)
    g.conv("conv2.3", 256, 256, 3, 3, bias=True)
    g.bn("conv2.4", 256)
    g.conv("conv3.0", 512, 256, 3, 3, bias=True)
    g.bn("conv3.1", 512)
    g.conv("conv3.3", 512, 512, 3, 3, bias=True)
    g.bn("conv3.4", 512)
    g.conv("conv4.0", 512, 512, 2, 1, bias=True)
    g.bn("conv4.1", 512)
    # larger-than-default scales so that the recurrent state really depends on the input and the logits are O(1)
    # with a time-varying arg-max (default nn.LSTM/Linear init gives |logit| ~ 0.02 and a constant arg-max)
    g.lstm("rnn.0.rnn", 512, 256, scale=2.0)
    g.linear("rnn.0.embedding", 256, 512, scale=4.0)
    g.lstm("rnn.1.rnn", 256, 256, scale=3.0)
    g.linear("rnn.1.embedding", 512, 512, scale=4.0)
    g.linear("cls", num_classes, 512, bias=False, scale=4.0)
    if conditioned:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "crnn_synth_classifier.npz"))
        if int(z["seed"]) != seed or num_classes != CRNN_NUM_CLASSES:
            raise ValueError(f"crnn_synth_classifier.npz was fitted on crnn_state_dict(seed={int(z['seed'])}), not seed={seed}")
        w = torch.zeros(num_classes, 512, dtype=g.sd["cls.weight"].dtype)
        w[torch.from_numpy(z["ids"].astype(np.int64))] = torch.from_numpy(z["rows"]).to(w.dtype)
        g.sd["cls.weight"] = w
    return g.sd


def convnext_vit_state_dict(seed: int = 0, num_labels: int = CRNN_NUM_CLASSES):
    """state_dict of ``ConvNextViT`` (convnext_vit/modeling_convnext_vit.py:20-36: ConvNext depths [3, 3, 8, 3], widths
    [96, 192, 256, 512], one input channel; ViT 12 x (192-d, 3 heads, 768 MLP) over 75 tokens; Linear(192 -> 7644)) in the
    key names of the checkpoint era the reference was written for (transformers 4.x ``vit.encoder.layer.N.attention.attention
    .query`` ...).  Layer-scale values are O(0.3) like a trained net (the 1e-6 initial value would make every block a no-op)."""
    g = _Gen(seed)
    r = g.rng

    def ln(name, c):
        g.put(name + ".weight", r.uniform(0.7, 1.3, (c,)))
        g.put(name + ".bias", r.uniform(-0.1, 0.1, (c,)))

    p = "cnn_model."
    g.put(p + "embeddings.patch_embeddings.weight", r.standard_normal((96, 1, 4, 4)) * 1.5)
    g.put(p + "embeddings.patch_embeddings.bias", r.uniform(-0.1, 0.1, (96,)))
    ln(p + "embeddings.layernorm", 96)
    dims, depths = (96, 192, 256, 512), (3, 3, 8, 3)
    for i, (d, dep) in enumerate(zip(dims, depths)):
        q = f"{p}encoder.stages.{i}."
        if i > 0:
            ln(q + "downsampling_layer.0", dims[i - 1])
            g.conv(q + "downsampling_layer.1", d, dims[i - 1], 2, 1, bias=True, gain=1.0)
        for j in range(dep):
            lq = f"{q}layers.{j}."
            g.put(lq + "layer_scale_parameter", r.uniform(0.2, 0.5, (d,)))
            g.put(lq + "dwconv.weight", r.standard_normal((d, 1, 7, 7)) * (1.0 / 7.0))
            g.put(lq + "dwconv.bias", r.uniform(-0.1, 0.1, (d,)))
            ln(lq + "layernorm", d)
            g.linear(lq + "pwconv1", 4 * d, d, scale=1.5)
            g.linear(lq + "pwconv2", d, 4 * d, scale=1.5)
    ln(p + "layernorm", 512)
    p = "vitstr.vit."
    g.put(p + "embeddings.cls_token", r.standard_normal((1, 1, 192)) * 0.02)
    g.put(p + "embeddings.position_embeddings", r.standard_normal((1, 76, 192)) * 0.2)
    g.conv(p + "embeddings.patch_embeddings.projection", 192, 512, 1, 1, bias=True, gain=1.0)
    for l in range(12):
        q = f"{p}encoder.layer.{l}."
        for n in ("query", "key", "value"):
            g.linear(q + "attention.attention." + n, 192, 192, scale=2.0)
        g.linear(q + "attention.output.dense", 192, 192, scale=1.0)
        ln(q + "layernorm_before", 192)
        ln(q + "layernorm_after", 192)
        g.linear(q + "intermediate.dense", 768, 192, scale=1.5)
        g.linear(q + "output.dense", 192, 768, scale=1.0)
    ln(p + "layernorm", 192)
    g.linear("vitstr.classifier", num_labels, 192, scale=4.0)
    return g.sd


def mtl_tabnet_backbone_state_dict(seed: int = 0):
    """state_dict of ``TableResNetExtra(layers=[1, 2, 5, 3], gcb_config=...)`` (table/mtl_tabnet/table_resnet_extra.py:205-247,
    mtl_tabnet_config.py:41-53): the backbone of MtlTabNet / TableMaster.  Used by the oracle's golden only so far."""
    g = _Gen(seed)
    r = g.rng

    def block(p, cin, planes, gcb):
        g.conv(p + ".conv1", planes, cin, 3, 3)
        g.bn(p + ".bn1", planes)
        g.conv(p + ".conv2", planes, planes, 3, 3, gain=1.0)
        g.bn(p + ".bn2", planes)
        if gcb:
            hid = int(planes * 0.0625)
            q = p + ".context_block"
            g.conv(q + ".conv_mask", 1, planes, 1, 1, bias=True, gain=4.0)
            g.conv(q + ".channel_add_conv.0", hid, planes, 1, 1, bias=True)
            g.put(q + ".channel_add_conv.1.weight", r.uniform(0.7, 1.3, (hid, 1, 1)))
            g.put(q + ".channel_add_conv.1.bias", r.uniform(-0.1, 0.1, (hid, 1, 1)))
            g.conv(q + ".channel_add_conv.3", planes, hid, 1, 1, bias=True)
        if cin != planes:
            g.conv(p + ".downsample.0", planes, cin, 1, 1, gain=1.0)
            g.bn(p + ".downsample.1", planes)

    g.conv("conv1", 64, 3, 3, 3)
    g.bn("bn1", 64)
    g.conv("conv2", 128, 64, 3, 3)
    g.bn("bn2", 128)
    inpl = 128
    for i, (planes, n, gcb) in enumerate(zip((256, 256, 512, 512), (1, 2, 5, 3), (False, True, True, True)), start=1):
        for j in range(n):
            block(f"layer{i}.{j}", inpl, planes, gcb and j == 0)
            inpl = planes
        g.conv(f"conv{i + 2}", planes, planes, 3, 3)
        g.bn(f"bn{i + 2}", planes)
    return g.sd


def mtl_tabnet_decoder_state_dict(seed: int = 0, num_classes: int = 43, num_classes_cell: int = 60, d_model: int = 512, d_ff: int = 2024,
                                  n_layers: int = 3, table_signal=None):
    """state_dict of ``MtlTabNetDecoder`` (table/mtl_tabnet/master_decoder.py:194-262, configuration of mtl_tabnet_config.py:59-77:
    N = 3, d_model 512, 8 heads, d_ff 2024) without the two ``pe`` buffers (position tables, recomputed).  Oracle golden only."""
    g = _Gen(seed)
    r = g.rng

    def layer(p):
        for a in ("self_attn", "src_attn"):
            for i in range(4):
                g.linear(f"{p}.{a}.linears.{i}", d_model, d_model, scale=1.5)
        g.linear(p + ".feed_forward.w_1", d_ff, d_model, scale=1.5)
        g.linear(p + ".feed_forward.w_2", d_model, d_ff, scale=1.0)
        for i in range(3):
            g.put(f"{p}.sublayer.{i}.norm.weight", r.uniform(0.7, 1.3, (d_model,)))
            g.put(f"{p}.sublayer.{i}.norm.bias", r.uniform(-0.1, 0.1, (d_model,)))

    for i in range(n_layers - 1):
        layer(f"layers.{i}")
    layer("cls_layer.0")
    layer("bbox_layer.0")
    layer("cell_layer.0")
    g.linear("cls_fc", num_classes, d_model, scale=6.0)
    g.linear("bbox_fc.0", 4, d_model, scale=2.0)
    g.linear("cell_fc", num_classes_cell, d_model, scale=6.0)
    g.put("norm.weight", r.uniform(0.7, 1.3, (d_model,)))
    g.put("norm.bias", r.uniform(-0.1, 0.1, (d_model,)))
    g.put("embedding.lut.weight", r.standard_normal((num_classes, d_model)) * 0.05)
    g.put("embedding_cell.lut.weight", r.standard_normal((num_classes_cell, d_model)) * 0.05)
    g.linear("cell_input_fc", d_model, 2 * d_model, scale=1.5)
    if table_signal is not None:
        _mtl_table_signal(g.sd, d_model, **table_signal)
    return g.sd


def table_master_decoder_state_dict(seed: int = 0, num_classes: int = 43, table_signal=None):
    """state_dict of ``TableMasterDecoder`` (table/mtl_tabnet/master_decoder.py:532-563, table_master_config.py:45-64): MtlTabNetDecoder's tensors without
    the cell-content decoder (cell_layer, cell_fc, cell_input_fc, embedding_cell) -- the same seeded values as ``mtl_tabnet_decoder_state_dict``."""
    ncc = 8 if table_signal is None else max([table_signal["sos_cell"], table_signal["eos_cell"]] + list(table_signal["cell"])) + 2
    sd = mtl_tabnet_decoder_state_dict(seed=seed, num_classes=num_classes, num_classes_cell=ncc, table_signal=table_signal)
    return {k: v for k, v in sd.items() if not k.startswith(("cell_", "embedding_cell."))}


def mtl_table_signal_from(conv, rows_until: int = 150, cell_text: str = "Varible%"):
    """the ids ``mtl_tabnet_decoder_state_dict(table_signal=...)`` needs, taken from a label convertor (pdf_table_amd.mtl_stage.MtlTabNetConvertor):
    one table row is ``<tr> <td></td> <td colspan="2" > </td> <eb></eb> </tr>`` (two cells with content, one empty), rows repeat until the decoded
    position passes ``rows_until``, then ``</tbody>`` and <EOS>; every content cell reads ``cell_text`` (distinct characters) and <EOS>."""
    c = conv.char2idx
    row = [c["<tr>"], c["<td></td>"], c["<td"], c[' colspan="2"'] if ' colspan="2"' in c else c['colspan="2"'], c[">"], c["</td>"], c["<eb></eb>"], c["</tr>"]]
    assert len(set(cell_text)) == len(cell_text), "the cell chain is first-order: characters must be distinct"
    return dict(sos=conv.start_idx, eos=conv.end_idx, open_=c["<tbody>"], row=row, close=c["</tbody>"], rows_until=rows_until,
                sos_cell=conv.start_idx_cell, eos_cell=conv.end_idx_cell, cell=[conv.char2idx_cell[ch] for ch in cell_text])


def _mtl_table_signal(sd, d, sos, eos, open_, row, close, rows_until, sos_cell, eos_cell, cell):
    """Hand-built channels that make the seeded MtlTabNet decoders emit a TABLE instead of noise (the analogue of _db_text_signal for the detector;
    VERDICT r03: with plain random weights the structure decoder never emits a cell tag, so bench.py's mtl_tabnet leg never ran the cell-content
    decoder).  A few residual-stream dimensions are reserved: no sub-layer writes to them (the rows of every attention / feed-forward OUTPUT
    projection and their biases are zeroed there), so they carry ``embedding * sqrt(d) + positional encoding`` unchanged to the final LayerNorm:
      * one dimension per token of the chain holds a one-hot of the PREVIOUS token (amplitude 64: it dominates the LayerNorm statistics); the
        classifier row of the token that follows reads it -- a first-order chain <SOS> <tbody> (<tr> ... </tr>)* </tbody> <EOS>;
      * after ``</tr>`` the choice between another ``<tr>`` and ``</tbody>`` reads the position: dimension 320 carries sin(pos / 316.2) (the model's
        own positional encoding, monotone up to position 496) and dimension 510 a constant sin(rows_until / 316.2) every chain token embeds;
        ``</tbody>`` wins once the first exceeds the second -- the LayerNorm's mean and variance cancel in that comparison, so every table closes at
        the first row boundary past ``rows_until`` whatever its image;
      * the cell-content decoder gets the same construction (its input projection passes the reserved dimensions of the character embedding
        through): <SOS> c0 c1 ... <EOS>.
    Everything else stays the seeded random network: the kernels see the same shapes and value ranges, attention over the source features included."""
    import math
    rt = math.sqrt(d)
    chain = [sos, open_] + list(row) + [close]
    cchain = [sos_cell] + list(cell)
    assert len(set(chain)) == len(chain) and len(set(cchain)) == len(cchain)
    tdim = {t: 384 + 2 * i for i, t in enumerate(chain)}            # even dimensions: the positional encoding there is sin(pos * 1e-3..) ~ 0
    cdim = {t: 420 + 2 * i for i, t in enumerate(cchain)}
    D_POS, D_REF = 320, 510
    reserved = sorted(set(tdim.values()) | set(cdim.values()) | {D_POS, D_REF})
    assert max(reserved) < d and len(reserved) == len(tdim) + len(cdim) + 2
    layers = [k[:-len(".self_attn.linears.3.weight")] for k in sd if k.endswith(".self_attn.linears.3.weight")]
    for p in layers:
        for name in (".self_attn.linears.3", ".src_attn.linears.3", ".feed_forward.w_2"):
            sd[p + name + ".weight"][reserved, :] = 0.0
            sd[p + name + ".bias"][reserved] = 0.0
    A, B, K = 64.0, 6.0, 4000.0
    c0 = math.sin(rows_until / 316.2277660168379)                    # 1 / div_term of dimension 320: 10000 ** (320 / 512)
    emb = sd["embedding.lut.weight"]
    emb[:, reserved] = 0.0
    for t in chain:
        emb[t, tdim[t]] = A / rt
        emb[t, D_REF] = c0 / rt
    w, b = sd["cls_fc.weight"], sd["cls_fc.bias"]
    w *= 0.25                                                        # the random logits stay, well below the chain's
    w[:, reserved] = 0.0
    nxt = {sos: open_, open_: row[0], close: eos}
    for a_, b_ in zip(row[:-1], row[1:]):
        nxt[a_] = b_
    for prev, n in nxt.items():
        w[n, tdim[prev]] += B
    last = row[-1]
    w[row[0], tdim[last]] += B                                       # after </tr>: <tr> ...
    w[close, tdim[last]] += B                                        # ... or </tbody>, decided by the position:
    g_, be = sd["norm.weight"], sd["norm.bias"]
    w[close, D_POS] += K / float(g_[D_POS])                          # K * [(h[320] - beta) / gamma - (h[510] - beta') / gamma'] = K * (x[320] - x[510]) / std
    w[close, D_REF] -= K / float(g_[D_REF])
    b[close] += -K * float(be[D_POS]) / float(g_[D_POS]) + K * float(be[D_REF]) / float(g_[D_REF])
    # cell-content decoder
    cemb = sd["embedding_cell.lut.weight"]
    cemb[:, reserved] = 0.0
    for t in cchain:
        cemb[t, cdim[t]] = A / rt
    wi, bi = sd["cell_input_fc.weight"], sd["cell_input_fc.bias"]
    wi[reserved, :] = 0.0
    bi[reserved] = 0.0
    for r_ in reserved:
        wi[r_, r_] = 1.0                                             # from the character-embedding half of cat(embedding, structure state)
    wc = sd["cell_fc.weight"]
    wc *= 0.25
    wc[:, reserved] = 0.0
    for a_, b_ in zip(cchain, cchain[1:] + [eos_cell]):
        wc[b_, cdim[a_]] += B


# --------------------------------------------------------------------------------------------------------------------
# Lore (table structure): DLA-34 + DCN detector and the logical-location processor
# --------------------------------------------------------------------------------------------------------------------
LORE_HEADS = {"hm": 2, "st": 8, "wh": 8, "ax": 256, "cr": 256, "reg": 2}     # lore/modeling_lore.py:89
DLA34_LEVELS = [1, 1, 1, 2, 2, 1]                                            # center_net/modeling_centernet.py:405-409
DLA34_CHANNELS = [16, 32, 64, 128, 256, 512]


def lore_dla34_state_dict(seed: int = 0, hm_bias=(-6.0, -5.0), hm_gain: float = 0.5, cell_half=(10.0, 6.0), dcn_gain: float = 0.1):
    """state_dict of ``get_dla_dcn(34, heads)`` = ``DLASeg`` (lore/lore_dla_34.py:137-206) on ``dla34``
    (center_net/modeling_centernet.py:274-409, incl. the unused 1000-way ``fc``).

    Deformable convs get non-zero offset/mask weights (the reference initialises them to zero, dcnv2.py:66-67;
    trained checkpoints are not) with offsets of about a third of a pixel.  ``hm``/``wh`` biases are chosen so that a random
    net yields a table-like number of cell centres and corner points (a few hundred each on a 256 x 256 map) with
    well-formed quads (corner i = centre - wh[2i:2i+2])."""
    g = _Gen(seed)
    ch = DLA34_CHANNELS
    g.conv("base.base_layer.0", ch[0], 3, 7, 7)
    g.bn("base.base_layer.1", ch[0])
    g.conv("base.level0.0", ch[0], ch[0], 3, 3)
    g.bn("base.level0.1", ch[0])
    g.conv("base.level1.0", ch[1], ch[0], 3, 3)
    g.bn("base.level1.1", ch[1])

    def block(p, cin, cout):
        g.conv(p + ".conv1", cout, cin, 3, 3)
        g.bn(p + ".bn1", cout)
        g.conv(p + ".conv2", cout, cout, 3, 3, gain=1.0)
        g.bn(p + ".bn2", cout)

    def tree(p, levels, cin, cout, level_root, root_dim=0):
        if root_dim == 0:
            root_dim = 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            block(p + ".tree1", cin, cout)
            block(p + ".tree2", cout, cout)
            g.conv(p + ".root.conv", cout, root_dim, 1, 1)
            g.bn(p + ".root.bn", cout)
        else:
            tree(p + ".tree1", levels - 1, cin, cout, False, 0)
            tree(p + ".tree2", levels - 1, cout, cout, False, root_dim + cout)
        if cin != cout:
            g.conv(p + ".project.0", cout, cin, 1, 1, gain=1.0)
            g.bn(p + ".project.1", cout)

    for lvl in range(2, 6):
        tree(f"base.level{lvl}", DLA34_LEVELS[lvl], ch[lvl - 1], ch[lvl], lvl > 2)
    g.conv("base.fc", 1000, ch[5], 1, 1, bias=True)

    def dcn(p, cin, cout):
        g.bn(p + ".actf.0", cout)
        g.conv(p + ".conv", cout, cin, 3, 3, bias=True)
        # 27 = 18 offsets + 9 mask logits, ~N(0, 0.3): sub-pixel offsets, so the bilinear path is exercised.  Larger
        # offsets make a RANDOM net ill-conditioned (its features are uncorrelated from pixel to pixel, so a 1e-5 px
        # offset change moves the output by 1e-3 -- measured; trained features are smooth)
        g.conv(p + ".conv.conv_offset_mask", 27, cin, 3, 3, bias=True, gain=dcn_gain)

    def up(p, c, f):
        # fill_up_weights (lore_dla_34.py:53-62): bilinear kernel, same for every channel; perturbed per channel
        # here so that the per-channel weights are really read
        k = 2 * f
        ff = math.ceil(k / 2)
        cc = (2 * ff - 1 - ff % 2) / (2.0 * ff)
        w = np.zeros((c, 1, k, k))
        for i in range(k):
            for j in range(k):
                w[:, 0, i, j] = (1 - math.fabs(i / ff - cc)) * (1 - math.fabs(j / ff - cc))
        w *= g.rng.uniform(0.9, 1.1, (c, 1, 1, 1))
        g.put(p + ".weight", w)

    def ida(p, o, channels, up_f):
        for i in range(1, len(channels)):
            dcn(f"{p}.proj_{i}", channels[i], o)
            dcn(f"{p}.node_{i}", o, o)
            up(f"{p}.up_{i}", o, int(up_f[i]))

    # DLAUp.__init__ (lore_dla_34.py:115-126) with channels [64,128,256,512], scales [1,2,4,8]
    channels = ch[2:]
    in_channels = list(channels)
    scales = np.array([1, 2, 4, 8])
    for i in range(len(channels) - 1):
        j = -i - 2
        ida(f"dla_up.ida_{i}", channels[j], in_channels[j:], scales[j:] // scales[j])
        scales[j + 1:] = scales[j]
        in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]
    ida("ida_up", ch[2], ch[2:5], [1, 2, 4])

    for h, k in LORE_HEADS.items():
        g.conv(f"{h}.0", 256, ch[2], 3, 3, bias=True)
        if h == "hm":
            g.conv(f"{h}.2", k, 256, 1, 1, bias=False, gain=hm_gain)
            # zero-sum weights per output: the hidden units are post-ReLU (positive, similar means), so this centres
            # both heat maps on the bias instead of on a random per-channel offset
            w = g.sd[f"{h}.2.weight"]
            g.sd[f"{h}.2.weight"] = w - w.mean(dim=1, keepdim=True)
            g.put(f"{h}.2.bias", np.asarray(hm_bias, dtype=np.float64).reshape(k))   # (cell centres, corners)
        elif h in ("wh", "st"):
            g.conv(f"{h}.2", k, 256, 1, 1, bias=False, gain=0.05)
            hw, hh = cell_half
            g.put(f"{h}.2.bias", np.array([hw, hh, -hw, hh, -hw, -hh, hw, -hh]))
        elif h == "reg":
            g.conv(f"{h}.2", k, 256, 1, 1, bias=False, gain=0.02)
            g.put(f"{h}.2.bias", np.array([0.5, 0.5]))
        else:
            g.conv(f"{h}.2", k, 256, 1, 1, bias=True, gain=0.5)
    return g.sd


def lore_processor_state_dict(seed: int = 0, layers: int = 4, stacking_layers: int = 4, conditioned: bool = False):
    """state_dict of ``LoreProcessModel`` (lore/lore_processor.py:399-437): stacker + tsfm_axis + 2 position tables.
    ``conditioned`` (seed 3 only): the stacker's last Linear(256 -> 4) comes from ``data/lore_synth_processor_head.npz``, fitted by
    ``tools/fit_lore_processor.py`` so that the logical locations of the end-to-end fixture's tables are near-integers like a trained model's (a seeded
    random layer emits arbitrary reals: two of the fixture's three tables had a location 5.6e-5 from the .5 rounding boundary) -- a workload device."""
    g = _Gen(seed)

    def norm(p, d):
        g.put(p + ".alpha", g.rng.uniform(0.8, 1.2, (d,)))
        g.put(p + ".bias", g.rng.uniform(-0.1, 0.1, (d,)))

    def transformer(p, in_size, hid, out_size, n):
        g.linear(p + ".linear", hid, in_size)
        # PositionalEncoder buffer (lore_processor.py:260-274) -- registered, never used in forward
        pe = np.zeros((900, hid))
        for pos in range(900):
            for i in range(0, hid, 2):
                pe[pos, i] = math.sin(pos / (10000 ** ((2 * i) / hid)))
                pe[pos, i + 1] = math.cos(pos / (10000 ** ((2 * (i + 1)) / hid)))
        g.put(p + ".encoder.pe.pe", pe[None])
        for li in range(n):
            q = f"{p}.encoder.layers.{li}"
            norm(q + ".norm_1", hid)
            norm(q + ".norm_2", hid)
            for nm in ("q_linear", "v_linear", "k_linear", "out"):
                g.linear(f"{q}.attn.{nm}", hid, hid, scale=2.0)
            g.linear(q + ".ff.linear_1", 2048, hid, scale=1.5)
            g.linear(q + ".ff.linear_2", hid, 2048, scale=1.5)
        norm(p + ".encoder.norm", hid)
        g.linear(p + ".decoder.linear.0", hid, hid, scale=2.0)
        g.linear(p + ".decoder.linear.2", out_size, hid, scale=2.0)

    g.linear("stacker.logi_encoder.0", 256, 4)
    g.linear("stacker.logi_encoder.2", 256, 256)
    transformer("stacker.tsfm", 512, 256, 4, stacking_layers)
    transformer("tsfm_axis", 256, 256, 4, layers)
    g.put("x_position_embeddings.weight", g.rng.standard_normal((256, 256)))
    g.put("y_position_embeddings.weight", g.rng.standard_normal((256, 256)))
    if conditioned:
        if seed != 3 or layers != 4 or stacking_layers != 4:
            raise ValueError("lore_synth_processor_head.npz was fitted on lore_processor_state_dict(seed=3) with 4 + 4 layers")
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lore_synth_processor_head.npz"))
        g.sd["stacker.tsfm.decoder.linear.2.weight"] = torch.from_numpy(z["weight"].astype(np.float32))
        g.sd["stacker.tsfm.decoder.linear.2.bias"] = torch.from_numpy(z["bias"].astype(np.float32))
    return g.sd


# --------------------------------------------------------------------------------------------------------------------
# PicoDet layout detector: LCNet x1.0 backbone + 4-level CSP-PAN + PicoHead (shared cls/reg tower)
# --------------------------------------------------------------------------------------------------------------------
LCNET_CONFIG = {   # k, in_c, out_c, stride, use_se -- picodet/lcnet.py:25-46
    "blocks2": [[3, 16, 32, 1, False]],
    "blocks3": [[3, 32, 64, 2, False], [3, 64, 64, 1, False]],
    "blocks4": [[3, 64, 128, 2, False], [3, 128, 128, 1, False]],
    "blocks5": [[3, 128, 256, 2, False]] + [[5, 256, 256, 1, False]] * 5,
    "blocks6": [[5, 256, 512, 2, True], [5, 512, 512, 1, True]],
}
PICODET_STANDIN = dict(neck_channels=128, num_convs=4, reg_max=7, strides=(8, 16, 32, 64))


def picodet_state_dict(seed: int = 0, num_classes: int = 5, cls_bias: float = -6.5, head_gain: float = 0.012, table_head: bool = False):
    """state_dicts of the in-tree PicoDet parts, keys prefixed ``backbone.`` / ``neck.`` / ``head.`` like
    ``PicoDet`` (picodet/modeling_picodet.py:31-36): ``LCNet(scale=1.0, feature_maps=[3,4,5])`` (lcnet.py:159-259),
    ``CSPPAN(in_channels=[128,256,512], out_channels=128, kernel_size=5, num_features=4)`` (csp_pan.py:233-347) and
    ``PicoHead(PicoFeat(128, 128, num_fpn_stride=4, num_convs=4, share_cls_reg=True), fpn_stride=[8,16,32,64],
    reg_max=7)`` (pico_head.py:56-167,966-1072).  The reference runs an ONNX export whose exact hyper-parameters are
    not in the tree: this configuration is the ASSUMED picodet_lcnet_x1_0 layout model.

    ``table_head=True`` (seed 4, five classes: bench.py's layout checkpoint): the stride-64 branch of the head -- four depthwise / pointwise pairs and
    ``head_cls3`` -- is replaced by the tensors of ``data/picodet_synth_table_head.npz``, fitted by ``tools/fit_layout_head.py`` on the features this very
    backbone and neck give on pages 0..511 of ``synth_pages.make_page`` so that class "table" fires on those pages' tables (a workload device that memorises
    its pages, not a layout detector -- see that script); the table logit of the other three levels is switched off."""
    g = _Gen(seed)
    nc = PICODET_STANDIN["neck_channels"]

    def conv_bn(p, cout, cin, k, groups=1, norm="bn", gain=2.0):
        g.conv(p + ".conv", cout, cin // groups, k, k, gain=gain)
        g.bn(p + "." + norm, cout)

    conv_bn("backbone.conv1", 16, 3, 3)
    for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
        for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
            p = f"backbone.{blk}.{i}"
            conv_bn(p + ".dw_conv", cin, cin, k, groups=cin, gain=2.0)
            if se:
                g.conv(p + ".se.conv1", cin // 4, cin, 1, 1, bias=True)
                g.conv(p + ".se.conv2", cin, cin // 4, 1, 1, bias=True)
            conv_bn(p + ".pw_conv", cout, cin, 1)

    def dp(p, c, k=5):       # DPModule (csp_pan.py:56-105)
        g.conv(p + ".dwconv", c, 1, k, k, gain=2.0)
        g.bn(p + ".bn1", c)
        g.conv(p + ".pwconv", c, c, 1, 1)
        g.bn(p + ".bn2", c)

    def csp(p, cin, cout):   # CSPLayer with one depthwise DarknetBottleneck (csp_pan.py:160-209)
        mid = cout // 2
        conv_bn(p + ".main_conv", mid, cin, 1)
        conv_bn(p + ".short_conv", mid, cin, 1)
        conv_bn(p + ".final_conv", cout, 2 * mid, 1)
        conv_bn(p + ".blocks.0.conv1", mid, mid, 1)
        dp(p + ".blocks.0.conv2", mid)

    for i, c in enumerate((128, 256, 512)):
        conv_bn(f"neck.conv_t.convs.{i}", nc, c, 1)
    dp("neck.first_top_conv", nc)
    dp("neck.second_top_conv", nc)
    for i in range(2):
        csp(f"neck.top_down_blocks.{i}", 2 * nc, nc)
    for i in range(2):
        dp(f"neck.downsamples.{i}", nc)
        csp(f"neck.bottom_up_blocks.{i}", 2 * nc, nc)

    nout = num_classes + 4 * (PICODET_STANDIN["reg_max"] + 1)
    for s in range(4):
        for i in range(PICODET_STANDIN["num_convs"]):
            conv_bn(f"head.conv_feat.cls_conv_dw{s}_{i}", nc, nc, 5, groups=nc, norm="norm", gain=2.0)
            conv_bn(f"head.conv_feat.cls_conv_pw{s}_{i}", nc, nc, 1, norm="norm")
    for lvl in (3, 4, 5, 6):
        g.put(f"head.p{lvl}_feat.scale_reg", np.ones((1,)))
    g.put("head.distribution_project.project", np.linspace(0, PICODET_STANDIN["reg_max"], PICODET_STANDIN["reg_max"] + 1))
    for s in range(4):
        # a random hardswish net is not scale-normalised on page-like inputs (tower outputs have std ~25): the small gain
        # brings the class logits to std ~1.5 so that, with the negative bias, a few tens of anchors per page pass 0.5
        g.conv(f"head.head_cls{s}", nout, nc, 1, 1, bias=False, gain=head_gain)
        b = g.rng.uniform(-0.5, 0.5, (nout,))
        b[:num_classes] += cls_bias            # few positives, like a trained detector's prior (pico_head.py:1041)
        g.put(f"head.head_cls{s}.bias", b)
    if table_head:
        import os
        import torch
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "picodet_synth_table_head.npz"))
        fit_seed, fit_ncls, table, level = (int(v) for v in z["meta"][:4])
        if (seed, num_classes) != (fit_seed, fit_ncls):
            raise ValueError(f"the fitted table head belongs to picodet_state_dict(seed={fit_seed}, num_classes={fit_ncls})")
        for k in z.files:
            if k != "meta":
                if tuple(g.sd[k].shape) != tuple(z[k].shape):
                    raise ValueError(f"picodet_synth_table_head.npz: {k} has shape {z[k].shape}, the checkpoint's is {tuple(g.sd[k].shape)}")
                g.sd[k] = torch.from_numpy(np.ascontiguousarray(z[k])).to(g.sd[k].dtype)
        for lvl in range(4):
            if lvl != level:
                g.sd[f"head.head_cls{lvl}.weight"][table] = 0.0
                g.sd[f"head.head_cls{lvl}.bias"][table] = -12.0
    return g.sd


def lore_wireless_state_dict(seed: int = 0, hm_bias=(-5.0, -4.5), cell_half=(10.0, 6.0)):
    """state_dict of ``LoreDetectModel`` (lore/lore_detector.py:155-286): the ResNet-18-style 'wireless' detector --
    7x7/s2 stem, max-pool, four 2-block stages (64, 128, 256, 256; every stage strided; block convs WITH bias), four
    ConvTranspose2d(256, 256, 4, stride 2, padding 1, no bias) + BN up-samplers with 1x1 lateral ``adaption`` convs, and
    heads of four 3x3 (->64) + ReLU then 1x1 (``reg``: one 3x3)."""
    g = _Gen(seed)
    g.conv("conv1", 64, 3, 7, 7)
    g.bn("bn1", 64)
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 256), start=1):
        for bi in range(2):
            p = f"layer{li}.{bi}"
            g.conv(p + ".conv1", planes, inpl, 3, 3, bias=True)
            g.bn(p + ".bn1", planes)
            g.conv(p + ".conv2", planes, planes, 3, 3, bias=True, gain=1.0)
            g.bn(p + ".bn2", planes)
            if bi == 0:                                    # stride 2 in every stage (:181-188) -> always a downsample
                g.conv(p + ".downsample.0", planes, inpl, 1, 1, gain=1.0)
                g.bn(p + ".downsample.1", planes)
            inpl = planes
    for name, cin in (("adaption3", 256), ("adaption2", 128), ("adaption1", 64), ("adaption0", 64), ("adaptionU1", 256)):
        g.conv(name, 256, cin, 1, 1, gain=1.0)
    for i in range(1, 5):
        # nn.ConvTranspose2d weight layout [Cin, Cout, 4, 4]; each output pixel sums 4 taps per input channel
        std = math.sqrt(2.0 / (256 * 4))
        g.put(f"deconv_layers{i}.0.weight", g.rng.standard_normal((256, 256, 4, 4)) * std)
        g.bn(f"deconv_layers{i}.1", 256)
    for h, k in LORE_HEADS.items():
        n3 = 1 if h == "reg" else 4
        cin = 256
        for j in range(n3):
            g.conv(f"{h}.{2 * j}", 64, cin, 3, 3, bias=True)
            cin = 64
        last = f"{h}.{2 * n3}"
        if h == "hm":
            g.conv(last, k, 64, 1, 1, bias=False, gain=0.5)
            w = g.sd[last + ".weight"]
            g.sd[last + ".weight"] = w - w.mean(dim=1, keepdim=True)
            g.put(last + ".bias", np.asarray(hm_bias, dtype=np.float64).reshape(k))
        elif h in ("wh", "st"):
            g.conv(last, k, 64, 1, 1, bias=False, gain=0.05)
            hw, hh = cell_half
            g.put(last + ".bias", np.array([hw, hh, -hw, hh, -hw, -hh, hw, -hh]))
        elif h == "reg":
            g.conv(last, k, 64, 1, 1, bias=False, gain=0.02)
            g.put(last + ".bias", np.array([0.5, 0.5]))
        else:
            g.conv(last, k, 64, 1, 1, bias=True, gain=0.5)
    return g.sd


def db_nas_state_dict(seed: int = 0, with_thresh_branch: bool = True, head_bias: float = -1.5):
    """state_dict of ``DBNasModel`` (db_net/dbnet.py:693-712; blocks from pdf_table_amd.dbnas_arch).  Key order follows
    module registration order (proxyless.py:99-161, layers.py:690-728 / :110-143 / :477-479, dbnet.py:370-391, 35-99)."""
    from .dbnas_arch import dbnas_blocks, INNER_CHANNELS, DW_KERNEL, WIDTH_STAGES, INPUT_CHANNEL
    g = _Gen(seed)

    def prelu(name):
        g.put(name + ".weight", g.rng.uniform(0.1, 0.4, (1,)))

    def dwconv(name, c, k, gain=2.0):
        g.put(name + ".weight", g.rng.standard_normal((c, 1, k, k)) * math.sqrt(gain / (k * k)))

    g.conv("backbone.first_conv.0", INPUT_CHANNEL, 3, 3, 3)
    g.bn("backbone.first_conv.1", INPUT_CHANNEL)
    for bi, b in enumerate(dbnas_blocks()):
        p = f"backbone.blocks.{bi}.mobile_inverted_conv"
        if b["kind"] == "se":
            g.conv(p + ".fc1", b["squeeze"], b["cin"], 1, 1, bias=True)
            g.conv(p + ".fc2", b["cin"], b["squeeze"], 1, 1, bias=True)
            continue
        g.conv(p + ".inverted_bottleneck.conv", b["mid"], b["cin"], 1, 1, gain=1.0)
        g.bn(p + ".inverted_bottleneck.bn", b["mid"])
        prelu(p + ".inverted_bottleneck.act")
        if b["kind"] == "rep":
            for ri, k in enumerate(b["sizes"]):
                dwconv(f"{p}.rep_conv.{ri}.conv", b["mid"], k, gain=2.0 / len(b["sizes"]))
                g.bn(f"{p}.rep_conv.{ri}.bn", b["mid"])
            prelu(p + ".act")
        else:
            dwconv(p + ".depth_conv.conv", b["mid"], b["sizes"][0])
            g.bn(p + ".depth_conv.bn", b["mid"])
            prelu(p + ".depth_conv.act")
        # the projection is linear (no activation after its BN); a residual branch at half gain keeps 20 blocks in range
        g.conv(p + ".point_conv.conv", b["cout"], b["mid"], 1, 1, gain=0.08 if b["shortcut"] else 0.3)
        g.bn(p + ".point_conv.bn", b["cout"])
    ic = INNER_CHANNELS
    for name, cin in zip(("in5", "in4", "in3", "in2"), WIDTH_STAGES[::-1]):
        g.conv("decoder." + name, ic, cin, 1, 1, gain=0.5)
    q = ic // 4
    dwconv("decoder.binarize.0.depthwise", ic, DW_KERNEL)
    g.bn("decoder.binarize.0.bn1", ic)
    g.conv("decoder.binarize.0.pointwise", q, ic, 1, 1)
    g.bn("decoder.binarize.1", q)

    def dwpw_t(name, cout, gain, bias=None):      # DwPwConvTranspose (dbnet.py:75-99): depthwise ConvT 2x2 s2 + BN + ReLU + 1x1
        base = g.rng.standard_normal((q, 1, 1, 1)) * math.sqrt(2.0)
        g.put(name + ".depthwise.weight", base * (1.0 + 0.05 * g.rng.standard_normal((q, 1, 2, 2))))
        g.put(name + ".depthwise.bias", g.rng.uniform(-0.1, 0.1, (q,)))
        g.bn(name + ".bn1", q)
        g.conv(name + ".pointwise", cout, q, 1, 1, bias=True, gain=gain)
        if bias is not None:
            g.put(name + ".pointwise.bias", np.full((cout,), bias))

    dwpw_t("decoder.binarize.3", q, 2.0)
    g.bn("decoder.binarize.4", q)
    dwpw_t("decoder.binarize.6", 1, 100.0, bias=head_bias)
    if with_thresh_branch:     # exists in checkpoints (adaptive=True), never run in eval (dbnet.py:470-473)
        g.conv("decoder.thresh.0", q, ic, DW_KERNEL, DW_KERNEL)
        g.bn("decoder.thresh.1", q)
        g.convT("decoder.thresh.3", q, q, 2, 2)
        g.bn("decoder.thresh.4", q)
        g.convT("decoder.thresh.6", q, 1, 2, 2)
    return g.sd


def pplcnet_state_dict(seed: int = 0, class_num: int = 2, logit_gain: float = 4.0):
    """state_dict of ``PPLCNet(scale=1.0, class_num=...)`` (model/cls/cls_pp_lcnet.py:164-260): conv1, blocks2..6 (the
    LCNet x1.0 table LCNET_CONFIG == NET_CONFIG :54-66), last_conv (512 -> 1280, no bias), fc.  stride_list does not
    change any shape."""
    g = _Gen(seed)

    def conv_bn(p, cout, cin, k, groups=1, gain=2.0):
        g.conv(p + ".conv", cout, cin // groups, k, k, gain=gain)
        g.bn(p + ".bn", cout)

    conv_bn("conv1", 16, 3, 3)
    for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
        for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
            p = f"{blk}.{i}"
            conv_bn(p + ".dw_conv", cin, cin, k, groups=cin, gain=2.0)
            if se:
                g.conv(p + ".se.conv1", cin // 4, cin, 1, 1, bias=True)
                g.conv(p + ".se.conv2", cin, cin // 4, 1, 1, bias=True)
            conv_bn(p + ".pw_conv", cout, cin, 1)
    g.conv("last_conv", 1280, 512, 1, 1)
    g.linear("fc", class_num, 1280, scale=logit_gain)
    return g.sd


def conditioned_state_dicts():
    """THE synthetic checkpoint set of bench.py's timed step and of the end-to-end fixture (tests/golden/e2e_page.npz) -- one set, so that what is
    timed is what the parity assertions run on: the detector with the hand-built text channel (its boxes feed the recogniser),
    the recogniser with the fitted classifier (trained-like arg-max margins), the layout net with the head branch fitted to the generator's pages (its
    "table" regions feed the table stage), the Lore detector with the heat-map bias that yields cells and DCN offsets of ~0.07 px (a random DLA-34
    with 16 stacked DCNs at the default ~0.3 px is chaotic in fp32 itself: DESIGN.md numerics), the Lore processor with the last Linear fitted to
    near-integer logical locations on the fixture's tables (tools/fit_lore_processor.py)."""
    return {"db": db_resnet18_state_dict(seed=0, text_signal=True), "crnn": crnn_state_dict(seed=1, conditioned=True),
            "pico": picodet_state_dict(seed=4, num_classes=5, table_head=True),
            "lore": lore_dla34_state_dict(seed=2, dcn_gain=0.02, hm_bias=(-2.0, -2.0), hm_gain=0.25), "proc": lore_processor_state_dict(seed=3, conditioned=True)}
