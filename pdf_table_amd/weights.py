"""Weight packer: reference ``state_dict`` layouts -> the engine's "PTW1" blobs.

Host-side only (PyTorch CPU).  Replaces the reference's ``torch.load`` + ``load_state_dict`` +
``model.half()`` (model/db_net/modeling_db_net.py:53-56, utils/deploy_utils.py:227-240) for the
``predictor_type == "hip"`` path.

What the packer does:
  * folds eval-mode BatchNorm into the preceding conv (float64 fold; weight rounded to bf16 with
    round-to-nearest-even, bias kept fp32) -- every backbone conv of the hot-path nets is Conv->BN->act;
  * re-tiles conv weights to the layout the MFMA kernel streams: ``[N/64][Cin/32][kh*kw][64][32]`` bf16
    (an LDS K-slice is one contiguous 16-byte-aligned run);
  * rewrites ``ConvTranspose2d(k=2, s=2)`` as a 1x1 GEMM with ``N = 4*Cout`` (quadrant-major) for the
    pixel-shuffle epilogue;
  * pads the 7x7 stem to K = [7][8][4] (zero tap / zero channel).

Blob format (little endian): ``b"PTW1"``, ``uint32 n``, then ``n`` records
``name[96] | dtype u32 (0 bf16, 1 f32, 2 i32) | ndim u32 | dims[6] u32 | offset u64 | nbytes u64``,
then 256-byte aligned payloads.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

BN_EPS = 1e-5
_DT = {"bf16": 0, "f32": 1, "i32": 2}

__all__ = ["pack_db_resnet18", "pack_crnn", "pack_lore_dla34", "pack_lore_processor", "pack_picodet", "pack_lore_wireless", "pack_db_nas", "pack_pplcnet", "pack_convnext_vit", "pack_mtl_backbone", "pack_mtl_decoder", "write_blob", "fold_conv_bn", "to_bf16_bits"]


FORMATS = {"bf16": torch.bfloat16, "f16": torch.float16}      # 16-bit storage formats of the engine (csrc/act16.h): PT_PRECISION_BF16* / PT_PRECISION_F16


def to_bf16_bits(t: torch.Tensor, fmt: str = "bf16") -> np.ndarray:
    """fp32 tensor -> uint16 array of the bits of its 16-bit rounding (round to nearest even): bfloat16, or IEEE half for
    ``fmt="f16"`` (saturating at +-65504 like the engine's stores; folded weights are nowhere near)."""
    t = t.detach().to(torch.float32).contiguous()
    if fmt == "f16":
        t = t.clamp(-65504.0, 65504.0)
    return t.to(FORMATS[fmt]).view(torch.int16).numpy().view(np.uint16)


def fold_conv_bn(sd: Dict[str, torch.Tensor], conv: str, bn: Optional[str], transposed: bool = False
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(weight fp32 with BN scale folded in, bias fp32).  ``transposed``: weight is [Cin, Cout, kh, kw]."""
    w = sd[conv + ".weight"].double()
    cout = w.shape[1] if transposed else w.shape[0]
    b = sd[conv + ".bias"].double() if (conv + ".bias") in sd else torch.zeros(cout, dtype=torch.float64)
    if bn is not None:
        scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
        shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale
        w = w * (scale.view(1, -1, 1, 1) if transposed else scale.view(-1, 1, 1, 1))
        b = b * scale + shift
    return w.float(), b.float()


def tile_conv_weight(w: torch.Tensor, fmt: str = "bf16") -> np.ndarray:
    """[N, Cin, kh, kw] fp32 -> 16-bit tiles [N/64][Cin/32][kh*kw][64][32] (bf16 bits, or fp16 bits for ``fmt="f16"``)."""
    n, cin, kh, kw = w.shape
    assert n % 64 == 0 and cin % 32 == 0, (n, cin)
    t = w.reshape(n // 64, 64, cin // 32, 32, kh, kw).permute(0, 2, 4, 5, 1, 3).contiguous()
    return to_bf16_bits(t, fmt).reshape(n // 64, cin // 32, kh * kw, 64, 32)


def split_bf16(t: torch.Tensor, fmt: str = "bf16"):
    """fp32 -> (hi, lo) with hi = round16(t), lo = round16(t - hi), both returned as fp32 tensors."""
    dt = FORMATS[fmt]
    hi = t.to(dt).to(torch.float32)
    lo = (t - hi).to(dt).to(torch.float32)
    return hi, lo


def tile_conv_weight_x3(w: torch.Tensor) -> np.ndarray:
    """BF16X3 tiles: K chunks are [w_hi (for x_hi) | w_hi (for x_lo) | w_lo (for x_hi)] -> [N/64][3*Cin/32][taps][64][32]."""
    hi, lo = split_bf16(w)
    th, tl = tile_conv_weight(hi), tile_conv_weight(lo)
    return np.concatenate([th, th, tl], axis=1)


def tile_conv_weight_f16x2(w: torch.Tensor) -> np.ndarray:
    """F16X2 tiles: ONE fp16 rounding of the weights (RNE), the same tile for the x_hi and the x_lo K chunks ->
    fp16 bits [N/64][2*Cin/32][taps][64][32] (stored under the 16-bit tag of the container)."""
    n, cin, kh, kw = w.shape
    assert n % 64 == 0 and cin % 32 == 0, (n, cin)
    t = w.reshape(n // 64, 64, cin // 32, 32, kh, kw).permute(0, 2, 4, 5, 1, 3).contiguous()
    t = t.to(torch.float16).view(torch.int16).numpy().view(np.uint16).reshape(n // 64, cin // 32, kh * kw, 64, 32)
    return np.concatenate([t, t], axis=1)


class _Blob:
    """One packed model.  ``fmt``: the 16-bit storage format every "bf16"-tagged tensor of the blob is rounded to -- "bf16" (PT_PRECISION_BF16 and,
    with ``x3``, the pair modes) or "f16" (PT_PRECISION_F16: single-pass IEEE half, no pair tiles).  An f16 blob carries the marker tensor
    ``__act_f16__``; the engine refuses a blob whose format is not the one its precision computes in (csrc/common.h pt_model_format_ok)."""

    def __init__(self, x3: bool = True, fmt: str = "bf16"):
        assert fmt in FORMATS, fmt
        self.items: "OrderedDict[str, Tuple[int, np.ndarray]]" = OrderedDict()
        self.fmt = fmt
        self.x3 = x3 and fmt == "bf16"   # also emit the (hi, lo) tiles of the BF16X3 precision mode
        if fmt == "f16":
            self.add("__act_f16__", np.ones(1, np.int32), "i32")

    def bits(self, t: torch.Tensor) -> np.ndarray:
        return to_bf16_bits(t, self.fmt)

    def split(self, t: torch.Tensor):
        return split_bf16(t, self.fmt)

    def add(self, name: str, arr: np.ndarray, dtype: str):
        assert len(name) < 96 and arr.ndim <= 6
        self.items[name] = (_DT[dtype], np.ascontiguousarray(arr))

    def add_conv(self, name: str, w: torch.Tensor, b: torch.Tensor):
        self.add(name + ".w", tile_conv_weight(w, self.fmt), "bf16")
        if self.x3:
            self.add(name + ".w3", tile_conv_weight_x3(w), "bf16")
            self.add(name + ".wh", tile_conv_weight_f16x2(w), "bf16")      # PT_PRECISION_F16X2 (fp16 bits)
        self.add(name + ".b", b.numpy().astype(np.float32), "f32")

    def tobytes(self) -> bytes:
        return write_blob(self.items)


def write_blob(items) -> bytes:
    n = len(items)
    head = 8 + n * 144
    off = (head + 255) & ~255
    table = []
    payload = []
    for name, (dt, arr) in items.items():
        raw = arr.tobytes()
        dims = list(arr.shape) + [0] * (6 - arr.ndim)
        table.append(struct.pack("<96sII6IQQ", name.encode(), dt, arr.ndim, *dims, off, len(raw)))
        payload.append((off, raw))
        off = (off + len(raw) + 255) & ~255
    out = bytearray(off)
    out[0:4] = b"PTW1"
    out[4:8] = struct.pack("<I", n)
    for i, rec in enumerate(table):
        out[8 + i * 144: 8 + (i + 1) * 144] = rec
    for o, raw in payload:
        out[o:o + len(raw)] = raw
    return bytes(out)


def _phase_conv_weights(w: torch.Tensor) -> torch.Tensor:
    """conv3x3(w, up2(x)) as a convolution of x itself: [O, K, 3, 3] -> [4 O, K, 3, 3], output (dy * 2 + dx) * O + o = phase (dy, dx) of the
    2x2 block an input pixel is up-sampled to (pixel-shuffle store).  Row kernels {w[-1], w[0] + w[1], 0} for dy = 0 and
    {0, w[-1] + w[0], w[1]} for dy = 1, columns alike: 4 of the 9 taps are non-zero per phase.  Summed in float64."""
    wd = w.double()
    rows = {0: (wd[:, :, 0], wd[:, :, 1] + wd[:, :, 2], torch.zeros_like(wd[:, :, 0])),
            1: (torch.zeros_like(wd[:, :, 0]), wd[:, :, 0] + wd[:, :, 1], wd[:, :, 2])}          # [o, k, s] per low-res row offset
    ph = torch.zeros(4, w.shape[0], w.shape[1], 3, 3, dtype=torch.float64)
    for dy in range(2):
        for dx in range(2):
            for R in range(3):
                r_ = rows[dy][R]                                                                   # [o, k, 3 (s)]
                cols = (r_[:, :, 0], r_[:, :, 1] + r_[:, :, 2], torch.zeros_like(r_[:, :, 0])) if dx == 0 else \
                       (torch.zeros_like(r_[:, :, 0]), r_[:, :, 0] + r_[:, :, 1], r_[:, :, 2])
                for S in range(3):
                    ph[dy * 2 + dx, :, :, R, S] = cols[S]
    return ph.reshape(4 * w.shape[0], w.shape[1], 3, 3).float()


def pack_db_resnet18(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``DBModel`` state_dict (db_net/dbnet.py:715-728) -> blob for PT_MODEL_DB_RESNET18.
    ``x3``: also pack the (hi, lo) weight tiles that PT_PRECISION_BF16X3 uses (3x the conv weight bytes)."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    # stem: [64,3,7,7] -> [64][r=7][s=8][c=4], tap s=7 / channel c=3 are zero
    w, b = fold_conv_bn(sd, "backbone.conv1", "backbone.bn1")
    stem = torch.zeros(64, 7, 8, 4)
    stem[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    bl.add("stem.w", bl.bits(stem).reshape(64, 224), "bf16")
    if x3:
        sh, sl = bl.split(stem)
        bl.add("stem.w3", np.stack([bl.bits(sh).reshape(64, 224), bl.bits(sl).reshape(64, 224)]), "bf16")
    bl.add("stem.b", b.numpy(), "f32")
    for li in range(1, 5):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            q = f"layer{li}.{bi}"
            bl.add_conv(q + ".conv1", *fold_conv_bn(sd, p + ".conv1", p + ".bn1"))
            bl.add_conv(q + ".conv2", *fold_conv_bn(sd, p + ".conv2", p + ".bn2"))
            if (p + ".downsample.0.weight") in sd:
                bl.add_conv(q + ".down", *fold_conv_bn(sd, p + ".downsample.0", p + ".downsample.1"))
    for k in (2, 3, 4, 5):
        bl.add_conv(f"in{k}", *fold_conv_bn(sd, f"decoder.in{k}", None))
        oname = f"decoder.out{k}.0" if k > 2 else "decoder.out2"
        bl.add_conv(f"out{k}", *fold_conv_bn(sd, oname, None))
    # out2 = conv3x3(W_o, in2(c2) + up2(o3)) without the 256-channel lateral at 1/4 resolution (29.5 MB per page, the most expensive
    # bandwidth-type tensor of the decoder):  conv3x3(W_o . W_i, c2)  +  conv3x3(W_o, up2(o3)).  The first term is a 64 -> 64 3x3 conv with
    # the composed weights (in2 has no bias, so zero padding commutes with it); the second term reads a NEAREST-up-sampled map: output
    # phase (dy, dx) of a 2x2 block sees only a 2x2 neighbourhood of o3 -- a 3x3 conv at HALF resolution with 4 x 64 outputs (pixel
    # shuffle) whose kernel rows are {w[-1], w[0] + w[1], 0} for dy = 0 and {0, w[-1] + w[0], w[1]} for dy = 1 (columns alike): 4 of the 9
    # taps are non-zero per phase and the kernels skip the others (ConvDesc.tap_mask).  Composed in float64.
    wi, bi = fold_conv_bn(sd, "decoder.in2", None)
    wo, bo = fold_conv_bn(sd, "decoder.out2", None)
    if float(bi.abs().max()) == 0.0:
        w1 = torch.einsum("okrs,kc->ocrs", wo.double(), wi.double()[:, :, 0, 0]).float()
        bl.add_conv("out2f", w1, bo)
        bl.add_conv("out2p", _phase_conv_weights(wo), torch.zeros(4 * wo.shape[0]))
    wb, bb = fold_conv_bn(sd, "decoder.binarize.0", "decoder.binarize.1")
    bl.add_conv("bin0", wb, bb)
    # binarize.0 without the 256-channel concat at 1/4 resolution (dbnet.py:631-633: fuse = cat(up8(p5), up4(p4), up2(p3), p2), 29.5 MB per
    # page written and read back).  The conv is linear in its input channels, and up4 = up2 . up2, up8 = up2 . up4:
    #   conv3x3(W, fuse) = conv3x3(W[:, 192:], p2) + conv3x3(W[:, :192], up2(cat(up4(p5), up2(p4), p3)))
    # -- the second term is the phase convolution above on a 192-channel concat at 1/8 RESOLUTION (5.5 MB per page), the first a 64 -> 64
    # conv that takes it as its residual and applies the folded BN bias + ReLU: 9.9 instead of 17.0 GFLOP per page.
    if wb.shape[1] == 256 and wb.shape[0] == 64:
        bl.add_conv("bin0p", _phase_conv_weights(wb[:, :192]), torch.zeros(4 * wb.shape[0]))
        bl.add_conv("bin0c", wb[:, 192:].contiguous(), bb)
    # ConvTranspose2d(64,64,2,2)+BN -> 1x1 GEMM, N index = (dy*2+dx)*64 + co
    wt, bt = fold_conv_bn(sd, "decoder.binarize.3", "decoder.binarize.4", transposed=True)  # [ci, co, 2, 2]
    wn = wt.permute(2, 3, 1, 0).reshape(4 * 64, 64, 1, 1)
    bl.add_conv("bin3", wn, bt.repeat(4))
    w6, b6 = fold_conv_bn(sd, "decoder.binarize.6", None, transposed=True)  # [64, 1, 2, 2]
    bl.add("bin6.w", bl.bits(w6[:, 0].permute(1, 2, 0).reshape(4, 64)), "bf16")
    bl.add("bin6.wf32", w6[:, 0].permute(1, 2, 0).reshape(4, 64).contiguous().numpy().astype(np.float32), "f32")
    bl.add("bin6.b", b6.numpy().reshape(1), "f32")
    return bl.tobytes()


def pack_crnn(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``CRNN`` state_dict (crnn/modeling_crnn.py:40-90) -> blob for PT_MODEL_CRNN.

    conv0 (K = 9) stays fp32 for the VALU kernel; conv4's (2,1) kernel is flattened to a 1x1 GEMM over
    K = kh*512 + ci (the preceding pool writes H into channels); each LSTM's two directions share one input
    projection GEMM (N = 2048, both biases folded in); W_hh is stored [hi|lo][dir][1024][256]; the classifier is
    padded from 7644 to 7680 classes with zero weights and a -3e38 bias so that padding can never win the arg-max."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    w0, b0 = fold_conv_bn(sd, "conv0.0", "conv0.1")
    bl.add("conv0.wf32", w0.reshape(64, 9).numpy().astype(np.float32), "f32")
    bl.add("conv0.wbf", w0.reshape(64, 9).to(FORMATS[fmt]).float().numpy(), "f32")     # the 16-bit rounding of the single-pass modes, held in fp32
    bl.add("conv0.b", b0.numpy(), "f32")
    for name, cv, bn in (("conv1", "conv1.0", "conv1.1"), ("conv2a", "conv2.0", "conv2.1"), ("conv2b", "conv2.3", "conv2.4"),
                         ("conv3a", "conv3.0", "conv3.1"), ("conv3b", "conv3.3", "conv3.4")):
        bl.add_conv(name, *fold_conv_bn(sd, cv, bn))
    w4, b4 = fold_conv_bn(sd, "conv4.0", "conv4.1")            # [512, 512, 2, 1]
    bl.add_conv("conv4", w4[:, :, :, 0].permute(0, 2, 1).reshape(512, 1024, 1, 1), b4)
    for li, p in ((1, "rnn.0"), (2, "rnn.1")):
        r = p + ".rnn."
        wih = torch.cat([sd[r + "weight_ih_l0"], sd[r + "weight_ih_l0_reverse"]], 0)
        bias = torch.cat([sd[r + "bias_ih_l0"] + sd[r + "bias_hh_l0"],
                          sd[r + "bias_ih_l0_reverse"] + sd[r + "bias_hh_l0_reverse"]], 0)
        # output columns re-ordered to [dir][unit][gate] (gate fastest) so that the LSTM kernel reads the four gate
        # pre-activations of a (line, unit) with one 8-byte load
        perm = torch.arange(2048).view(2, 4, 256).permute(0, 2, 1).reshape(-1)
        bl.add_conv(f"lstm{li}.xproj", wih[perm].reshape(2048, -1, 1, 1).float(), bias[perm].float())
        whh = torch.stack([sd[r + "weight_hh_l0"], sd[r + "weight_hh_l0_reverse"]], 0).float()   # [2, 1024, 256]
        hi, lo = bl.split(whh)
        # MFMA-fragment order: the kernel's wave `wave` reads, for (half, kq, gate, h), one contiguous 1 KB record of
        # 64 lanes x 8 k-values -- eight full 128-byte lines per load instead of 32 quarter-used ones
        def frag(w):     # [2, 1024 = (gate 4, wave 4, h 2, lx 32), 256 = (half 4, kq 4, q 2, j 8)]
            return (w.reshape(2, 4, 4, 2, 32, 4, 4, 2, 8).permute(0, 2, 5, 6, 1, 3, 7, 4, 8).contiguous()
                    .reshape(2, 1024, 256))
        bl.add(f"lstm{li}.whh", np.stack([bl.bits(frag(hi)), bl.bits(frag(lo))]), "bf16")
        we = sd[p + ".embedding.weight"].float()
        bl.add_conv(f"lstm{li}.emb", we.reshape(we.shape[0], we.shape[1], 1, 1), sd[p + ".embedding.bias"].float())
    wc = sd["cls.weight"].float()
    ncls = wc.shape[0]
    npad = (ncls + 63) // 64 * 64
    wpad = torch.zeros(npad, wc.shape[1])
    wpad[:ncls] = wc
    bpad = torch.zeros(npad)
    bpad[ncls:] = -3.0e38
    bl.add_conv("cls", wpad.reshape(npad, wc.shape[1], 1, 1), bpad)
    return bl.tobytes()


# --------------------------------------------------------------------------------------------------------------------
# Lore: DLA-34 + DCN detector (lore/lore_dla_34.py:137-206 on center_net/modeling_centernet.py:274-409)
# --------------------------------------------------------------------------------------------------------------------
def _pad_conv(w: torch.Tensor, b: torch.Tensor, n_to: int, cin_to: int):
    """zero-pad [N, Cin, kh, kw] / [N] to n_to outputs and cin_to inputs (thin DLA levels, 27-channel offset convs)."""
    n, cin, kh, kw = w.shape
    wp = torch.zeros(n_to, cin_to, kh, kw)
    wp[:n, :cin] = w
    bp = torch.zeros(n_to)
    bp[:n] = b
    return wp, bp


def pack_lore_dla34(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``DLASeg`` state_dict -> blob for PT_MODEL_LORE_DLA34.

    * every Conv->BN pair folded; the 16-input-channel levels (level0, level1) are packed tap-major for the thin kernel;
    * Root 1x1 convs over a channel concat keep their weight (K in the order [x2, x1, *children] of the concat): the
      engine's 1x1 GEMM walks K over the child tensors, the concat is never materialised;
    * DCN: ``.om`` = the 27-channel offset/mask conv padded to 64 outputs (fp32 out), ``.dcn`` = the deformable conv
      as a 1x1 GEMM over the 9*C sampled columns (tap-major K), with ``actf`` BN folded in;
    * depthwise ConvTranspose2d up-samplers: fp32 ``[k*k][C]``."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    w, b = fold_conv_bn(sd, "base.base_layer.0", "base.base_layer.1")
    stem = torch.zeros(64, 7, 8, 4)
    stem[:16, :, :7, :3] = w.permute(0, 2, 3, 1)
    bp = torch.zeros(64)
    bp[:16] = b
    bl.add("base_layer.w", bl.bits(stem).reshape(64, 224), "bf16")
    if x3:
        sh, sl = bl.split(stem)
        bl.add("base_layer.w3", np.stack([bl.bits(sh).reshape(64, 224), bl.bits(sl).reshape(64, 224)]), "bf16")
    bl.add("base_layer.b", bp.numpy(), "f32")
    for name, key in (("level0", "base.level0"), ("level1", "base.level1")):
        # thin 16-input-channel levels: [9 taps][32 outputs (zero padded)][16 channels] for conv3x3_c16_kernel
        w, b = fold_conv_bn(sd, key + ".0", key + ".1")
        wt = torch.zeros(9, 32, 16)
        wt[:, :w.shape[0]] = w.permute(2, 3, 0, 1).reshape(9, w.shape[0], 16)
        bt = torch.zeros(32)
        bt[:w.shape[0]] = b
        bl.add(name + ".wt", bl.bits(wt), "bf16")
        if x3:
            hi, lo = bl.split(wt)
            bl.add(name + ".wt3", np.stack([bl.bits(hi), bl.bits(lo)]), "bf16")
        bl.add(name + ".bt", bt.numpy(), "f32")

    def block(p, q):
        bl.add_conv(q + ".conv1", *fold_conv_bn(sd, p + ".conv1", p + ".bn1"))
        bl.add_conv(q + ".conv2", *fold_conv_bn(sd, p + ".conv2", p + ".bn2"))

    def tree(p, q, levels, cin, cout, level_root, inherited=()):
        # Tree.forward (modeling_centernet.py:259-271): children = inherited (+ bottom if level_root); a leaf tree's
        # root sees [x2, x1, *children]; an inner tree appends its tree1 output and hands the list to tree2
        children = list(inherited) + ([cin] if level_root else [])
        if levels == 1:
            block(p + ".tree1", q + ".tree1")
            block(p + ".tree2", q + ".tree2")
            w, b = fold_conv_bn(sd, p + ".root.conv", p + ".root.bn")
            widths = [cout, cout] + children
            assert sum(widths) == w.shape[1], (p, widths, w.shape)
            bl.add_conv(f"{q}.root", w, b)
            if cin != cout:
                bl.add_conv(q + ".project", *fold_conv_bn(sd, p + ".project.0", p + ".project.1"))
        else:
            tree(p + ".tree1", q + ".tree1", levels - 1, cin, cout, False)
            tree(p + ".tree2", q + ".tree2", levels - 1, cout, cout, False, children + [cout])

    lv = [1, 1, 1, 2, 2, 1]
    ch = [16, 32, 64, 128, 256, 512]
    for l in range(2, 6):
        tree(f"base.level{l}", f"level{l}", lv[l], ch[l - 1], ch[l], l > 2)

    def dcn(p, q):
        w, b = fold_conv_bn(sd, p + ".conv.conv_offset_mask", None)
        bl.add_conv(q + ".om", *_pad_conv(w, b, 64, w.shape[1]))
        w, b = fold_conv_bn(sd, p + ".conv", p + ".actf.0")
        o, c = w.shape[0], w.shape[1]
        w1 = w.permute(0, 2, 3, 1).reshape(o, 9 * c, 1, 1).contiguous()       # K = tap * C + c
        bl.add_conv(q + ".dcn", w1, b)

    def ida(p, n):
        for j in range(1, n + 1):
            dcn(f"{p}.proj_{j}", f"{p}.proj_{j}")
            dcn(f"{p}.node_{j}", f"{p}.node_{j}")
            wu = sd[f"{p}.up_{j}.weight"]                                       # [C, 1, k, k]
            k = wu.shape[2]
            bl.add(f"{p}.up_{j}.wf32", wu[:, 0].permute(1, 2, 0).reshape(k * k, -1).contiguous().numpy(), "f32")

    ida("dla_up.ida_0", 1)
    ida("dla_up.ida_1", 2)
    ida("dla_up.ida_2", 3)
    ida("ida_up", 2)
    for h, k in (("hm", 2), ("st", 8), ("wh", 8), ("ax", 256), ("cr", 256), ("reg", 2)):
        bl.add_conv(f"{h}.0", *fold_conv_bn(sd, f"{h}.0", None))
        w, b = fold_conv_bn(sd, f"{h}.2", None)
        nt = (k + 63) // 64 * 64
        bl.add_conv(f"{h}.2", *_pad_conv(w, b, nt, 256))
    return bl.tobytes()


def pack_lore_processor(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``LoreProcessModel`` state_dict (lore/lore_processor.py:399-437) -> blob for PT_MODEL_LORE_PROCESSOR.
    Every nn.Linear becomes a 1x1 GEMM tile set; q/k/v projections are fused into one 256 -> 768 GEMM ([q; k; v]);
    ``logi_encoder.0`` (4 inputs) and the 4-output decoders are zero-padded to 32 inputs / 64 outputs; Norm
    parameters and the two position tables stay fp32.  ``meta`` = [tsfm layers, stacking layers]."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3

    def lin(name, key, n_to=None, cin_to=None):
        w = sd[key + ".weight"].float()
        b = sd[key + ".bias"].float() if (key + ".bias") in sd else torch.zeros(w.shape[0])
        n_to = n_to or w.shape[0]
        cin_to = cin_to or w.shape[1]
        bl.add_conv(name, *_pad_conv(w.reshape(w.shape[0], w.shape[1], 1, 1), b, n_to, cin_to))

    def transformer(p, q):
        lin(q + ".linear", p + ".linear")
        n = 0
        while f"{p}.encoder.layers.{n}.norm_1.alpha" in sd:
            lp, lq = f"{p}.encoder.layers.{n}", f"{q}.l{n}"
            for nm in ("norm_1", "norm_2"):
                bl.add(f"{lq}.{nm}.alpha", sd[f"{lp}.{nm}.alpha"].float().numpy(), "f32")
                bl.add(f"{lq}.{nm}.bias", sd[f"{lp}.{nm}.bias"].float().numpy(), "f32")
            wq = torch.cat([sd[f"{lp}.attn.{k}_linear.weight"] for k in ("q", "k", "v")], 0).float()
            bq = torch.cat([sd[f"{lp}.attn.{k}_linear.bias"] for k in ("q", "k", "v")], 0).float()
            bl.add_conv(lq + ".qkv", wq.reshape(768, 256, 1, 1), bq)
            lin(lq + ".out", lp + ".attn.out")
            lin(lq + ".ff1", lp + ".ff.linear_1")
            lin(lq + ".ff2", lp + ".ff.linear_2")
            n += 1
        lin(q + ".dec0", p + ".decoder.linear.0")
        lin(q + ".dec2", p + ".decoder.linear.2", n_to=64)
        return n

    n_axis = transformer("tsfm_axis", "axis")
    lin("stk.le0", "stacker.logi_encoder.0", cin_to=32)
    lin("stk.le2", "stacker.logi_encoder.2")
    n_stack = transformer("stacker.tsfm", "stk")
    bl.add("x_pe", sd["x_position_embeddings.weight"].float().numpy(), "f32")
    bl.add("y_pe", sd["y_position_embeddings.weight"].float().numpy(), "f32")
    bl.add("meta", np.array([n_axis, n_stack], dtype=np.int32), "i32")
    return bl.tobytes()


# --------------------------------------------------------------------------------------------------------------------
# PicoDet layout detector (picodet/lcnet.py, csp_pan.py, pico_head.py)
# --------------------------------------------------------------------------------------------------------------------
def _fold_named(sd, conv_key: str, bn_key: str):
    """like fold_conv_bn, for modules whose conv / norm children have arbitrary names"""
    w = sd[conv_key + ".weight"].double()
    b = sd[conv_key + ".bias"].double() if (conv_key + ".bias") in sd else torch.zeros(w.shape[0], dtype=torch.float64)
    scale = sd[bn_key + ".weight"].double() / torch.sqrt(sd[bn_key + ".running_var"].double() + BN_EPS)
    shift = sd[bn_key + ".bias"].double() - sd[bn_key + ".running_mean"].double() * scale
    return (w * scale.view(-1, 1, 1, 1)).float(), (b * scale + shift).float()


def pack_picodet(sd: Dict[str, torch.Tensor], num_classes: int = 5, x3: bool = True, fmt: str = "bf16") -> bytes:
    """PicoDet state_dict (keys backbone. / neck. / head., pdf_table_amd.synth_weights.picodet_state_dict) -> blob for
    PT_MODEL_PICODET.  BN folded everywhere; depthwise kernels fp32 ``[k*k][C]`` + bias; 1x1 convs tiled for the MFMA
    kernel (16-channel tensors stored 32 wide; convs over a channel concat split per operand: ``.a`` / ``.b``);
    SE fully-connected layers fp32; head convs padded to 64 outputs (``meta`` = [num_classes, reg_max])."""
    from .synth_weights import LCNET_CONFIG, PICODET_STANDIN
    bl = _Blob(x3, fmt)
    x3 = bl.x3

    def dw(name, conv_key, bn_key):
        w, b = _fold_named(sd, conv_key, bn_key)                    # [C, 1, k, k]
        c, _, k, _ = w.shape
        cp = max(c, 32)
        wp = torch.zeros(k * k, cp)
        wp[:, :c] = w[:, 0].permute(1, 2, 0).reshape(k * k, c)
        bp = torch.zeros(cp)
        bp[:c] = b
        bl.add(name + ".wf32", wp.numpy(), "f32")
        bl.add(name + ".b", bp.numpy(), "f32")

    def pw(name, conv_key, bn_key, split_at=None):
        w, b = _fold_named(sd, conv_key, bn_key)
        n, cin = w.shape[0], w.shape[1]
        nt = (n + 63) // 64 * 64
        if split_at is None:
            bl.add_conv(name, *_pad_conv(w, b, nt, max(cin, 32)))
        else:
            bl.add_conv(name + ".a", *_pad_conv(w[:, :split_at].contiguous(), b, nt, split_at))
            bl.add_conv(name + ".b", *_pad_conv(w[:, split_at:].contiguous(), torch.zeros_like(b), nt, cin - split_at))

    # stem: conv 3x3 s2 3 -> 16 (+BN): fp32 [16][3][3][4] (channel 3 zero) for the direct kernel
    w, b = _fold_named(sd, "backbone.conv1.conv", "backbone.conv1.bn")
    st = torch.zeros(16, 3, 3, 4)
    st[:, :, :, :3] = w.permute(0, 2, 3, 1)
    bl.add("stem.wf32", st.numpy(), "f32")
    bl.add("stem.b", b.numpy(), "f32")
    for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
        for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
            p, q = f"backbone.{blk}.{i}", f"{blk}.{i}"
            dw(q + ".dw", p + ".dw_conv.conv", p + ".dw_conv.bn")
            if se:
                bl.add(q + ".se.w1", sd[p + ".se.conv1.weight"][:, :, 0, 0].float().numpy(), "f32")     # [C/4, C]
                bl.add(q + ".se.b1", sd[p + ".se.conv1.bias"].float().numpy(), "f32")
                bl.add(q + ".se.w2", sd[p + ".se.conv2.weight"][:, :, 0, 0].float().numpy(), "f32")     # [C, C/4]
                bl.add(q + ".se.b2", sd[p + ".se.conv2.bias"].float().numpy(), "f32")
            pw(q + ".pw", p + ".pw_conv.conv", p + ".pw_conv.bn")
    nc = PICODET_STANDIN["neck_channels"]

    def dp(q, p):
        dw(q + ".dw", p + ".dwconv", p + ".bn1")
        pw(q + ".pw", p + ".pwconv", p + ".bn2")

    def csp(q, p):
        pw(q + ".main", p + ".main_conv.conv", p + ".main_conv.bn", split_at=nc)
        pw(q + ".short", p + ".short_conv.conv", p + ".short_conv.bn", split_at=nc)
        pw(q + ".conv1", p + ".blocks.0.conv1.conv", p + ".blocks.0.conv1.bn")
        dp(q + ".dp", p + ".blocks.0.conv2")
        pw(q + ".final", p + ".final_conv.conv", p + ".final_conv.bn", split_at=nc // 2)

    for i in range(3):
        pw(f"neck.t{i}", f"neck.conv_t.convs.{i}.conv", f"neck.conv_t.convs.{i}.bn")
    dp("neck.top1", "neck.first_top_conv")
    dp("neck.top2", "neck.second_top_conv")
    for i in range(2):
        csp(f"neck.td{i}", f"neck.top_down_blocks.{i}")
        dp(f"neck.down{i}", f"neck.downsamples.{i}")
        csp(f"neck.bu{i}", f"neck.bottom_up_blocks.{i}")
    for s in range(4):
        for i in range(PICODET_STANDIN["num_convs"]):
            dw(f"head.{s}.{i}.dw", f"head.conv_feat.cls_conv_dw{s}_{i}.conv", f"head.conv_feat.cls_conv_dw{s}_{i}.norm")
            pw(f"head.{s}.{i}.pw", f"head.conv_feat.cls_conv_pw{s}_{i}.conv", f"head.conv_feat.cls_conv_pw{s}_{i}.norm")
        w = sd[f"head.head_cls{s}.weight"].float()
        bl.add_conv(f"head.{s}.out", *_pad_conv(w, sd[f"head.head_cls{s}.bias"].float(), 64, nc))
    bl.add("meta", np.array([num_classes, PICODET_STANDIN["reg_max"]], dtype=np.int32), "i32")
    return bl.tobytes()


def deconv4x4s2_as_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k=4, s=2, p=1) weight [Cin, Cout, 4, 4] -> the equivalent 3x3 convolution with 4*Cout outputs
    (quadrant-major: q = 2*py + px feeds output pixel (2y+py, 2x+px)) for the pixel-shuffle epilogue.  Output row
    2y+py takes input rows y+dy with kernel row ky: py=0: (dy=-1, ky=3), (0, 1); py=1: (0, 2), (+1, 0)."""
    cin, cout = w.shape[0], w.shape[1]
    taps = {0: ((-1, 3), (0, 1)), 1: ((0, 2), (1, 0))}
    out = torch.zeros(4 * cout, cin, 3, 3, dtype=w.dtype)
    for py in (0, 1):
        for px in (0, 1):
            q = 2 * py + px
            for dy, ky in taps[py]:
                for dx, kx in taps[px]:
                    out[q * cout:(q + 1) * cout, :, dy + 1, dx + 1] = w[:, :, ky, kx].t()
    return out


def pack_lore_wireless(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``LoreDetectModel`` state_dict (lore/lore_detector.py:155-286) -> blob for PT_MODEL_LORE_RESNET18.  Conv+BN folded
    (block convs carry a bias); the 4x4/s2 transposed convolutions become 3x3 convs with a pixel-shuffle epilogue."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    w, b = fold_conv_bn(sd, "conv1", "bn1")
    stem = torch.zeros(64, 7, 8, 4)
    stem[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    bl.add("stem.w", bl.bits(stem).reshape(64, 224), "bf16")
    if x3:
        sh, sl = bl.split(stem)
        bl.add("stem.w3", np.stack([bl.bits(sh).reshape(64, 224), bl.bits(sl).reshape(64, 224)]), "bf16")
    bl.add("stem.b", b.numpy(), "f32")
    for li in range(1, 5):
        for bi in range(2):
            p = f"layer{li}.{bi}"
            bl.add_conv(p + ".conv1", *fold_conv_bn(sd, p + ".conv1", p + ".bn1"))
            bl.add_conv(p + ".conv2", *fold_conv_bn(sd, p + ".conv2", p + ".bn2"))
            if (p + ".downsample.0.weight") in sd:
                bl.add_conv(p + ".down", *fold_conv_bn(sd, p + ".downsample.0", p + ".downsample.1"))
    for name in ("adaption3", "adaption2", "adaption1", "adaption0", "adaptionU1"):
        bl.add_conv(name, *fold_conv_bn(sd, name, None))
    for i in range(1, 5):
        wt = sd[f"deconv_layers{i}.0.weight"].double()
        bn = f"deconv_layers{i}.1"
        scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
        shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale
        w3 = deconv4x4s2_as_conv3x3(wt * scale.view(1, -1, 1, 1))
        bl.add_conv(f"deconv{i}", w3.float(), shift.repeat(4).float())
    for h, k in (("hm", 2), ("st", 8), ("wh", 8), ("ax", 256), ("cr", 256), ("reg", 2)):
        n3 = 1 if h == "reg" else 4
        for j in range(n3):
            bl.add_conv(f"{h}.c{j}", *fold_conv_bn(sd, f"{h}.{2 * j}", None))
        w, b = fold_conv_bn(sd, f"{h}.{2 * n3}", None)
        bl.add_conv(f"{h}.out", *_pad_conv(w, b, (k + 63) // 64 * 64, 64))
    return bl.tobytes()


def pack_db_nas(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``DBNasModel`` state_dict (db_net/dbnet.py:693-712) -> blob for PT_MODEL_DB_NAS (csrc/dbnas_model.hip).

    * every BatchNorm folded (float64); channel counts padded to multiples of 64 with zero weights;
    * a ``rep`` block's parallel depthwise branches (1x1 / 3x3 / 5x5, each + BN, summed: layers.py:732-745) become ONE
      centred 5x5 depthwise kernel + bias -- convolution is linear, so the sum of the folded branches is exact;
    * PReLU slopes (one scalar per activation) as 1-element fp32 tensors read on the device;
    * ``dec.tail``: the two DwPwConvTranspose blocks + BN of LightSegDetector.binarize as the 449-float table
      dbnas_tail_kernel documents."""
    from .dbnas_arch import dbnas_blocks
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    p64 = lambda c: (c + 63) // 64 * 64

    def bn_affine(bn):
        scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + BN_EPS)
        return scale, sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale

    w, b = fold_conv_bn(sd, "backbone.first_conv.0", "backbone.first_conv.1")
    st = torch.zeros(32, 3, 3, 4)
    st[:, :, :, :3] = w.permute(0, 2, 3, 1)
    bl.add("stem.wf32", st.numpy(), "f32")
    bl.add("stem.b", b.numpy(), "f32")
    for bi, blk in enumerate(dbnas_blocks()):
        p, q = f"backbone.blocks.{bi}.mobile_inverted_conv", f"b{bi}"
        if blk["kind"] == "se":
            c, cp, hid = blk["cin"], p64(blk["cin"]), blk["squeeze"]
            w1 = torch.zeros(hid, cp)
            w1[:, :c] = sd[p + ".fc1.weight"][:, :, 0, 0].float()
            w2 = torch.zeros(cp, hid)
            w2[:c] = sd[p + ".fc2.weight"][:, :, 0, 0].float()
            b2 = torch.zeros(cp)
            b2[:c] = sd[p + ".fc2.bias"].float()
            bl.add(q + ".se.w1", w1.numpy(), "f32")
            bl.add(q + ".se.b1", sd[p + ".fc1.bias"].float().numpy(), "f32")
            bl.add(q + ".se.w2", w2.numpy(), "f32")
            bl.add(q + ".se.b2", b2.numpy(), "f32")
            continue
        mid, cin, cout = blk["mid"], blk["cin"], blk["cout"]
        w, b = fold_conv_bn(sd, p + ".inverted_bottleneck.conv", p + ".inverted_bottleneck.bn")
        bl.add_conv(q + ".exp", *_pad_conv(w, b, mid, p64(cin)))
        bl.add(q + ".exp.slope", sd[p + ".inverted_bottleneck.act.weight"].float().numpy().reshape(1), "f32")
        kmax = max(blk["sizes"])
        wk = torch.zeros(mid, kmax, kmax, dtype=torch.float64)
        bk = torch.zeros(mid, dtype=torch.float64)
        if blk["kind"] == "rep":
            branches = [(f"{p}.rep_conv.{ri}.conv", f"{p}.rep_conv.{ri}.bn", k) for ri, k in enumerate(blk["sizes"])]
            slope = sd[p + ".act.weight"]
        else:
            branches = [(p + ".depth_conv.conv", p + ".depth_conv.bn", blk["sizes"][0])]
            slope = sd[p + ".depth_conv.act.weight"]
        for conv, bn, k in branches:
            sc, sh = bn_affine(bn)
            o = (kmax - k) // 2
            wk[:, o:o + k, o:o + k] += sd[conv + ".weight"][:, 0].double() * sc.view(-1, 1, 1)
            bk += sh
        bl.add(q + ".dw.wf32", wk.permute(1, 2, 0).reshape(kmax * kmax, mid).float().numpy(), "f32")
        bl.add(q + ".dw.b", bk.float().numpy(), "f32")
        bl.add(q + ".dw.slope", slope.float().numpy().reshape(1), "f32")
        w, b = fold_conv_bn(sd, p + ".point_conv.conv", p + ".point_conv.bn")
        bl.add_conv(q + ".proj", *_pad_conv(w, b, p64(cout), mid))
    for name in ("in5", "in4", "in3", "in2"):
        w = sd[f"decoder.{name}.weight"].float()
        bl.add_conv("dec." + name, *_pad_conv(w, torch.zeros(w.shape[0]), 64, p64(w.shape[1])))
    sc, sh = bn_affine("decoder.binarize.0.bn1")
    wd = sd["decoder.binarize.0.depthwise.weight"][:, 0].double() * sc.view(-1, 1, 1)          # [64, 5, 5]
    k = wd.shape[-1]
    bl.add("dec.dw.wf32", wd.permute(1, 2, 0).reshape(k * k, -1).float().numpy(), "f32")
    bl.add("dec.dw.b", sh.float().numpy(), "f32")
    w, b = fold_conv_bn(sd, "decoder.binarize.0.pointwise", "decoder.binarize.1")
    bl.add_conv("dec.pw", *_pad_conv(w, b, 64, 64))
    tail = np.zeros(449, dtype=np.float64)

    def dwt(prefix, w_off, b_off):      # depthwise ConvTranspose2d(k=2, s=2) + BN: out[2y+dy, 2x+dx, c] = in[y, x, c] * W[2dy+dx][c] + B[c]
        sc, sh = bn_affine(prefix + ".bn1")
        wt = sd[prefix + ".depthwise.weight"][:, 0].double() * sc.view(-1, 1, 1)                # [16, 2, 2]
        tail[w_off:w_off + 64] = wt.permute(1, 2, 0).reshape(-1).numpy()
        tail[b_off:b_off + 16] = (sd[prefix + ".depthwise.bias"].double() * sc + sh).numpy()

    dwt("decoder.binarize.3", 0, 64)
    w, b = fold_conv_bn(sd, "decoder.binarize.3.pointwise", "decoder.binarize.4")
    tail[80:336] = w[:, :, 0, 0].double().reshape(-1).numpy()
    tail[336:352] = b.double().numpy()
    dwt("decoder.binarize.6", 352, 416)
    tail[432:448] = sd["decoder.binarize.6.pointwise.weight"][0, :, 0, 0].double().numpy()
    tail[448] = float(sd["decoder.binarize.6.pointwise.bias"][0])
    bl.add("dec.tail", tail.astype(np.float32), "f32")
    return bl.tobytes()


def pack_pplcnet(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``PPLCNet`` state_dict (model/cls/cls_pp_lcnet.py:164-260) -> blob for PT_MODEL_PPLCNET (+ slot).  The backbone
    tensors have the names and layouts of pack_picodet's LCNet part (the same launch code runs both); ``last_conv`` and
    ``fc`` are 1x1 GEMMs over the pooled vectors (fc padded to 64 outputs, 16 stored); ``fc.nclass`` carries the class
    count in its length."""
    from .synth_weights import LCNET_CONFIG
    bl = _Blob(x3, fmt)
    x3 = bl.x3

    def dw(name, conv_key, bn_key):
        w, b = _fold_named(sd, conv_key, bn_key)                    # [C, 1, k, k]
        c, _, k, _ = w.shape
        cp = max(c, 32)
        wp = torch.zeros(k * k, cp)
        wp[:, :c] = w[:, 0].permute(1, 2, 0).reshape(k * k, c)
        bp = torch.zeros(cp)
        bp[:c] = b
        bl.add(name + ".wf32", wp.numpy(), "f32")
        bl.add(name + ".b", bp.numpy(), "f32")

    w, b = _fold_named(sd, "conv1.conv", "conv1.bn")
    st = torch.zeros(16, 3, 3, 4)
    st[:, :, :, :3] = w.permute(0, 2, 3, 1)
    bl.add("stem.wf32", st.numpy(), "f32")
    bl.add("stem.b", b.numpy(), "f32")
    for blk in ("blocks2", "blocks3", "blocks4", "blocks5", "blocks6"):
        for i, (k, cin, cout, s, se) in enumerate(LCNET_CONFIG[blk]):
            p = f"{blk}.{i}"
            dw(p + ".dw", p + ".dw_conv.conv", p + ".dw_conv.bn")
            if se:
                bl.add(p + ".se.w1", sd[p + ".se.conv1.weight"][:, :, 0, 0].float().numpy(), "f32")
                bl.add(p + ".se.b1", sd[p + ".se.conv1.bias"].float().numpy(), "f32")
                bl.add(p + ".se.w2", sd[p + ".se.conv2.weight"][:, :, 0, 0].float().numpy(), "f32")
                bl.add(p + ".se.b2", sd[p + ".se.conv2.bias"].float().numpy(), "f32")
            w, b = _fold_named(sd, p + ".pw_conv.conv", p + ".pw_conv.bn")
            bl.add_conv(p + ".pw", *_pad_conv(w, b, (w.shape[0] + 63) // 64 * 64, max(w.shape[1], 32)))
    wl = sd["last_conv.weight"].float()
    bl.add_conv("last_conv", wl, torch.zeros(wl.shape[0]))
    wf = sd["fc.weight"].float()
    ncls = wf.shape[0]
    assert ncls <= 16, "the classifier head stores 16 logits per image"
    bl.add_conv("fc", *_pad_conv(wf.reshape(ncls, -1, 1, 1), sd["fc.bias"].float(), 64, wf.shape[1]))
    bl.add("fc.nclass", np.zeros(ncls, dtype=np.float32), "f32")
    return bl.tobytes()


# --------------------------------------------------------------------------------------------------------------------
# ConvNextViT recogniser (convnext_vit/modeling_convnext_vit.py:20-45; kernels: csrc/cvit_model.hip)
# --------------------------------------------------------------------------------------------------------------------
_CVIT_V5_TO_V4 = [
    (r"vit\.layers\.(\d+)\.attention\.q_proj\.", r"vit.encoder.layer.\1.attention.attention.query."),
    (r"vit\.layers\.(\d+)\.attention\.k_proj\.", r"vit.encoder.layer.\1.attention.attention.key."),
    (r"vit\.layers\.(\d+)\.attention\.v_proj\.", r"vit.encoder.layer.\1.attention.attention.value."),
    (r"vit\.layers\.(\d+)\.attention\.o_proj\.", r"vit.encoder.layer.\1.attention.output.dense."),
    (r"vit\.layers\.(\d+)\.mlp\.fc1\.", r"vit.encoder.layer.\1.intermediate.dense."),
    (r"vit\.layers\.(\d+)\.mlp\.fc2\.", r"vit.encoder.layer.\1.output.dense."),
    (r"vit\.layers\.(\d+)\.layernorm_", r"vit.encoder.layer.\1.layernorm_"),
]


def pack_convnext_vit(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``ConvNextViT`` state_dict -> blob for PT_MODEL_CONVNEXT_VIT.  Key names of the reference's checkpoint era
    (transformers 4.x, ``vitstr.vit.encoder.layer.N.attention.attention.query`` ...) or of transformers 5.x
    (``vitstr.vit.layers.N.attention.q_proj`` ...), with or without the ``recognizer.`` / ``module.`` prefixes
    (modeling_ocr_recognition.py:108-111).

    Patch embedding and depthwise 7x7 weights stay fp32 (VALU kernels; taps-major ``[49][C]``); every Linear is a 1x1 GEMM
    tile set; ``layer_scale_parameter`` is folded into ``pwconv2`` (scale * (W h + b) = (scale W) h + scale b, in fp64); the
    (2,1)-kernel down-sampling convs become GEMMs over K = row * Cin + c; q/k/v are one 192 -> 576 GEMM with the attention's
    1/sqrt(64) folded into q (a power of two: exact); position embeddings without the class-token slot; the classifier is
    padded from 7644 to 7680 classes with a -3e38 bias."""
    import re
    csd = {}
    for k, v in sd.items():
        k = k.replace("recognizer.", "").replace("module.", "")
        for pat, rep in _CVIT_V5_TO_V4:
            k = re.sub(pat, rep, k)
        csd[k] = v.detach().float()
    sd = csd
    bl = _Blob(x3, fmt)
    x3 = bl.x3

    def f32(name, t):
        bl.add(name, t.contiguous().numpy().astype(np.float32), "f32")

    def ln(name, key):
        f32(name + ".g", sd[key + ".weight"])
        f32(name + ".b", sd[key + ".bias"])

    def lin(name, w, b, n_to=None):
        n_to = n_to or w.shape[0]
        bl.add_conv(name, *_pad_conv(w.reshape(w.shape[0], w.shape[1], 1, 1), b, n_to, w.shape[1]))

    # the fused MLP kernel (cvit_mlp_kernel, bf16 mode, C <= 256): W1 rows as they are ([4C][C]: a 32-unit chunk is
    # contiguous); W2 [C][4C] with the K order of every 32-unit chunk permuted to the MFMA accumulator layout the hidden
    # values are produced in: slot (s2, half, j) of chunk hc holds hidden unit hc * 32 + (j & 3) + 8 * (2 * s2 + (j >> 2)) +
    # 4 * half; stored [chunk][C][32]
    s2_, half_, j_ = np.meshgrid(np.arange(2), np.arange(2), np.arange(8), indexing="ij")
    perm = torch.from_numpy(((j_ & 3) + 8 * (2 * s2_ + (j_ >> 2)) + 4 * half_).reshape(-1))

    def mlp(name, w1, w2):
        c = w1.shape[1]
        if c > 256:
            return
        bl.add(name + ".mlp.w1", bl.bits(w1), "bf16")
        w2c = w2.reshape(c, 4 * c // 32, 32)[:, :, perm].permute(1, 0, 2).contiguous()          # [chunk][C][32]
        bl.add(name + ".mlp.w2p", bl.bits(w2c), "bf16")

    p = "cnn_model."
    f32("embed.w", sd[p + "embeddings.patch_embeddings.weight"].reshape(96, 16))
    f32("embed.b", sd[p + "embeddings.patch_embeddings.bias"])
    ln("embed.ln", p + "embeddings.layernorm")
    dims, depths = (96, 192, 256, 512), (3, 3, 8, 3)
    for i, (d, dep) in enumerate(zip(dims, depths)):
        q = f"{p}encoder.stages.{i}."
        if i > 0:
            ln(f"s{i}.down.ln", q + "downsampling_layer.0")
            w = sd[q + "downsampling_layer.1.weight"]                       # [N, Cin, 2, 1]
            lin(f"s{i}.down", w[:, :, :, 0].permute(0, 2, 1).reshape(d, 2 * dims[i - 1]), sd[q + "downsampling_layer.1.bias"])
        for j in range(dep):
            lq, o = f"{q}layers.{j}.", f"s{i}.l{j}"
            f32(o + ".dw.w", sd[lq + "dwconv.weight"].reshape(d, 49).t())   # [49][C]
            f32(o + ".dw.b", sd[lq + "dwconv.bias"])
            ln(o + ".ln", lq + "layernorm")
            lin(o + ".pw1", sd[lq + "pwconv1.weight"], sd[lq + "pwconv1.bias"])
            g = sd[lq + "layer_scale_parameter"].double() if (lq + "layer_scale_parameter") in sd else torch.ones(d, dtype=torch.float64)
            w2s = (sd[lq + "pwconv2.weight"].double() * g[:, None]).float()
            lin(o + ".pw2", w2s, (sd[lq + "pwconv2.bias"].double() * g).float(), n_to=(d + 63) // 64 * 64)
            mlp(o, sd[lq + "pwconv1.weight"], w2s)
    p = "vitstr.vit."
    lin("vit.embed", sd[p + "embeddings.patch_embeddings.projection.weight"].reshape(192, 512), sd[p + "embeddings.patch_embeddings.projection.bias"])
    f32("vit.pos", sd[p + "embeddings.position_embeddings"][0, 1:, :])
    n = 0
    while f"{p}encoder.layer.{n}.layernorm_before.weight" in sd:
        q, o = f"{p}encoder.layer.{n}.", f"vit.l{n}"
        ln(o + ".ln1", q + "layernorm_before")
        ln(o + ".ln2", q + "layernorm_after")
        a = q + "attention.attention."
        wq = torch.cat([sd[a + "query.weight"] * 0.125, sd[a + "key.weight"], sd[a + "value.weight"]], 0)
        bq = torch.cat([sd[a + "query.bias"] * 0.125, sd[a + "key.bias"], sd[a + "value.bias"]], 0)
        lin(o + ".qkv", wq, bq)
        lin(o + ".out", sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"])
        lin(o + ".fc1", sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"])
        lin(o + ".fc2", sd[q + "output.dense.weight"], sd[q + "output.dense.bias"])
        mlp(o, sd[q + "intermediate.dense.weight"], sd[q + "output.dense.weight"])
        n += 1
    assert n == 12, f"ConvNextViT: {n} ViT layers in the checkpoint, the launch graph runs 12 (modeling_convnext_vit.py:28-35)"
    ln("vit.ln", p + "layernorm")
    wc, bc = sd["vitstr.classifier.weight"], sd["vitstr.classifier.bias"]
    ncls = wc.shape[0]
    npad = 7680
    assert ncls <= npad
    wpad = torch.zeros(npad, 192)
    wpad[:ncls] = wc
    bpad = torch.full((npad,), -3.0e38)
    bpad[:ncls] = bc
    bl.add_conv("cls", wpad.reshape(npad, 192, 1, 1), bpad)
    bl.add("meta", np.array([ncls], dtype=np.int32), "i32")
    return bl.tobytes()


# --------------------------------------------------------------------------------------------------------------------
# MtlTabNet backbone (table/mtl_tabnet/table_resnet_extra.py:205-318; kernels: csrc/mtl_model.hip)
# --------------------------------------------------------------------------------------------------------------------
def pack_mtl_backbone(sd: Dict[str, torch.Tensor], x3: bool = True, fmt: str = "bf16") -> bytes:
    """``TableResNetExtra`` state_dict (layers [1, 2, 5, 3], context blocks in the first block of stages 2-4) -> blob for
    PT_MODEL_MTL_BACKBONE.  Every conv with its BatchNorm folded in (conv1's 3 input channels zero-padded to 32); the context
    blocks' small tensors stay fp32: mask conv ``wm [C]`` / ``bm``, ``w0 [hid][C]`` / ``b0``, LayerNorm ``lg`` / ``lb [hid]``,
    ``w3 [C][hid]`` / ``b3``."""
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    w, b = fold_conv_bn(sd, "conv1", "bn1")
    bl.add_conv("conv1", *_pad_conv(w, b, 64, 32))
    for i in range(2, 7):
        bl.add_conv(f"conv{i}", *fold_conv_bn(sd, f"conv{i}", f"bn{i}"))
    for li, nblk in enumerate((1, 2, 5, 3), start=1):
        for j in range(nblk):
            p = f"layer{li}.{j}"
            bl.add_conv(p + ".conv1", *fold_conv_bn(sd, p + ".conv1", p + ".bn1"))
            bl.add_conv(p + ".conv2", *fold_conv_bn(sd, p + ".conv2", p + ".bn2"))
            if (p + ".downsample.0.weight") in sd:
                bl.add_conv(p + ".down", *fold_conv_bn(sd, p + ".downsample.0", p + ".downsample.1"))
            q = p + ".context_block"
            if (q + ".conv_mask.weight") in sd:
                f = lambda t: t.detach().float().contiguous().numpy().astype(np.float32)
                c = sd[q + ".conv_mask.weight"].shape[1]
                hid = sd[q + ".channel_add_conv.0.weight"].shape[0]
                bl.add(p + ".gc.wm", f(sd[q + ".conv_mask.weight"].reshape(c)), "f32")
                bl.add(p + ".gc.bm", f(sd[q + ".conv_mask.bias"].reshape(1)), "f32")
                bl.add(p + ".gc.w0", f(sd[q + ".channel_add_conv.0.weight"].reshape(hid, c)), "f32")
                bl.add(p + ".gc.b0", f(sd[q + ".channel_add_conv.0.bias"]), "f32")
                bl.add(p + ".gc.lg", f(sd[q + ".channel_add_conv.1.weight"].reshape(hid)), "f32")
                bl.add(p + ".gc.lb", f(sd[q + ".channel_add_conv.1.bias"].reshape(hid)), "f32")
                bl.add(p + ".gc.w3", f(sd[q + ".channel_add_conv.3.weight"].reshape(c, hid)), "f32")
                bl.add(p + ".gc.b3", f(sd[q + ".channel_add_conv.3.bias"]), "f32")
    return bl.tobytes()


MTL_LAYERS = ("l0", "l1", "cls", "bbox", "cell")      # slots of the five DecoderLayers in the blob and in the cross K / V tensor
MTL_D = 512
MTL_PE_ROWS = 4096


def mtl_positional_table(rows: int = MTL_PE_ROWS, d_model: int = MTL_D) -> torch.Tensor:
    """PositionalEncoding.pe (table/mtl_tabnet/master_decoder.py:166-180) -- computed with torch exactly as the reference does."""
    import math
    pe = torch.zeros(rows, d_model)
    position = torch.arange(0, rows).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -math.log(10000.0) / d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def pack_mtl_decoder(sd: Dict[str, torch.Tensor], cfg: Dict, x3: bool = True, fmt: str = "bf16") -> bytes:
    """``MtlTabNetDecoder`` state_dict (table/mtl_tabnet/master_decoder.py:194-262; N = 3: two shared DecoderLayers, then the
    structure, box and cell-content layers) -> blob for PT_MODEL_MTL_DECODER.  ``cfg``: sos / eos / pad / max_len, sos_cell /
    eos_cell / pad_cell / max_len_cell, idx_tag_cell (master_convertor.py:541-549).

    Every nn.Linear is a 1x1 GEMM tile set.  Per layer: ``qkv`` = [q; k / 8; v] of the self-attention (self_attention() scales
    the KEYS by 1 / sqrt(d_k) = 1 / 8, :65 -- a power of two, folded exactly), ``so`` its output projection, ``cq`` / ``co`` the
    query and output projections of the source attention, ``ff1`` / ``ff2`` with d_ff = 2024 zero-padded to 2048 (ReLU(0) = 0).
    ``kv``: the key / value projections of the five source attentions as ONE 512 -> 5 x 1024 GEMM over the feature sequence
    ([k_l / 8 | v_l] per layer, computed once per table).  ``cell_in`` = cell_input_fc (K = 1024: [embedding | x_i]); the three
    classifiers zero-padded to multiples of 64 outputs; embeddings pre-multiplied by sqrt(d_model) in fp32 (the reference's own
    fp32 product, :26); the position table as the reference computes it."""
    import math
    bl = _Blob(x3, fmt)
    x3 = bl.x3
    d = MTL_D

    def W(key):
        return sd[key + ".weight"].float(), sd[key + ".bias"].float()

    def lin(name, w, b, n_to=None, cin_to=None):
        n_to = n_to or w.shape[0]
        cin_to = cin_to or w.shape[1]
        bl.add_conv(name, *_pad_conv(w.reshape(w.shape[0], w.shape[1], 1, 1), b, n_to, cin_to))

    n_shared = 0
    while f"layers.{n_shared}.self_attn.linears.0.weight" in sd:
        n_shared += 1
    assert n_shared == 2, "the engine's MtlTabNet decoder is built for N = 3 (two shared layers)"
    src = {"l0": "layers.0", "l1": "layers.1", "cls": "cls_layer.0", "bbox": "bbox_layer.0", "cell": "cell_layer.0"}
    # TableMasterDecoder (master_decoder.py:532-563) is this decoder without the cell-content layer: its slot of the key / value projection stays (zero
    # weights: the tensor's channel layout is the engine's), its tensors are absent and the meta tensor says "0 cell classes"
    has_cell = "cell_layer.0.self_attn.linears.0.weight" in sd
    kv_w, kv_b = [], []
    d_ff = sd["layers.0.feed_forward.w_1.weight"].shape[0]
    ffp = (d_ff + 63) // 64 * 64
    for q in MTL_LAYERS:
        if q == "cell" and not has_cell:
            kv_w += [torch.zeros(d, d), torch.zeros(d, d)]
            kv_b += [torch.zeros(d), torch.zeros(d)]
            continue
        p = src[q]
        (wq, bq), (wk, bk), (wv, bv), (wo, bo) = [W(f"{p}.self_attn.linears.{i}") for i in range(4)]
        lin(q + ".qkv", torch.cat([wq, wk / 8.0, wv], 0), torch.cat([bq, bk / 8.0, bv], 0))
        lin(q + ".so", wo, bo)
        (wq, bq), (wk, bk), (wv, bv), (wo, bo) = [W(f"{p}.src_attn.linears.{i}") for i in range(4)]
        lin(q + ".cq", wq, bq)
        lin(q + ".co", wo, bo)
        kv_w += [wk / 8.0, wv]
        kv_b += [bk / 8.0, bv]
        lin(q + ".ff1", *W(p + ".feed_forward.w_1"), n_to=ffp)
        lin(q + ".ff2", *W(p + ".feed_forward.w_2"), cin_to=ffp)
        for i in range(3):
            bl.add(f"{q}.ln{i}.g", sd[f"{p}.sublayer.{i}.norm.weight"].float().numpy(), "f32")
            bl.add(f"{q}.ln{i}.b", sd[f"{p}.sublayer.{i}.norm.bias"].float().numpy(), "f32")
    lin("kv", torch.cat(kv_w, 0), torch.cat(kv_b, 0))
    ncls, ncell = sd["cls_fc.weight"].shape[0], sd["cell_fc.weight"].shape[0] if has_cell else 0
    lin("cls_fc", *W("cls_fc"), n_to=(ncls + 63) // 64 * 64)
    lin("bbox_fc", *W("bbox_fc.0"), n_to=64)
    if has_cell:
        lin("cell_fc", *W("cell_fc"), n_to=(ncell + 63) // 64 * 64)
        lin("cell_in", *W("cell_input_fc"))
        bl.add("emb_cell", (sd["embedding_cell.lut.weight"].float() * math.sqrt(d)).numpy(), "f32")
    bl.add("norm.g", sd["norm.weight"].float().numpy(), "f32")
    bl.add("norm.b", sd["norm.bias"].float().numpy(), "f32")
    bl.add("emb", (sd["embedding.lut.weight"].float() * math.sqrt(d)).numpy(), "f32")
    bl.add("pe", mtl_positional_table().numpy(), "f32")
    tc = cfg.get("idx_tag_cell", [0, 0])
    cc = [cfg.get(k, 0) if has_cell else 0 for k in ("sos_cell", "eos_cell", "pad_cell", "max_len_cell")]
    bl.add("meta", np.array([ncls, ncell, cfg["sos"], cfg["eos"], cfg["pad"], cfg["max_len"], cc[0], cc[1], cc[2], cc[3], tc[0], tc[1], ffp],
                            dtype=np.int32), "i32")
    return bl.tobytes()
