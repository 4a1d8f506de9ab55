"""tools/fp8_cvit_drift.py: end-to-end drift of the ConvNextViT recogniser with fp8 (OCP e4m3, the gfx950 MFMA format) operands on the ViT's qkv / proj / MLP
GEMMs only -- BASELINE.json configs[4] says "fp8 MFMA"; VERDICT r04 item 8 asks for the product option or the numbers that close the question.

CPU emulation of exactly what a PT_PRECISION_FP8 would compute on those GEMMs: both operands rounded to torch.float8_e4m3fn (round-to-nearest-even, the
hardware conversion's rounding) with a per-tensor power-of-two scale that maps the tensor's max to the top binade (the best case for a static scale),
products accumulated in fp32, everything else (ConvNext stem, LayerNorm, soft-max, P.V, residual stream, classifier) in fp32 -- i.e. LESS error than a
real mode, whose other tensors would be bf16.  Compared on the 64 text lines of tests/test_gpu_fullsize.py::test_fullsize_convnext_vit_64_lines_oracle_parity
with (a) the fp32 oracle and (b) the same emulation with bf16 operands on the same GEMMs.  Output: profiles/r05/fp8_cvit_drift.txt."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import convnext_vit as ocv                                  # noqa: E402
from oracle import crnn as ocrnn                                        # noqa: E402
from pdf_table_amd.synth_pages import make_page                         # noqa: E402
from pdf_table_amd.synth_weights import convnext_vit_state_dict         # noqa: E402


def q_e4m3(t):
    s = 2.0 ** torch.floor(torch.log2(448.0 / t.abs().max().clamp_min(1e-30)))     # power-of-two scale: max lands in [224, 448]
    return (t * s).to(torch.float8_e4m3fn).to(torch.float32) / s


def q_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class Patched:
    """oracle.convnext_vit's `F` with the ViT-width linears (192 <-> 192 / 576 / 768) running on quantised operands"""
    def __init__(self, q):
        self.q, self.n = q, 0

    def __getattr__(self, k):
        return getattr(F, k)

    def linear(self, x, w, b=None):
        if w.shape[1] in (ocv.VIT_DIM, 4 * ocv.VIT_DIM) and w.shape[0] in (ocv.VIT_DIM, 4 * ocv.VIT_DIM) and x.shape[-2] == ocv.VIT_TOKENS:
            self.n += 1
            return F.linear(self.q(x), self.q(w), b)
        return F.linear(x, w, b)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    sd = ocv.canonical_state_dict(convnext_vit_state_dict(seed=3))
    img, meta = make_page(1, 1024)
    l = meta["lines"].astype(np.float64)
    quads = np.stack([l[:, 0], l[:, 1], l[:, 2], l[:, 1], l[:, 2], l[:, 3], l[:, 0], l[:, 3]], 1)
    quads = np.concatenate([quads, quads + 3.0])[:64]
    xs = [ocv.chunk_preprocess(ocrnn.crop_image(img, ocrnn.order_point(q))) for q in quads]
    with torch.no_grad():
        ref = torch.cat([ocv.convnext_vit_forward_fp32(sd, x) for x in xs])
    top2 = torch.topk(ref, 2, dim=-1)
    scale = float(ref.abs().max())
    margin = top2.values[..., 0] - top2.values[..., 1]
    lines = [f"ConvNextViT, 64 lines, ViT qkv / proj / MLP GEMM operands quantised (CPU emulation, fp32 accumulate, everything else fp32); logit scale {scale:.1f}"]
    for name, q in (("bf16", q_bf16), ("fp8 e4m3 (per-tensor 2^k scale)", q_e4m3)):
        ocv.F = p = Patched(q)
        try:
            with torch.no_grad():
                got = torch.cat([ocv.convnext_vit_forward_fp32(sd, x) for x in xs])
        finally:
            ocv.F = F
        win = got.gather(-1, top2.indices[..., :1])[..., 0]
        d = float((win - top2.values[..., 0]).abs().max())
        diff = got.argmax(-1) != top2.indices[..., 0]
        lines.append(f"  {name}: {p.n // len(xs)} GEMMs per line quantised; max |d winning logit| {d:.3e} = {d / scale:.3e} of scale; {int(diff.sum())} of {diff.numel()} token ids "
                     f"differ (largest oracle margin among them {float(margin[diff].max()) if diff.any() else 0.0:.3e})")
    lines.append("  bound of tests/test_gpu_fullsize.py for the bf16 mode: 0.06 of scale, ids free inside a 0.12 margin; the engine's measured bf16 drift (all tensors bf16): 5.1e-3")
    print("\n".join(lines))
    with open(os.path.join(REPO, "profiles", "r05", "fp8_cvit_drift.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
