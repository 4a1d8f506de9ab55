"""Op-level GPU parity of the fused modulated deformable convolution (pt_op_dcn, ABI 12) against oracle.lore_net.deform_conv2d
(itself bit-exact against the reference's own DCNv2 CPU im2col, tests/test_oracle_lore.py).

Reference: model/lore/dcnv2.py:71-86 (offset / mask channel order, sigmoid on the mask), sampling rule
model/lore/DCNv2_latest/src/cuda/dcn_v2_im2col_cuda.cu:121-191 = src/cpu/dcn_v2_im2col_cpu.cpp:26-55,123-190: bilinear, a sample
outside (-1, H) x (-1, W) is zero, and each of the four corners is bounds-checked on its own.

The whole-network tests only ever produce sub-pixel offsets (synthetic weights); a trained Lore checkpoint moves samples by
pixels.  Here: offsets ~ N(0, 3 px) with a tail far beyond the map, mask logits ~ N(0, 2), C in {32 .. 512}, maps whose size is
not a multiple of the kernel's 8 x 16 pixel tile -- so every branch of the geometry table (all four corners in, 1-3 corners out,
whole sample out) and both kernels (dcn_fused64_kernel for C % 64 == 0 with 64- and 128-wide blocks, dcn_fused_kernel otherwise)
run in both precision modes.
"""
import numpy as np
import pytest
import torch

from oracle import lore_net
from pdf_table_amd.weights import tile_conv_weight, tile_conv_weight_x3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _case(B, C, N, H, W, seed, field="wide", rounded=True):
    """field "wide": offsets ~ N(0, 3 px) + a 6 % tail of jumps beyond the map (what the bench's random-init Lore detector produces: |offset|
    2 .. 9 px on average, up to 40 px); "local": offsets ~ N(0, 0.6 px), no tail (the sub-pixel regime of the parity nets)."""
    sigma = 3.0 if field == "wide" else 0.6
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g).abs()          # post-ReLU activations, as every DCN input in DLASeg is
    w = torch.randn(N, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    if rounded:
        x, w = _bf16(x), _bf16(w)
    off = torch.randn(B, 18, H, W, generator=g) * sigma
    # a tail that leaves the map: 6 % of the taps get an offset of +-(0.5 .. 1.5) map sizes, some land exactly on -1 / H
    far = torch.rand(B, 18, H, W, generator=g) < (0.06 if field == "wide" else 0.0)
    jump = (torch.rand(B, 18, H, W, generator=g) + 0.5) * max(H, W) * torch.sign(torch.randn(B, 18, H, W, generator=g))
    off = torch.where(far, jump, off)
    off[:, :, 0, 0] = torch.tensor([-1.0, 0.0] * 9)         # tap (0,0) of pixel (0,0): h_im = -2 (out); others at integer positions
    off[:, :, -1, -1] = torch.tensor([1.0, 1.0] * 9)        # bottom-right pixel: h_im up to H + 1 (out), H - 1 + 0 (edge)
    off[:, 0::2, 1, 1] = -0.5                               # half-pixel positions straddling the top border
    mlog = torch.randn(B, 9, H, W, generator=g) * 2.0
    return x, w, b, off, mlog


def _om(off, mlog):
    B, _, H, W = off.shape
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = mlog.permute(0, 2, 3, 1)
    om[..., 27:] = 123.0                                    # padding lanes must be ignored
    return om.contiguous()


SHAPES = [
    # C, N, H, W            kernel reached in bf16 mode / in BF16X3 mode
    (64, 64, 13, 21),      # dcn_fused64<64,512,0> / <64,512,1>; partial tiles in both directions
    (64, 64, 8, 16),       # exactly one tile
    (128, 64, 37, 50),     # two 64-channel stages per tap
    (128, 128, 19, 33),    # <128,256,0> / <128,512,1>
    (256, 128, 9, 70),
    (256, 256, 16, 16),    # two 128-wide blocks
    (512, 256, 7, 11),     # eight stages per tap
    (96, 64, 11, 23),      # C % 64 != 0: dcn_fused_kernel<.,64>
    (32, 128, 10, 17),     # dcn_fused_kernel<0,128> / <1,64>
]


@pytest.mark.parametrize("C,N,H,W", SHAPES)
@pytest.mark.parametrize("relu,field", [(True, "wide"), (False, "wide"), (True, "local")])
@pytest.mark.parametrize("blend", ["mfma", "mfma2", "valu"])
def test_dcn_op_bf16(eng, C, N, H, W, relu, field, blend):
    """bf16 mode: operands exactly representable; the kernel rounds every sampled column to bf16 before the product, so it is compared
    (a) with the oracle whose columns are rounded the same way -- difference = fp32 summation order + the output's own bf16 rounding --
    and (b) with the un-rounded oracle within the half-ulp-per-column bound.
    blend "valu" (the default): dcn_fused64_kernel (fp32 bilinear x mask weights).  blend "mfma" (pt_engine_set_dcn_mfma): dcn_mfma_kernel
    blends on the matrix pipe against the four weights ROUNDED TO bf16 (2^-9 relative each): a column is then within 2^-9 of its magnitude
    plus its own half ulp, so (a) and (b) carry one more column-ulp term; C % 64 != 0 shapes run dcn_fused_kernel in both settings."""
    B = 2
    eng.set_dcn_mfma({"valu": 0, "mfma": 1, "mfma2": 2}[blend])      # (mfma2: the 64-output shapes; the others run the VALU blend)
    wq = 2.0 ** -8 if blend != "valu" else 0.0      # extra column error of the bf16 weights, in units of the column magnitudes
    x, w, b, off, mlog = _case(B, C, N, H, W, seed=C * 1000 + N + H, field=field)
    mask = torch.sigmoid(mlog)
    cols = lore_net.deform_conv2d(x, off, mask, w, None, return_cols=True)          # [B,C,9,H,W] fp32
    ref = lore_net.deform_conv2d(x, off, mask, w, b)
    ref_r = torch.einsum("ock,bckhw->bohw", w.reshape(N, C, 9).double(), _bf16(cols).double()).float() + b.view(1, N, 1, 1)
    if relu:
        ref, ref_r = torch.relu(ref), torch.relu(ref_r)
    dev = torch.device("cuda", 0)
    w1 = w.permute(0, 2, 3, 1).reshape(N, 9 * C, 1, 1).contiguous()
    out = eng.op_dcn(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev), _om(off, mlog).to(dev),
                     torch.from_numpy(tile_conv_weight(w1).view(np.int16)).to(dev), b.to(dev), relu=relu)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    # (a) same-rounding oracle: output bf16 rounding (2^-9 relative) + a column that rounds the other way because the kernel's blend
    # uses fused multiply-adds (<= 1 bf16 ulp of ONE column times its weight: bounded through the column magnitudes)
    col_mag = torch.einsum("ock,bckhw->bohw", w.reshape(N, C, 9).abs(), cols.abs())
    err_a = (got - ref_r).abs()
    tol_a = ref_r.abs() * 2.0 ** -8 + col_mag * (2.0 ** -12 + wq) + 1e-4
    assert bool((err_a <= tol_a).all()), f"vs rounded-column oracle: max err {err_a.max().item()} (tol there {tol_a.flatten()[err_a.argmax()].item()})"
    # (b) un-rounded oracle: every column within half a bf16 ulp
    err_b = (got - ref).abs()
    tol_b = ref.abs() * 2.0 ** -8 + col_mag * (2.0 ** -9 + wq) + 1e-4
    assert bool((err_b <= tol_b).all()), f"vs oracle: max err {err_b.max().item()}"
    eng.set_dcn_mfma(False)
    print(f"dcn bf16 [{blend}] {field} {C}->{N} @{H}x{W}: max err vs rounded-column oracle {err_a.max().item():.3e}, vs oracle {err_b.max().item():.3e}, scale {ref.abs().max().item():.2f}")


@pytest.mark.parametrize("C,N,H,W", SHAPES)
def test_dcn_op_bf16x3(eng, C, N, H, W):
    """BF16X3 (hi | lo operands, three MFMA passes): within 1e-3 of the fp64 evaluation on UN-rounded fp32 operands (north_star)."""
    B = 2
    x, w, b, off, mlog = _case(B, C, N, H, W, seed=C * 1000 + N + H + 1, rounded=False)
    ref = torch.relu(lore_net.deform_conv2d(x.double(), off.double(), torch.sigmoid(mlog.double()), w.double(), b.double())).float()
    dev = torch.device("cuda", 0)
    t = x.permute(0, 2, 3, 1)
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    w1 = w.permute(0, 2, 3, 1).reshape(N, 9 * C, 1, 1).contiguous()
    out = eng.op_dcn(torch.cat([hi, lo], -1).contiguous().to(dev), _om(off, mlog).to(dev),
                     torch.from_numpy(tile_conv_weight_x3(w1).view(np.int16)).to(dev), b.to(dev), relu=True, split=1)
    torch.cuda.synchronize()
    o = out.float().cpu()
    got = (o[..., :N] + o[..., N:]).permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    print(f"dcn x3 {C}->{N} @{H}x{W}: max abs err {err:.3e}, ref scale {ref.abs().max().item():.2f}")
    assert err <= 1e-3 * max(1.0, ref.abs().max().item())
    assert err <= 1e-3, "absolute 1e-3 (outputs here are O(1..5))"


def test_dcn_op_zero_offsets_is_a_plain_conv(eng):
    """Known answer: zero offsets and mask == 1 (logit +inf -> 40) make the operator an ordinary 3x3 convolution."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    B, C, N, H, W = 1, 64, 64, 20, 28
    x = _bf16(torch.randn(B, C, H, W, generator=g))
    w = _bf16(torch.randn(N, C, 3, 3, generator=g) * 0.05)
    b = torch.randn(N, generator=g) * 0.1
    om = torch.zeros(B, H, W, 32)
    om[..., 18:27] = 40.0
    dev = torch.device("cuda", 0)
    w1 = w.permute(0, 2, 3, 1).reshape(N, 9 * C, 1, 1).contiguous()
    out = eng.op_dcn(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev), om.to(dev),
                     torch.from_numpy(tile_conv_weight(w1).view(np.int16)).to(dev), b.to(dev), relu=False)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(x, w, b, 1, 1)
    err = (got - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-4).all()), err.max().item()


def test_dcn_op_rejects_bad_arguments(eng):
    from pdf_table_amd.lib import PtError
    dev = torch.device("cuda", 0)
    x = torch.zeros(1, 8, 16, 48, dtype=torch.bfloat16, device=dev)        # C = 48: not a multiple of 32
    om = torch.zeros(1, 8, 16, 32, device=dev)
    with pytest.raises(PtError):
        eng.op_dcn(x, om, torch.zeros(64 * 9 * 48, dtype=torch.int16, device=dev), torch.zeros(64, device=dev))
    with pytest.raises(ValueError):
        eng.op_dcn(torch.zeros(1, 8, 16, 64, dtype=torch.bfloat16, device=dev), torch.zeros(1, 8, 16, 27, device=dev),
                   torch.zeros(64 * 9 * 64, dtype=torch.int16, device=dev), torch.zeros(64, device=dev))
