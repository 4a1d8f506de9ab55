"""PT_PRECISION_F16 -- single-pass IEEE half, the reference's own GPU arithmetic (base_infer_task.py:56-57 precision="fp16",
utils/deploy_utils.py:227-240 model.half()) -- at operator level: the storage format of csrc/act16.h (round-to-nearest-even, saturation at
+-65504 instead of Inf), the fp16 instantiation of the MFMA conv kernels against torch fp32 on identical operands, the blob-format guard,
and the stores of the element-wise kernels.  The nets at BASELINE sizes against the fp32 oracle: tests/test_gpu_fullsize.py [f16]."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pdf_table_amd import lib as L
from pdf_table_amd.weights import tile_conv_weight, to_bf16_bits

pytestmark = pytest.mark.gpu

F16_MAX = 65504.0


@pytest.fixture(scope="module")
def eng16():
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.set_precision(L.PT_PRECISION_F16)
    assert e.act_dtype == torch.float16 and e.weight_fmt == "f16" and not e.split
    yield e
    e.close()


def _h(t):
    return t.to(torch.float16).to(torch.float32)


def _conv(eng, x, w, b, ks, stride=1, relu=False, res=None, res_mode=0):
    dev = torch.device("cuda", 0)
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(dev)
    wt = torch.from_numpy(tile_conv_weight(w, "f16").view(np.int16)).to(dev)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(dev)
    out = eng.op_conv2d(xd, wt, b.to(dev), ks, stride, relu=relu, res=rd, res_mode=res_mode)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16
    return out.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("case", [
    dict(B=1, H=16, W=64, Cin=64, N=64, ks=3, stride=1),
    dict(B=2, H=37, W=45, Cin=64, N=128, ks=3, stride=1, relu=True, res_mode=1),
    dict(B=1, H=30, W=30, Cin=512, N=512, ks=3, stride=1, relu=True),
    dict(B=2, H=38, W=70, Cin=64, N=128, ks=3, stride=2, relu=True),
    dict(B=2, H=33, W=47, Cin=256, N=512, ks=3, stride=2, relu=True),
    dict(B=2, H=24, W=40, Cin=256, N=256, ks=1, stride=1, res_mode=2),
    dict(B=1, H=15, W=17, Cin=512, N=256, ks=1, stride=1),
    dict(B=1, H=38, W=70, Cin=64, N=128, ks=1, stride=2),
    dict(B=1, H=64, W=64, Cin=64, N=64, ks=3, stride=1, relu=True),       # the weight-stationary 64 -> 64 kernel's shape class
])
def test_conv_f16_vs_torch_fp32(eng16, case):
    """fp16 operands (values that are exactly representable), fp32 accumulate, ONE fp16 rounding of the result: within half an fp16 ulp
    (2^-11 relative) + fp32 summation noise of torch's fp32 convolution on the same operands"""
    g = torch.Generator().manual_seed(1000 + case["Cin"] + case["N"] + case["ks"])
    B, H, W, Cin, N, ks, stride = (case[k] for k in ("B", "H", "W", "Cin", "N", "ks", "stride"))
    x = _h(torch.randn(B, Cin, H, W, generator=g))
    w = _h(torch.randn(N, Cin, ks, ks, generator=g) * (2.0 / (Cin * ks * ks)) ** 0.5)
    b = torch.randn(N, generator=g) * 0.1
    ref = F.conv2d(x, w, b, stride=stride, padding=ks // 2)
    res = None
    rm = case.get("res_mode", 0)
    if rm == 1:
        res = _h(torch.randn(ref.shape, generator=g))
        ref = ref + res
    elif rm == 2:
        res = _h(torch.randn(B, N, ref.shape[2] // 2, ref.shape[3] // 2, generator=g))
        ref = ref + F.interpolate(res, scale_factor=2, mode="nearest")
    if case.get("relu"):
        ref = F.relu(ref)
    got = _conv(eng16, x, w, b, ks, stride, relu=case.get("relu", False), res=res, res_mode=rm)
    assert got.shape == ref.shape
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -11 + 2e-4
    assert bool((err <= tol).all()), f"max err {err.max().item()} at ref {ref.flatten()[err.argmax()].item()}"
    # and it is NOT the bf16 instantiation reading the same bits: eight times finer than half a bf16 ulp on the large values
    big = ref.abs() > 0.5
    assert float((err[big] / ref.abs()[big]).max()) <= 2.0 ** -10


def test_store_saturates_instead_of_inf(eng16):
    """an activation beyond the half range is stored as +-65504, never as Inf (a trained net never gets there; a broken one must not
    poison the next layer's accumulators with Inf - Inf = NaN): conv epilogue with and without ReLU, residual add, element-wise add"""
    dev = torch.device("cuda", 0)
    B, H, W, C = 1, 8, 32, 64
    x = torch.full((B, C, H, W), 30.0)
    w = torch.zeros(64, C, 1, 1)
    w[torch.arange(64), torch.arange(64), 0, 0] = _h(torch.linspace(-4096.0, 4096.0, 64))     # outputs -122 880 .. 122 880
    b = torch.zeros(64)
    ref = F.conv2d(x, w, b)
    for relu in (False, True):
        got = _conv(eng16, x, w, b, 1, relu=relu)
        r = F.relu(ref) if relu else ref
        assert bool(torch.isfinite(got).all()), "Inf / NaN stored"
        assert torch.equal(got, _h(r.clamp(-F16_MAX, F16_MAX)))
        assert float(got.max()) == F16_MAX and (relu or float(got.min()) == -F16_MAX)
    # residual add pushing a finite product over the edge
    w2 = torch.zeros(64, C, 1, 1)
    w2[torch.arange(64), torch.arange(64), 0, 0] = 2000.0                                   # 60 000: finite
    res = torch.full((B, 64, H, W), 60000.0)
    got = _conv(eng16, x, w2, b, 1, res=res, res_mode=1)
    assert bool(torch.isfinite(got).all()) and float(got.min()) == F16_MAX == float(got.max())
    # element-wise add kernel (graph_ops / layout kernels' store path)
    a = torch.full((1, 4, 4, 32), 60000.0, dtype=torch.float16, device=dev)
    s = eng16.op_add(a, a)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(s.float()).all()) and float(s.float().max()) == F16_MAX
    s = eng16.op_add(-a, -a)
    torch.cuda.synchronize()
    assert float(s.float().min()) == -F16_MAX
    # the other kernel families' stores (the saturation is MODE.FP16_OVFL, set at the top of EVERY kernel: csrc/act16.h a16_kernel_enter)
    m = torch.full((1, 4, 4, 32), 300.0, dtype=torch.float16, device=dev)
    s = eng16.op_mul(m, m)                                                  # 90 000 (graph_ops.hip)
    torch.cuda.synchronize()
    assert float(s.float().max()) == F16_MAX == float(s.float().min())
    wt = torch.zeros(9, 32, device=dev)
    wt[4] = 3.0
    s = eng16.op_dwconv(torch.full((1, 8, 8, 32), 30000.0, dtype=torch.float16, device=dev), wt, torch.zeros(32, device=dev), 3)      # depthwise: 90 000 (layout_kernels.hip)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(s.float()).all()) and float(s.float().max()) == F16_MAX


def test_blob_format_guard(eng16):
    """a bf16 blob under PT_PRECISION_F16 (and an fp16 blob under PT_PRECISION_BF16) is refused by the forward call, loudly"""
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.synth_weights import db_resnet18_state_dict
    from pdf_table_amd.weights import pack_db_resnet18
    sd = db_resnet18_state_dict(seed=0)
    x16 = torch.zeros(1, 64, 64, 4, dtype=torch.float16, device="cuda")
    eng16.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd, x3=False))            # bf16 tiles
    with pytest.raises(L.PtError, match="bf16 tiles"):
        eng16.det_forward_net(x16)
    eng16.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd, fmt="f16"))
    prob, _ = eng16.det_forward_net(x16, want_logits=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(prob).all())
    e = HipEngine(0)
    try:
        e.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd, fmt="f16"))
        with pytest.raises(L.PtError, match="fp16 tiles"):
            e.det_forward_net(torch.zeros(1, 64, 64, 4, dtype=torch.bfloat16, device="cuda"))
        # and the dtype of the tensors crossing the ABI follows the precision
        with pytest.raises(L.PtError, match="float16"):
            e.det_forward_net(x16)
    finally:
        e.close()


def test_det_preprocess_f16_is_the_half_rounding_of_the_oracle(eng16):
    """pt_det_preprocess under PT_PRECISION_F16: the same integer resize and float normalisation as in bf16 mode (bit-exact with the oracle
    there, tests/test_gpu_det.py), stored as the fp16 rounding of the fp32 value"""
    from oracle import db_pre
    from pdf_table_amd.synth_pages import make_page
    img = make_page(3, 1024)[0]
    chw, _ = db_pre.preprocess_db_pp(img)
    x = eng16.det_preprocess(torch.from_numpy(img[None]).cuda(), L.PT_DET_PRE_DB_PP)
    torch.cuda.synchronize()
    assert x.dtype == torch.float16 and tuple(x.shape) == (1, 960, 960, 4)
    ref = torch.from_numpy(np.ascontiguousarray(chw)).permute(1, 2, 0).to(torch.float16)
    assert torch.equal(x[0, :, :, :3].cpu(), ref)
    assert float(x[0, :, :, 3].float().abs().max()) == 0.0


def test_bits_helper_matches_torch_half():
    """weights.to_bf16_bits(fmt="f16") is torch's RNE half cast, saturating"""
    t = torch.tensor([0.1, -1.0, 65504.0, 70000.0, -1e9, 6e-8, 0.0])
    bits = to_bf16_bits(t, "f16")
    back = torch.from_numpy(bits.view(np.int16)).view(torch.float16).float()
    assert torch.equal(back, t.clamp(-F16_MAX, F16_MAX).to(torch.float16).float())


@pytest.mark.parametrize("C,N,H,W", [(64, 64, 13, 21), (128, 64, 37, 50), (128, 128, 19, 33), (256, 256, 16, 16), (96, 64, 11, 23)])
@pytest.mark.parametrize("relu,field", [(True, "wide"), (False, "local")])
def test_dcn_op_f16(eng16, C, N, H, W, relu, field):
    """pt_op_dcn in PT_PRECISION_F16 against oracle.lore_net.deform_conv2d on fp16-exact operands, with the offset fields of tests/test_gpu_dcn_op.py
    (multi-pixel, out-of-map, sub-pixel): the blend of namespace pt_f16 runs on mixed-precision FMAs (v_fma_mix_f32 / v_fma_mixlo|hi_f16, act16.h) -- fp32
    accumulation, ONE fp16 rounding per sampled column -- so the bounds are test_dcn_op_bf16's with the half ulp of fp16 (2^-11) in place of bf16's (2^-8)."""
    from oracle import lore_net
    from test_gpu_dcn_op import _case, _om
    B = 2
    x, w, b, off, mlog = _case(B, C, N, H, W, seed=C * 1000 + N + H, field=field, rounded=False)
    x, w = _h(x), _h(w)
    mask = torch.sigmoid(mlog)
    cols = lore_net.deform_conv2d(x, off, mask, w, None, return_cols=True)
    ref = lore_net.deform_conv2d(x, off, mask, w, b)
    ref_r = torch.einsum("ock,bckhw->bohw", w.reshape(N, C, 9).double(), _h(cols).double()).float() + b.view(1, N, 1, 1)
    if relu:
        ref, ref_r = torch.relu(ref), torch.relu(ref_r)
    dev = torch.device("cuda", 0)
    w1 = w.permute(0, 2, 3, 1).reshape(N, 9 * C, 1, 1).contiguous()
    out = eng16.op_dcn(x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(dev), _om(off, mlog).to(dev),
                       torch.from_numpy(tile_conv_weight(w1, "f16").view(np.int16)).to(dev), b.to(dev), relu=relu)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    col_mag = torch.einsum("ock,bckhw->bohw", w.reshape(N, C, 9).abs(), cols.abs())
    err_a, err_b = (got - ref_r).abs(), (got - ref).abs()
    tol_a = ref_r.abs() * 2.0 ** -11 + col_mag * 2.0 ** -15 + 1e-4
    tol_b = ref.abs() * 2.0 ** -11 + col_mag * 2.0 ** -12 + 1e-4
    print(f"dcn f16 {field} {C}->{N} @{H}x{W}: max err vs rounded-column oracle {err_a.max().item():.3e}, vs oracle {err_b.max().item():.3e}, scale {ref.abs().max().item():.2f}")
    assert bool((err_a <= tol_a).all()), f"vs rounded-column oracle: max err {err_a.max().item()} (tol there {tol_a.flatten()[err_a.argmax()].item()})"
    assert bool((err_b <= tol_b).all()), f"vs oracle: max err {err_b.max().item()}"
