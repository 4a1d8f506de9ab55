"""PT_PRECISION_F16 on the model families beside the four headline nets (those: tests/test_gpu_fullsize.py [f16], tests/test_gpu_e2e.py): every
translation unit of the library is instantiated for IEEE-half storage (csrc/act16.h), so every family must run in it -- DB-ProxylessNAS, the PP-LCNet
classifiers, the Lore ResNet-18 ("wireless") detector, the Lore processor, the MtlTabNet backbone and decoders.  Each case: the fp32 oracle on seeded
weights, the engine in bf16 and in f16 on the SAME weights (packed for each format), the f16 drift inside a bound several times tighter than bf16's
test bound and below the measured bf16 drift."""
import os

import numpy as np
import pytest
import torch

from pdf_table_amd import lib as L

pytestmark = pytest.mark.gpu


def _engines():
    from pdf_table_amd.engine import HipEngine
    a, b = HipEngine(0), HipEngine(0)
    b.set_precision(L.PT_PRECISION_F16)
    return a, b


def _x4(x, dtype):
    n, _, H, W = x.shape
    x4 = torch.zeros(n, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    return x4.to(dtype).cuda()


def _report(tag, d16, dbf, scale):
    print(f"F16 FAMILY {tag}: drift f16 {d16 / scale:.3e}, bf16 {dbf / scale:.3e} of scale {scale:.2f}")


def test_db_proxylessnas():
    from oracle import db_nas
    from pdf_table_amd.synth_weights import db_nas_state_dict
    from pdf_table_amd.weights import pack_db_nas
    sd = db_nas_state_dict(seed=13)
    x = torch.randn(2, 3, 96, 224, generator=torch.Generator().manual_seed(5))
    ref = db_nas.dbnas_forward_fp32(sd, x, return_logits=True)[:, 0]
    eb, eh = _engines()
    try:
        eb.load_weights(L.PT_MODEL_DB_NAS, pack_db_nas(sd, x3=False))
        eh.load_weights(L.PT_MODEL_DB_NAS, pack_db_nas(sd, fmt="f16"))
        lb = eb.det_forward_net(_x4(x, torch.bfloat16), want_logits=True)[1].cpu()
        lh = eh.det_forward_net(_x4(x, torch.float16), want_logits=True)[1].cpu()
    finally:
        eb.close()
        eh.close()
    scale = max(1.0, ref.abs().max().item())
    d16, dbf = (lh - ref).abs().max().item(), (lb - ref).abs().max().item()
    _report("DB-ProxylessNAS logits", d16, dbf, scale)
    assert d16 <= 0.02 * scale and d16 < dbf


def test_pplcnet_classifier():
    from cls_synth import CLS_GOLDEN_TASKS, cls_inputs
    from oracle import pplcnet
    from pdf_table_amd.synth_weights import pplcnet_state_dict
    from pdf_table_amd.weights import pack_pplcnet
    task = "textline_orientation"
    cn, textline, hw, seed = CLS_GOLDEN_TASKS[task]
    sd = pplcnet_state_dict(seed, cn)
    x = torch.from_numpy(cls_inputs(seed + 50, 37, hw))
    ref = pplcnet.pplcnet_forward(sd, x, textline=textline).numpy()
    eb, eh = _engines()
    try:
        eb.load_weights(L.PT_MODEL_PPLCNET, pack_pplcnet(sd, x3=False))
        eh.load_weights(L.PT_MODEL_PPLCNET, pack_pplcnet(sd, fmt="f16"))
        gb = eb.cls_forward_net(_x4(x, torch.bfloat16), slot=0, textline=textline).cpu().numpy()
        gh = eh.cls_forward_net(_x4(x, torch.float16), slot=0, textline=textline).cpu().numpy()
    finally:
        eb.close()
        eh.close()
    scale = max(1.0, float(np.abs(ref).max()))
    d16, dbf = float(np.abs(gh - ref).max()), float(np.abs(gb - ref).max())
    _report("PP-LCNet logits", d16, dbf, scale)
    assert d16 <= 0.01 * scale and d16 < dbf


def test_lore_wireless_detector_and_processor():
    from oracle import lore_net
    from oracle import lore_processor as op
    from pdf_table_amd.synth_weights import lore_processor_state_dict, lore_wireless_state_dict
    from pdf_table_amd.weights import pack_lore_processor, pack_lore_wireless
    sd = lore_wireless_state_dict(seed=23)
    psd = lore_processor_state_dict(seed=31)
    x = torch.randn(1, 3, 192, 256, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = lore_net.lore_wireless_forward(sd, x)
    g = torch.Generator().manual_seed(77)
    counts = [150, 60]
    logi = torch.zeros(2, L.PT_TSR_MAX_CELLS, 256)
    dets = torch.zeros(2, L.PT_TSR_MAX_CELLS, 9)
    for t, c in enumerate(counts):
        logi[t, :c] = torch.randn(c, 256, generator=g)
        dets[t, :c, :8] = torch.rand(c, 8, generator=g) * 250
    with torch.no_grad():
        pref = [op.processor_forward(psd, logi[t:t + 1, :c], None)[1][0] for t, c in enumerate(counts)]
    eb, eh = _engines()
    out = {}
    try:
        for nm, e, dt, kw in (("bf16", eb, torch.bfloat16, dict(x3=False)), ("f16", eh, torch.float16, dict(fmt="f16"))):
            e.load_weights(L.PT_MODEL_LORE_RESNET18, pack_lore_wireless(sd, **kw))
            e.load_weights(L.PT_MODEL_LORE_PROCESSOR, pack_lore_processor(psd, **kw))
            heads = e.tsr_forward_net(_x4(x, dt), wireless=True)
            _, stacked = e.tsr_process(logi.cuda(), dets.cuda(), np.asarray(counts, np.int32))
            torch.cuda.synchronize()
            out[nm] = ({k: heads[k].cpu().permute(0, 3, 1, 2) for k in ref}, stacked.cpu())
    finally:
        eb.close()
        eh.close()
    worst = {}
    for nm, (heads, _) in out.items():
        worst[nm] = max((heads[k] - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item()) for k in ref)
    _report("Lore wireless heads (worst, relative)", worst["f16"], worst["bf16"], 1.0)
    assert worst["f16"] <= 0.03 and worst["f16"] < worst["bf16"]
    pw = {}
    for nm, (_, stacked) in out.items():
        pw[nm] = max((stacked[t, :c] - pref[t]).abs().max().item() for t, c in enumerate(counts))
    ps = max(1.0, max(float(p.abs().max()) for p in pref))
    _report("Lore processor stacked logits", pw["f16"], pw["bf16"], ps)
    assert pw["f16"] <= 0.02 * ps and pw["f16"] < pw["bf16"]


def test_mtl_tabnet_backbone_and_decoders(golden_dir):
    from oracle import mtl_tabnet as omt
    from pdf_table_amd.synth_weights import mtl_tabnet_backbone_state_dict, mtl_tabnet_decoder_state_dict
    from pdf_table_amd.weights import pack_mtl_backbone, pack_mtl_decoder
    from test_gpu_mtl import BASE_CFG, _decoder_inputs, _oracle_decode
    g = np.load(os.path.join(golden_dir, "mtl_tabnet_backbone.npz"))
    sd = mtl_tabnet_backbone_state_dict(seed=int(g["seed"]))
    x = g["x"]
    with torch.no_grad():
        want = omt.backbone_forward_fp32(sd, torch.from_numpy(x))[2]
    gd, fmap = _decoder_inputs(golden_dir)
    dsd = mtl_tabnet_decoder_state_dict(seed=int(gd["seed"]), num_classes=43, num_classes_cell=60)
    cfg = dict(BASE_CFG)
    dwant = _oracle_decode(dsd, fmap, cfg)
    eb, eh = _engines()
    res = {}
    try:
        for nm, e, kw in (("bf16", eb, dict(x3=False)), ("f16", eh, dict(fmt="f16"))):
            e.load_weights(L.PT_MODEL_MTL_BACKBONE, pack_mtl_backbone(sd, **kw))
            e.load_weights(L.PT_MODEL_MTL_DECODER, pack_mtl_decoder(dsd, cfg, **kw))
            f3 = e.mtl_backbone_forward(torch.from_numpy(x).cuda()).cpu()
            ff = torch.from_numpy(fmap).permute(0, 2, 3, 1).reshape(fmap.shape[0], -1, 512).contiguous().cuda()
            dec = e.mtl_decode(ff)
            torch.cuda.synchronize()
            res[nm] = (f3, dec["tag_logits"].cpu().numpy(), [int(v) for v in dec["lens"]])
    finally:
        eb.close()
        eh.close()
    scale = want.abs().max().item()
    d16, dbf = (res["f16"][0] - want).abs().max().item(), (res["bf16"][0] - want).abs().max().item()
    _report("MtlTabNet backbone f3", d16, dbf, scale)
    assert d16 <= 0.02 * scale and d16 < dbf
    # decoders: the first structure position (identical inputs for every arithmetic: <SOS> + the same feature map) against the oracle's logits
    ts = max(float(np.abs(w[0]).max()) for w in dwant)
    e16 = max(float(np.abs(res["f16"][1][b, 0] - dwant[b][0][0]).max()) for b in range(len(dwant)))
    ebf = max(float(np.abs(res["bf16"][1][b, 0] - dwant[b][0][0]).max()) for b in range(len(dwant)))
    _report("MtlTabNet structure decoder, first position", e16, ebf, ts)
    assert e16 <= 0.02 * ts and e16 < ebf and all(n > 0 for n in res["f16"][2])
