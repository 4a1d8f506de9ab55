"""GPU parity tests of the DB-ProxylessNAS detector (`DBNasModel`, db_net/dbnet.py:693-712) through the C ABI.

PT_PRECISION_BF16X3 within 1e-3 (probability map) of the oracle's fp32 restatement and of the tensors the reference's
own module produced (tests/golden/db_nas.npz); PT_PRECISION_BF16 bounded end-to-end drift (see test_gpu_det.py for why
a 50-layer bf16 graph cannot agree to 1e-3 with any other evaluation order)."""
import os

import numpy as np
import pytest
import torch

from oracle import db_nas
from pdf_table_amd import lib as L
from pdf_table_amd.synth_weights import db_nas_state_dict, db_resnet18_state_dict
from pdf_table_amd.weights import pack_db_nas, pack_db_resnet18
from test_gpu_det import _x4

pytestmark = pytest.mark.gpu

TOL_PROB = 1e-3          # north_star tolerance on the float output
BF16_E2E_PROB = 0.1      # throughput mode: bounded drift only


@pytest.fixture(scope="module")
def nas_sd():
    return db_nas_state_dict(seed=13)


@pytest.fixture(scope="module")
def eng_nas(nas_sd):
    from pdf_table_amd.engine import HipEngine
    e = HipEngine(0)
    e.load_weights(L.PT_MODEL_DB_NAS, pack_db_nas(nas_sd))
    yield e
    e.close()


@pytest.fixture()
def eng_nas_x3(eng_nas):
    eng_nas.set_precision(L.PT_PRECISION_BF16X3)
    yield eng_nas
    eng_nas.set_precision(L.PT_PRECISION_BF16)


@pytest.mark.parametrize("shape", [(1, 128, 160), (2, 96, 224), (1, 256, 256)])
def test_nas_net_x3_matches_fp32_oracle(eng_nas_x3, nas_sd, shape):
    n, H, W = shape
    g = torch.Generator().manual_seed(300 + H)
    x = torch.randn(n, 3, H, W, generator=g)
    ref_logits = db_nas.dbnas_forward_fp32(nas_sd, x, return_logits=True)[:, 0]
    ref_prob = torch.sigmoid(ref_logits)
    prob, logits = eng_nas_x3.det_forward_net(_x4(x, split=True).cuda(), want_logits=True)
    torch.cuda.synchronize()
    dl = (logits.cpu() - ref_logits).abs().max().item()
    dp = (prob.cpu() - ref_prob).abs().max().item()
    print(f"DB-NAS x3 {shape}: max|dlogit|={dl:.3e} (scale {ref_logits.abs().max().item():.1f}), max|dprob|={dp:.3e}")
    assert dp <= TOL_PROB, dp
    assert dl <= 1e-3 * max(1.0, ref_logits.abs().max().item()), dl


def test_nas_net_x3_matches_reference_golden(eng_nas_x3, golden_dir):
    g = np.load(os.path.join(golden_dir, "db_nas.npz"))
    for tag in ("a", "b"):
        x = torch.from_numpy(g[f"x_{tag}"])
        prob = eng_nas_x3.det_forward_net(_x4(x, split=True).cuda()).cpu().numpy()
        d = np.abs(prob[0] - g[f"prob_{tag}"][0, 0]).max()
        print(f"HIP bf16x3 DB-NAS vs reference fp32 golden {tag}: max|dprob| = {d:.3e}")
        assert d <= TOL_PROB, d


@pytest.mark.parametrize("shape", [(1, 128, 160), (2, 256, 256)])
def test_nas_net_bf16_drift(eng_nas, nas_sd, shape):
    n, H, W = shape
    g = torch.Generator().manual_seed(300 + H)
    x = torch.randn(n, 3, H, W, generator=g).to(torch.bfloat16).float()
    ref = db_nas.dbnas_forward_fp32(nas_sd, x)[:, 0]
    prob = eng_nas.det_forward_net(_x4(x).cuda()).cpu()
    dp = (prob - ref).abs()
    print(f"DB-NAS bf16 {shape}: max|dprob|={dp.max().item():.3e} mean {dp.mean().item():.3e}")
    assert dp.max().item() <= BF16_E2E_PROB and dp.mean().item() <= 1e-2


def test_active_detector_is_the_one_loaded_last(eng_nas, nas_sd):
    """pt_det_forward* runs `DBModel` or `DBNasModel`, whichever was loaded last (modeling_db_net.py:47-52)."""
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    a = eng_nas.det_forward_net(_x4(x).cuda()).cpu()
    eng_nas.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(db_resnet18_state_dict(seed=11)))
    b = eng_nas.det_forward_net(_x4(x).cuda()).cpu()
    eng_nas.load_weights(L.PT_MODEL_DB_NAS, pack_db_nas(nas_sd))
    c = eng_nas.det_forward_net(_x4(x).cuda()).cpu()
    assert torch.equal(a, c) and not torch.equal(a, b)


def test_detection_task_proxylessnas_backbone():
    """OcrDetectionTask(model='db', backbone='proxylessnas') end to end on two synthetic pages: boxes come back in
    source pixels, and the probability map behind them is the NAS network's."""
    from pdf_table_amd.ocr_detection_task import OcrDetectionTask
    from pdf_table_amd.synth_pages import make_page
    pages = [make_page(i, 512)[0] for i in (3, 4)]
    task = OcrDetectionTask(model="db", backbone="proxylessnas", synthetic_seed=13)
    out = task(pages)
    assert len(out) == 2
    for boxes in out:
        assert boxes.ndim == 2 and boxes.shape[1] == 8
        if len(boxes):
            assert boxes.min() >= 0 and boxes.max() <= 512
