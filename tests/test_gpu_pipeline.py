"""Drop-in surfaces on the GPU: OcrDetectionTask / OcrRecognitionTask call shapes and OcrTablePipeline.predict()."""
import numpy as np
import pytest
import torch

from oracle import crnn as ocrnn
from pdf_table_amd import lib as L
from pdf_table_amd.synth_pages import make_page

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe():
    from pdf_table_amd.pipeline import OcrTablePipeline
    return OcrTablePipeline(device=0, detect_model="db", recognizer="CRNN", synthetic_seed=0)


def test_detection_task_call_shape(pipe):
    img = make_page(1)[0][:480, :640].copy()
    out = pipe.text_detector([img, img[:320]])
    assert isinstance(out, list) and len(out) == 2
    for r in out:
        assert isinstance(r, np.ndarray) and r.ndim == 2 and r.shape[1] == 8


def test_recognition_task_matches_oracle_ids(pipe):
    """already-cropped inputs (the reference call shape): resize + CRNN on the device, ids vs the oracle (x3 mode)."""
    from pdf_table_amd.synth_weights import crnn_state_dict
    eng = pipe.engine
    sd = crnn_state_dict(seed=1)
    rng = np.random.default_rng(0)
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in ((20, 130), (32, 640), (64, 300), (17, 900), (40, 12))]
    eng.set_precision(L.PT_PRECISION_BF16X3)
    try:
        ids, mx = eng.rec_forward_crops(crops)
        ids = ids.cpu().numpy()
        for i, c in enumerate(crops):
            x = ocrnn.rec_preprocess(c)
            with torch.no_grad():
                logits = ocrnn.crnn_forward_fp32(sd, x)[0]
            top2 = torch.topk(logits, 2, dim=-1)
            margin = (top2.values[:, 0] - top2.values[:, 1]).numpy()
            d = ids[i] != top2.indices[:, 0].numpy()
            assert (margin[d] <= 2e-3).all()
            assert np.abs(mx[i].cpu().numpy() - top2.values[:, 0].numpy()).max() <= 1e-3
    finally:
        eng.set_precision(L.PT_PRECISION_BF16)
    texts = pipe.text_recognizer(crops)
    assert isinstance(texts, list) and len(texts) == len(crops) and all(isinstance(t, str) for t in texts)


def test_pipeline_predict(pipe):
    pages = [make_page(i)[0] for i in range(3)] + [make_page(9)[0][:512, :768].copy()]
    res = pipe.predict(pages)
    assert len(res) == 4
    for r in res:
        assert r.det_result.ndim == 2 and r.det_result.shape[1] == 8
        assert len(r.ocr_result) == len(r.det_result)
        ys = [0.01 * b[::2].sum() / 4 + b[1::2].sum() / 4 for b in r.det_result]
        assert ys == sorted(ys)                                   # reading order (ocr_system_task.py:161)
    assert pipe.metric["text_recognition"]["total"] == sum(len(r.ocr_result) for r in res)


def test_missing_stages_fail_loudly():
    from pdf_table_amd.pipeline import OcrTablePipeline
    with pytest.raises(NotImplementedError):
        OcrTablePipeline(device=0, synthetic_seed=0, table_structure=True)
