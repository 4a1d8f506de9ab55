"""ONNX importer on the GPU: a model.onnx written by PyTorch's exporter runs on the HIP engine
through the same kernels as the checkpoint it was exported from."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

from oracle import db_net                                                     # noqa: E402
from pdf_table_amd import lib as L                                            # noqa: E402
from pdf_table_amd.synth_pages import make_page                               # noqa: E402
from pdf_table_amd.synth_weights import crnn_state_dict, db_resnet18_state_dict   # noqa: E402

pytestmark = pytest.mark.gpu


def test_onnx_session_runs_the_exported_detector(tmp_path):
    """HipOnnxSession.run(None, {"x": ...}) -- the reference's predictor.run surface (base_infer_task.py:366-370) -- on a
    torch-exported DB-ResNet18: BF16X3 within 1e-3 of the fp32 oracle of the ORIGINAL weights; the unfolded graph gives
    bit-identical maps to the engine loaded from the state_dict"""
    from onnx_export import export_db_resnet18, write_db_resnet18
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.onnx_import import HipOnnxSession
    from pdf_table_amd.weights import pack_db_resnet18
    sd = db_resnet18_state_dict(seed=0)
    p = tmp_path / "model.onnx"
    p.write_bytes(export_db_resnet18(sd))
    sess = HipOnnxSession(str(tmp_path))                     # a directory holding model.onnx, like prepare_onnx_model's onnx_dir
    assert sess.arch == "db_resnet18" and sess.get_inputs()[0].name == "x" and "HipExecutionProvider" in sess.get_providers()
    x = np.random.default_rng(1).standard_normal((2, 3, 96, 160)).astype(np.float32)
    with torch.no_grad():
        ref = db_net.db_forward_fp32(sd, torch.from_numpy(x)).numpy()
    sess.engine.set_precision(L.PT_PRECISION_BF16X3)
    (y,) = sess.run(None, {"x": x})
    assert y.shape == ref.shape == (2, 1, 96, 160) and y.dtype == np.float32
    d = float(np.abs(y - ref).max())
    print(f"onnx session (torch export, folded BN) x3: max|dprob| = {d:.2e}")
    assert d <= 1e-3
    (y16,) = sess.run(None, {"x": x.astype(np.float16)})     # fp16 feeds (build_onnx_infer_batch, base_infer_task.py:355-364) come back as fp16
    assert y16.dtype == np.float16
    sess.engine.set_precision(L.PT_PRECISION_BF16)
    (yb,) = sess.run(None, {"x": x})
    eng = HipEngine(0)
    eng.load_weights(L.PT_MODEL_DB_RESNET18, pack_db_resnet18(sd))
    sess2 = HipOnnxSession(write_db_resnet18(sd), engine=HipEngine(0))
    (yu,) = sess2.run(None, {"x": x})
    x4 = torch.zeros(2, 96, 160, 4)
    x4[..., :3] = torch.from_numpy(x).permute(0, 2, 3, 1)
    want = eng.det_forward_net(x4.to(torch.bfloat16).cuda()).cpu().numpy()[:, None]
    assert np.array_equal(yu, want)                           # unfolded graph == the state_dict, bit for bit
    assert np.abs(yb - want).max() <= 0.1                     # exporter-folded weights round to bf16 differently in ~1e-4 of the places: bf16-class drift (measured 0.036)


def test_tasks_load_onnx_checkpoints(tmp_path):
    """OcrDetectionTask(model='db_pp', task_path=<dir with model.onnx>) and OcrRecognitionTask(task_path=<dir with an exported
    CRNN>) serve the imported weights: same boxes / strings as the tasks built from the state_dicts"""
    from onnx_export import export_crnn, export_db_resnet18
    from pdf_table_amd.engine import HipEngine
    from pdf_table_amd.ocr_detection_task import OcrDetectionTask
    from pdf_table_amd.ocr_recognition_task import OcrRecognitionTask
    from pdf_table_amd.onnx_import import UnsupportedOnnxGraph
    eng = HipEngine(0)
    dsd = db_resnet18_state_dict(seed=0, text_signal=True)
    ddir, rdir = tmp_path / "det", tmp_path / "rec"
    ddir.mkdir()
    rdir.mkdir()
    (ddir / "model.onnx").write_bytes(export_db_resnet18(dsd))
    page = make_page(3)[0][:512, :640].copy()
    det = OcrDetectionTask(model="db_pp", task_path=str(ddir), engine=eng, thresh=0.3)
    boxes = det(page)[0]
    os.remove(ddir / "model.onnx")
    assert len(boxes) >= 5 and boxes.shape[1] == 8
    # the same weights as a .pt checkpoint through the db_pp pre/post (allow_stand_in: the route without an ONNX file)
    torch.save(dsd, str(ddir / "pytorch_model.pt"))
    ref_boxes = OcrDetectionTask(model="db_pp", task_path=str(ddir), engine=eng, thresh=0.3, allow_stand_in=True)(page)[0]
    assert abs(len(ref_boxes) - len(boxes)) <= 2            # exporter-folded BN rounds to bf16 a little differently
    csd = crnn_state_dict(seed=1)
    (rdir / "model.onnx").write_bytes(export_crnn(csd))
    rec = OcrRecognitionTask(model="CRNN", task_path=str(rdir), engine=eng)
    crops = [page[40:72, 30:400].copy(), page[200:240, 100:300].copy()]
    got = rec(crops)
    want = OcrRecognitionTask(model="CRNN", synthetic_seed=1, engine=eng)(crops)
    assert got == want and len(got) == 2
    with pytest.raises(UnsupportedOnnxGraph):               # a recogniser graph is not a detector
        OcrDetectionTask(model="db_pp", task_path=str(rdir), engine=eng)
