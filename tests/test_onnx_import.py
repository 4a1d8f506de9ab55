"""ONNX importer on the CPU: the dependency-free protobuf reader against files written by
PyTorch's own exporter, the engine layer list, and the round trip model.onnx -> reference-layout state_dict -> engine blob."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

from oracle import crnn as ocrnn                      # noqa: E402
from oracle import db_net, onnx_ref                   # noqa: E402
from pdf_table_amd import onnx_proto as P             # noqa: E402
from pdf_table_amd.onnx_import import UnsupportedOnnxGraph, load_onnx, recognise   # noqa: E402
from pdf_table_amd.synth_weights import crnn_state_dict, db_resnet18_state_dict    # noqa: E402
from pdf_table_amd.weights import pack_crnn, pack_db_resnet18                      # noqa: E402


@pytest.fixture(scope="module")
def db_sd():
    return db_resnet18_state_dict(seed=3)


@pytest.fixture(scope="module")
def db_export(db_sd):
    from onnx_export import export_db_resnet18
    return export_db_resnet18(db_sd)


def test_reader_executes_a_torch_export_like_the_module(db_sd, db_export):
    """bytes from torch's TorchScript ONNX serialiser -> parse_model -> oracle/onnx_ref.run == the exported network"""
    m = P.parse_model(db_export)
    assert m.producer == "pytorch" and m.opset == 13 and [i.name for i in m.inputs] == ["x"] and [o.name for o in m.outputs] == ["y"]
    assert m.inputs[0].shape == (1, 3, 64, 64)
    x = np.random.default_rng(0).standard_normal((1, 3, 64, 64)).astype(np.float32)
    (y,) = onnx_ref.run(m, {"x": x})
    with torch.no_grad():
        ref = db_net.db_forward_fp32(db_sd, torch.from_numpy(x)).numpy()
    assert y.shape == ref.shape == (1, 1, 64, 64) and np.abs(y - ref).max() <= 2e-6


def test_writer_reader_round_trip(db_export):
    m = P.parse_model(db_export)
    m2 = P.parse_model(P.serialize_model(m))
    assert [(n.op_type, n.inputs, n.outputs, n.name) for n in m.nodes] == [(n.op_type, n.inputs, n.outputs, n.name) for n in m2.nodes]
    for a, b in zip(m.nodes, m2.nodes):
        assert a.attrs.keys() == b.attrs.keys()
        for k in a.attrs:
            assert np.array_equal(np.asarray(a.attrs[k]), np.asarray(b.attrs[k])), (a.name, k)
    assert m.initializers.keys() == m2.initializers.keys()
    for k in m.initializers:
        assert m.initializers[k].dtype == m2.initializers[k].dtype and np.array_equal(m.initializers[k], m2.initializers[k])
    assert [(v.name, v.shape, v.elem_type) for v in m.inputs + m.outputs] == [(v.name, v.shape, v.elem_type) for v in m2.inputs + m2.outputs]
    with pytest.raises(ValueError):
        P.parse_model(db_export[:1000])                   # truncated file
    with pytest.raises(ValueError):
        P.parse_model(b"not a protobuf at all \xff\xff\xff\xff")


def test_db_resnet18_import(db_sd, db_export):
    """(a) the unfolded hand-written graph imports to the SAME engine blob as the state_dict; (b) torch's eval-mode export
    (BatchNorm folded into Conv in fp32 by the exporter) imports to a network equal to 2e-5 of the logit scale"""
    from onnx_export import write_db_resnet18
    g = load_onnx(write_db_resnet18(db_sd))
    assert g.unsupported_ops() == []
    arch, sd = recognise(g)
    assert arch == "db_resnet18"
    assert pack_db_resnet18(sd, x3=True) == pack_db_resnet18(db_sd, x3=True)
    g = load_onnx(db_export)
    h = g.op_histogram()
    assert h["Conv"] == 29 and h["ConvTranspose"] == 2 and h["Resize"] == 6 and h.get("BatchNormalization", 0) == 1
    layers = g.layers()
    assert sum(1 for l in layers if l.op == "conv") == 29 and not any(l.op in ("bn", "unsupported") for l in layers)
    assert [l.act for l in layers if l.op == "convT"] == ["relu", "sigmoid"] and [l.bn_folded for l in layers if l.op == "convT"] == [True, False]
    arch, sd = recognise(g)
    assert arch == "db_resnet18" and set(sd) >= {k for k in db_sd if not k.startswith("decoder.thresh")}
    x = torch.randn(1, 3, 96, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = db_net.db_forward_fp32(db_sd, x, return_logits=True)
        b = db_net.db_forward_fp32(sd, x, return_logits=True)
    assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())


def test_crnn_import_reorders_the_lstm_gates():
    """torch exports nn.LSTM as ONNX LSTM with gates (i, o, f, c) and W / R / B stacked per direction: the importer restores
    torch's (i, f, g, o) layout -- logits of the imported state_dict equal the original's, token ids identical"""
    from onnx_export import export_crnn
    csd = crnn_state_dict(seed=2)
    data = export_crnn(csd)
    m = P.parse_model(data)
    x = np.random.default_rng(5).uniform(0, 1, (1, 3, 32, 640)).astype(np.float32)
    (y,) = onnx_ref.run(m, {"x": x})                       # the reader + executor agree with the network ...
    with torch.no_grad():
        ref = ocrnn.crnn_forward_fp32(csd, torch.from_numpy(x), native_lstm=True).numpy()
    assert y.shape == ref.shape and np.abs(y - ref).max() <= 1e-4
    g = load_onnx(data)
    assert g.unsupported_ops() == [] and g.op_histogram()["LSTM"] == 2
    arch, sd = recognise(g)                                # ... and so does the state_dict the importer rebuilds
    assert arch == "crnn" and set(sd) == set(csd)
    with torch.no_grad():
        got = ocrnn.crnn_forward_fp32(sd, torch.from_numpy(x), native_lstm=True).numpy()
    assert np.abs(got - ref).max() <= 1e-4 and np.array_equal(got.argmax(-1), ref.argmax(-1))
    for k in csd:
        if "rnn" in k or k.startswith("cls"):
            assert torch.equal(sd[k], csd[k]), k              # LSTM / linear tensors come back bit for bit
    assert len(pack_crnn(sd, x3=False)) == len(pack_crnn(csd, x3=False))


def test_unknown_architecture_fails_loudly_with_an_inventory():
    """a graph of another architecture parses and is listed, but nothing pretends to run it"""
    from onnx_export import torch_export
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, 2, 1), torch.nn.BatchNorm2d(8), torch.nn.Hardswish(),
                              torch.nn.Conv2d(8, 8, 3, 1, 1, groups=8), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                              torch.nn.Linear(8, 4), torch.nn.Softmax(-1))
    g = load_onnx(torch_export(net, torch.zeros(1, 3, 32, 32)))
    layers = g.layers()
    # (the exporter already folded the BatchNorm into the conv, and wrote hardswish as x * HardSigmoid(x): opset 13 has no HardSwish)
    assert layers[0].op == "conv" and layers[0].bias is not None and layers[0].act == "hardswish" and layers[0].attrs["strides"] == [2, 2]
    assert layers[1].op == "conv" and layers[1].attrs["group"] == 8
    assert any(l.op == "gap" for l in layers) and any(l.op == "gemm" and l.weight.shape == (4, 8) for l in layers)
    with pytest.raises(UnsupportedOnnxGraph) as e:
        recognise(g)
    assert "none of the architectures" in str(e.value) and "conv w[8, 3, 3, 3]" in str(e.value)


def test_executor_layer_kinds_of_exported_cnns():
    """the layer lists of PyTorch-exported CNNs (an LCNet-type classifier with SE blocks, an FPN-type detector) consist of the
    kinds pdf_table_amd.onnx_exec executes -- conv (incl. depthwise) with a fused activation, convT, maxpool, gap, add, mul,
    resize, concat, gemm, act, Flatten glue -- and carry what the executor reads (HardSigmoid's alpha / beta, Resize scales);
    the executor itself needs the GPU (tests/test_gpu_onnx_exec.py)"""
    import importlib.util
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "tools"))
    spec = importlib.util.spec_from_file_location("gpu_onnx_exec_models", os.path.join(here, "test_gpu_onnx_exec.py"))
    mods = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mods)
    from onnx_export import torch_export
    from pdf_table_amd.onnx_import import load_onnx
    import pdf_table_amd.onnx_exec as ex
    assert hasattr(ex, "HipGraphExecutor")
    known = {"conv", "convT", "maxpool", "gap", "add", "mul", "resize", "concat", "gemm", "act", "glue"}
    seen = set()
    for m, x in ((mods.LcNetLike(), torch.randn(1, 3, 64, 96)), (mods.FpnLike(), torch.randn(1, 3, 64, 96))):
        g = load_onnx(torch_export(mods._randomise(m, 1), x))
        assert g.unsupported_ops() == []
        for lay in g.layers():
            assert lay.op in known, lay.describe()
            seen.add(lay.op)
            if lay.act == "hardsigmoid":
                assert abs(lay.attrs["act_alpha"] - 1.0 / 6.0) < 1e-6 and lay.attrs["act_beta"] == 0.5
            if lay.op == "resize":
                assert lay.attrs["scale"] == [1.0, 1.0, 2.0, 2.0] and lay.attrs["mode"] == "nearest"
            if lay.op == "glue":
                assert lay.attrs["onnx_op"] == "Flatten"
    assert seen == known
