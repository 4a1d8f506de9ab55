"""Dependency-free reader / writer for the subset of the ONNX protobuf schema an inference graph uses.

The reference runs its PP-OCR / PicoDet models through onnxruntime (``DeployUtils.prepare_onnx_model``,
utils/deploy_utils.py:243-280: ``onnx.load_model`` -> fp16 conversion -> ``ort.InferenceSession``).  Neither ``onnx`` nor
``onnxruntime`` exists on the build / GPU boxes, and the engine does not need them: an ``.onnx`` file is a protobuf
``ModelProto``; this module decodes the wire format (varint / fixed32 / fixed64 / length-delimited, packed repeated
scalars) for the messages below and encodes them again (the writer exists so that the importer can be validated against
graphs serialised from the in-tree state_dicts: tools/onnx_export.py).

Field numbers follow onnx/onnx.proto3 (public schema, ONNX IR version 8):
  ModelProto      1 ir_version, 2 producer_name, 7 graph, 8 opset_import{1 domain, 2 version}
  GraphProto      1 node, 2 name, 5 initializer, 11 input, 12 output
  NodeProto       1 input, 2 output, 3 name, 4 op_type, 5 attribute
  AttributeProto  1 name, 2 f, 3 i, 4 s, 5 t, 7 floats, 8 ints, 20 type
  TensorProto     1 dims, 2 data_type, 4 float_data, 5 int32_data, 7 int64_data, 8 name, 9 raw_data
  ValueInfoProto  1 name, 2 type{1 tensor_type{1 elem_type, 2 shape{1 dim{1 dim_value, 2 dim_param}}}}
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

__all__ = ["OnnxTensor", "OnnxNode", "OnnxValueInfo", "OnnxModel", "parse_model", "serialize_model", "DTYPES"]

# TensorProto.DataType
DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16,
          11: np.float64}
_DTYPE_ID = {np.dtype(v): k for k, v in DTYPES.items()}
# AttributeProto.AttributeType
_AT_FLOAT, _AT_INT, _AT_STRING, _AT_TENSOR, _AT_FLOATS, _AT_INTS = 1, 2, 3, 4, 6, 7


@dataclass
class OnnxNode:
    op_type: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    name: str = ""


@dataclass
class OnnxValueInfo:
    name: str
    elem_type: int = 1
    shape: Tuple[Union[int, str], ...] = ()


@dataclass
class OnnxModel:
    nodes: List[OnnxNode] = field(default_factory=list)
    initializers: Dict[str, np.ndarray] = field(default_factory=dict)
    inputs: List[OnnxValueInfo] = field(default_factory=list)
    outputs: List[OnnxValueInfo] = field(default_factory=list)
    opset: int = 13
    ir_version: int = 8
    producer: str = ""
    graph_name: str = "graph"


OnnxTensor = np.ndarray


# ---------------------------------------------------------------------------------------------------------------------
# wire format
# ---------------------------------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: bytes):
    """yield (field number, wire type, value) -- value: int for varint / fixed, memoryview for length-delimited"""
    pos, n = 0, len(buf)
    mv = memoryview(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = mv[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt} (field {fn})")
        yield fn, wt, v


def _sint64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v) -> List[int]:
    b = bytes(v)
    out, pos = [], 0
    while pos < len(b):
        x, pos = _varint(b, pos)
        out.append(_sint64(x))
    return out


def _tensor(buf) -> Tuple[str, np.ndarray]:
    buf = bytes(buf)
    dims: List[int] = []
    dtype = 1
    name = ""
    raw = None
    floats: List[float] = []
    ints: List[int] = []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
        elif fn == 2:
            dtype = v
        elif fn == 4:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]])
        elif fn in (5, 7):
            ints.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = bytes(v)
    if dtype not in DTYPES:
        raise ValueError(f"tensor '{name}': unsupported ONNX data type {dtype}")
    dt = np.dtype(DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dt.newbyteorder("<")).astype(dt)
    elif floats:
        arr = np.asarray(floats, dt)
    else:
        arr = np.asarray(ints, dt)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def _attribute(buf) -> Tuple[str, object]:
    name, atype = "", 0
    f = i = s = t = None
    floats: List[float] = []
    ints: List[int] = []
    for fn, wt, v in _fields(bytes(buf)):
        if fn == 1:
            name = bytes(v).decode()
        elif fn == 2:
            f = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fn == 3:
            i = _sint64(v)
        elif fn == 4:
            s = bytes(v)
        elif fn == 5:
            t = _tensor(v)[1]
        elif fn == 7:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]])
        elif fn == 8:
            ints.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
        elif fn == 20:
            atype = v
    if atype == _AT_FLOAT or (atype == 0 and f is not None):
        return name, float(f)
    if atype == _AT_INT or (atype == 0 and i is not None):
        return name, int(i)
    if atype == _AT_STRING or (atype == 0 and s is not None):
        return name, s.decode("utf-8", "replace")
    if atype == _AT_TENSOR or (atype == 0 and t is not None):
        return name, t
    if atype == _AT_FLOATS or (atype == 0 and floats):
        return name, [float(x) for x in floats]
    return name, [int(x) for x in ints]


def _node(buf) -> OnnxNode:
    n = OnnxNode("", [], [])
    for fn, wt, v in _fields(bytes(buf)):
        if fn == 1:
            n.inputs.append(bytes(v).decode())
        elif fn == 2:
            n.outputs.append(bytes(v).decode())
        elif fn == 3:
            n.name = bytes(v).decode()
        elif fn == 4:
            n.op_type = bytes(v).decode()
        elif fn == 5:
            k, a = _attribute(v)
            n.attrs[k] = a
    return n


def _value_info(buf) -> OnnxValueInfo:
    vi = OnnxValueInfo("")
    for fn, wt, v in _fields(bytes(buf)):
        if fn == 1:
            vi.name = bytes(v).decode()
        elif fn == 2:
            for fn2, _, v2 in _fields(bytes(v)):
                if fn2 != 1:
                    continue
                for fn3, _, v3 in _fields(bytes(v2)):
                    if fn3 == 1:
                        vi.elem_type = v3
                    elif fn3 == 2:
                        dims = []
                        for fn4, _, v4 in _fields(bytes(v3)):
                            if fn4 != 1:
                                continue
                            d: Union[int, str] = "?"
                            for fn5, wt5, v5 in _fields(bytes(v4)):
                                if fn5 == 1:
                                    d = _sint64(v5)
                                elif fn5 == 2:
                                    d = bytes(v5).decode()
                            dims.append(d)
                        vi.shape = tuple(dims)
    return vi


def parse_model(data: bytes) -> OnnxModel:
    """bytes of a ModelProto -> OnnxModel; raises ValueError on anything that is not one"""
    m = OnnxModel()
    graph = None
    try:
        for fn, wt, v in _fields(data):
            if fn == 1 and wt == 0:
                m.ir_version = v
            elif fn == 2 and wt == 2:
                m.producer = bytes(v).decode("utf-8", "replace")
            elif fn == 7 and wt == 2:
                graph = bytes(v)
            elif fn == 8 and wt == 2:
                dom, ver = "", None
                for fn2, _, v2 in _fields(bytes(v)):
                    if fn2 == 1:
                        dom = bytes(v2).decode()
                    elif fn2 == 2:
                        ver = v2
                if dom in ("", "ai.onnx") and ver is not None:
                    m.opset = ver
        if graph is None:
            raise ValueError("no GraphProto (field 7)")
        for fn, wt, v in _fields(graph):
            if fn == 1:
                m.nodes.append(_node(v))
            elif fn == 2:
                m.graph_name = bytes(v).decode()
            elif fn == 5:
                k, a = _tensor(v)
                m.initializers[k] = a
            elif fn == 11:
                m.inputs.append(_value_info(v))
            elif fn == 12:
                m.outputs.append(_value_info(v))
    except (IndexError, struct.error) as e:
        raise ValueError(f"not an ONNX ModelProto: truncated or malformed ({e})") from None
    m.inputs = [vi for vi in m.inputs if vi.name not in m.initializers]     # old exporters list initializers as inputs too
    return m


# ---------------------------------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------------------------------
def _enc_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(fn: int, wt: int) -> bytes:
    return _enc_varint((fn << 3) | wt)


def _ld(fn: int, payload: bytes) -> bytes:
    return _key(fn, 2) + _enc_varint(len(payload)) + payload


def _enc_tensor(name: str, arr: np.ndarray) -> bytes:
    arr = np.asarray(arr)
    dt = np.dtype(arr.dtype)
    if dt not in _DTYPE_ID:
        raise ValueError(f"tensor '{name}': dtype {dt} has no ONNX counterpart here")
    out = b"".join(_key(1, 0) + _enc_varint(int(d)) for d in arr.shape)
    out += _key(2, 0) + _enc_varint(_DTYPE_ID[dt])
    out += _ld(8, name.encode())
    out += _ld(9, np.ascontiguousarray(arr).astype(dt.newbyteorder("<")).tobytes())
    return out


def _enc_attr(name: str, v) -> bytes:
    out = _ld(1, name.encode())
    if isinstance(v, bool):
        v = int(v)
    if isinstance(v, float):
        out += _key(2, 5) + struct.pack("<f", v) + _key(20, 0) + _enc_varint(_AT_FLOAT)
    elif isinstance(v, int):
        out += _key(3, 0) + _enc_varint(v) + _key(20, 0) + _enc_varint(_AT_INT)
    elif isinstance(v, str):
        out += _ld(4, v.encode()) + _key(20, 0) + _enc_varint(_AT_STRING)
    elif isinstance(v, np.ndarray):
        out += _ld(5, _enc_tensor("", v)) + _key(20, 0) + _enc_varint(_AT_TENSOR)
    elif isinstance(v, (list, tuple)) and v and isinstance(v[0], float):
        out += _ld(7, b"".join(struct.pack("<f", x) for x in v)) + _key(20, 0) + _enc_varint(_AT_FLOATS)
    elif isinstance(v, (list, tuple)):
        out += _ld(8, b"".join(_enc_varint(int(x)) for x in v)) + _key(20, 0) + _enc_varint(_AT_INTS)
    else:
        raise TypeError(f"attribute '{name}': unsupported value {type(v)}")
    return out


def _enc_value_info(vi: OnnxValueInfo) -> bytes:
    dims = b""
    for d in vi.shape:
        dims += _ld(1, (_key(1, 0) + _enc_varint(d)) if isinstance(d, int) else _ld(2, str(d).encode()))
    tensor_type = _key(1, 0) + _enc_varint(vi.elem_type) + _ld(2, dims)
    return _ld(1, vi.name.encode()) + _ld(2, _ld(1, tensor_type))


def serialize_model(m: OnnxModel) -> bytes:
    g = b""
    for n in m.nodes:
        nb = b"".join(_ld(1, s.encode()) for s in n.inputs) + b"".join(_ld(2, s.encode()) for s in n.outputs)
        nb += _ld(3, n.name.encode()) + _ld(4, n.op_type.encode())
        nb += b"".join(_ld(5, _enc_attr(k, v)) for k, v in n.attrs.items())
        g += _ld(1, nb)
    g += _ld(2, m.graph_name.encode())
    for k, a in m.initializers.items():
        g += _ld(5, _enc_tensor(k, a))
    g += b"".join(_ld(11, _enc_value_info(vi)) for vi in m.inputs)
    g += b"".join(_ld(12, _enc_value_info(vi)) for vi in m.outputs)
    out = _key(1, 0) + _enc_varint(m.ir_version) + _ld(2, m.producer.encode()) + _ld(7, g)
    out += _ld(8, _ld(1, b"") + _key(2, 0) + _enc_varint(m.opset))
    return out
