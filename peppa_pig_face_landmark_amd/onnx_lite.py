"""Dependency-free reader (and a minimal writer) of ONNX model files -- protobuf wire format only.

The reference loads its two networks from ``pretrained/yolov5n-0.5.onnx`` and ``pretrained/kps_student.onnx``
(Skps/config/Skps.yml:4,12 -> ``rt.InferenceSession(onnx_f)``, Skps/core/api/onnx_model_base.py:14).  Neither ``onnx``
nor ``onnxruntime`` (nor torch) is a dependency of this engine, so the weights are lifted out of the file here:
``read_model(path)`` returns the graph's nodes (op type, inputs, outputs, attributes) and its initializers as numpy
arrays.  Only what a Conv/BatchNormalization network export needs is decoded; unknown fields are skipped, as protobuf
prescribes.  ``write_model`` produces files the same reader (and onnx / onnxruntime) can load; the tests use it to build
synthetic exports.

Field numbers (onnx.proto3): ModelProto.graph = 7; GraphProto.node = 1, .name = 2, .initializer = 5, .input = 11,
.output = 12; NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5; AttributeProto.name = 1, .f = 2,
.i = 3, .s = 4, .t = 5, .floats = 7, .ints = 8, .type = 20; TensorProto.dims = 1, .data_type = 2, .float_data = 4,
.int32_data = 5, .int64_data = 7, .name = 8, .raw_data = 9, .double_data = 10.
"""
from __future__ import annotations

import struct
from typing import Dict, List, NamedTuple, Tuple

import numpy as np

_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


class Node(NamedTuple):
    op_type: str
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object]


class Model(NamedTuple):
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[str]
    outputs: List[str]


# ---- wire format -------------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: bytes):
    """Yield (field number, wire type, value) -- value is an int for varint / fixed types, a memoryview slice for
    length-delimited ones."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v) -> List[int]:
    out, pos, b = [], 0, bytes(v)
    while pos < len(b):
        x, pos = _varint(b, pos)
        out.append(_signed64(x))
    return out


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw = 1, "", None
    floats: List[float] = []
    ints: List[int] = []
    doubles: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v) if wt == 2 else [_signed64(v)]
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno in (5, 7):
            ints += _packed_varints(v) if wt == 2 else [_signed64(v)]
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            doubles += list(np.frombuffer(bytes(v), "<f8")) if wt == 2 else [struct.unpack("<d", struct.pack("<Q", v))[0]]
        elif fno == 13 or fno == 14:
            if fno == 14 and v == 1:
                raise ValueError("tensor %r keeps its data in an external file (not supported)" % name)
    if dtype not in _DT:
        raise ValueError("tensor %r has unsupported data_type %d" % (name, dtype))
    np_dt = np.dtype(_DT[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, np_dt.newbyteorder("<")).astype(np_dt)
    elif floats:
        arr = np.asarray(floats, np.float32).astype(np_dt)
    elif doubles:
        arr = np.asarray(doubles, np.float64).astype(np_dt)
    elif dtype == 10 and ints:          # float16 stored as uint16 bit patterns in int32_data
        arr = np.asarray(ints, np.uint16).view(np.float16)
    else:
        arr = np.asarray(ints, np.int64).astype(np_dt)
    count = int(np.prod(dims)) if dims else arr.size
    if arr.size != count:
        raise ValueError("tensor %r: %d values for dims %s" % (name, arr.size, dims))
    return name, arr.reshape(dims) if dims else arr.reshape(())


def _attribute(buf: bytes) -> Tuple[str, object]:
    name, val = "", None
    floats: List[float] = []
    ints: List[int] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(bytes(v))[1]
        elif fno == 7:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno == 8:
            ints += _packed_varints(v) if wt == 2 else [_signed64(v)]
    if floats:
        val = [float(x) for x in floats]
    elif ints:
        val = ints
    return name, val


def _node(buf: bytes) -> Node:
    ins: List[str] = []
    outs: List[str] = []
    name, op = "", ""
    attrs: Dict[str, object] = {}
    for fno, _, v in _fields(buf):
        if fno == 1:
            ins.append(bytes(v).decode())
        elif fno == 2:
            outs.append(bytes(v).decode())
        elif fno == 3:
            name = bytes(v).decode()
        elif fno == 4:
            op = bytes(v).decode()
        elif fno == 5:
            k, a = _attribute(bytes(v))
            attrs[k] = a
    return Node(op, name, ins, outs, attrs)


def _value_info_name(buf: bytes) -> str:
    for fno, _, v in _fields(buf):
        if fno == 1:
            return bytes(v).decode()
    return ""


def parse_model(data: bytes) -> Model:
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = bytes(v)
    if graph is None:
        raise ValueError("not an ONNX ModelProto: no graph field")
    nodes: List[Node] = []
    inits: Dict[str, np.ndarray] = {}
    inputs: List[str] = []
    outputs: List[str] = []
    for fno, wt, v in _fields(graph):
        if wt != 2:
            continue
        if fno == 1:
            nodes.append(_node(bytes(v)))
        elif fno == 5:
            name, arr = _tensor(bytes(v))
            inits[name] = arr
        elif fno == 11:
            inputs.append(_value_info_name(bytes(v)))
        elif fno == 12:
            outputs.append(_value_info_name(bytes(v)))
    for n in nodes:                      # Constant nodes are initializers in disguise
        if n.op_type == "Constant" and n.outputs and isinstance(n.attrs.get("value"), np.ndarray):
            inits[n.outputs[0]] = n.attrs["value"]
    return Model(nodes, inits, [i for i in inputs if i not in inits], outputs)


def read_model(path: str) -> Model:
    with open(path, "rb") as f:
        return parse_model(f.read())


# ---- minimal writer (synthetic exports for the tests; also loadable by onnx / onnxruntime) ------------------------------
def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fno: int, v: int) -> bytes:
    return _enc_varint(fno << 3) + _enc_varint(v)


def _enc_tensor(name: str, arr: np.ndarray) -> bytes:
    arr = np.ascontiguousarray(arr)
    code = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.float64): 11, np.dtype(np.int32): 6}[arr.dtype]
    out = b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, code) + _ld(8, name.encode())
    return out + _ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())


def _enc_attr(name: str, val) -> bytes:
    out = _ld(1, name.encode())
    if isinstance(val, float):
        return out + _enc_varint((2 << 3) | 5) + struct.pack("<f", val) + _vi(20, 1)
    if isinstance(val, int):
        return out + _vi(3, val) + _vi(20, 2)
    if isinstance(val, (bytes, str)):
        return out + _ld(4, val if isinstance(val, bytes) else val.encode()) + _vi(20, 3)
    if isinstance(val, np.ndarray):
        return out + _ld(5, _enc_tensor("", val)) + _vi(20, 4)
    vals = list(val)
    if vals and isinstance(vals[0], float):
        return out + _ld(7, struct.pack("<%df" % len(vals), *vals)) + _vi(20, 6)
    return out + _ld(8, b"".join(_enc_varint(int(v)) for v in vals)) + _vi(20, 7)


def write_model(path: str, nodes: List[Node], initializers: Dict[str, np.ndarray], inputs: List[str], outputs: List[str],
                opset: int = 12, producer: str = "peppa-hip onnx_lite"):
    def value_info(name: str) -> bytes:
        return _ld(1, name.encode())
    g = b""
    for n in nodes:
        body = b"".join(_ld(1, i.encode()) for i in n.inputs) + b"".join(_ld(2, o.encode()) for o in n.outputs)
        body += _ld(3, n.name.encode()) + _ld(4, n.op_type.encode())
        body += b"".join(_ld(5, _enc_attr(k, v)) for k, v in n.attrs.items())
        g += _ld(1, body)
    g += _ld(2, b"graph")
    for name, arr in initializers.items():
        g += _ld(5, _enc_tensor(name, arr))
    g += b"".join(_ld(11, value_info(i)) for i in inputs) + b"".join(_ld(12, value_info(o)) for o in outputs)
    model = _vi(1, 7) + _ld(2, producer.encode()) + _ld(7, g) + _ld(8, _ld(1, b"") + _vi(2, opset))
    with open(path, "wb") as f:
        f.write(model)
