"""Serialized ``tensorflow.TensorProto`` without TensorFlow: what ``tf.io.serialize_tensor`` writes and
``tf.io.parse_tensor`` reads (Pyesian/distributions/Sampled.py:34-60 stores every HMC sample that way, as
``samples/sample<i>.tf``).  Wire format (protobuf, tensorflow/core/framework/tensor.proto):

    field 1  dtype          varint   DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_INT64 = 9
    field 2  tensor_shape   message  TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
    field 4  tensor_content bytes    the elements, little endian, row major
  (parse also accepts the repeated-value forms: float_val = 5, double_val = 6, int_val = 7, int64_val = 10.)

Known answer (TensorFlow's TFRecord guide): the float32 tensor [[1, 2, 3], [4, 5, 6]] serializes to
``08 01 12 08 12 02 08 02 12 02 08 03 22 18`` followed by its 24 content bytes."""

from __future__ import annotations

import struct

import numpy as np

_DT = {"float32": 1, "float64": 2, "int32": 3, "int64": 9}
_NP = {v: np.dtype(k).newbyteorder("<") for k, v in _DT.items()}
_VAL_FIELD = {5: "<f4", 6: "<f8", 7: None, 10: None}   # repeated-value fields -> packed fixed format (None: varints)


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _len_delimited(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def serialize_tensor(a) -> bytes:
    a = np.asarray(a)
    name = a.dtype.name
    if name not in _DT:
        raise TypeError(f"unsupported dtype {name}")
    dims = b"".join(_len_delimited(2, _varint(1 << 3) + _varint(int(s))) for s in a.shape)
    content = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"))).tobytes()
    return _varint(1 << 3) + _varint(_DT[name]) + _len_delimited(2, dims) + _len_delimited(4, content)


def _parse_dim(sub: bytes) -> int:
    pos, size = 0, 0
    while pos < len(sub):
        key, pos = _read_varint(sub, pos)
        if key & 7 == 0:
            v, pos = _read_varint(sub, pos)
            if key >> 3 == 1:
                size = _signed(v)
        elif key & 7 == 2:
            n, pos = _read_varint(sub, pos)
            pos += n
        else:
            raise ValueError("unexpected wire type in TensorShapeProto.Dim")
    return size


def _parse_shape(buf: bytes):
    pos, dims = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        if key & 7 == 2:
            n, pos = _read_varint(buf, pos)
            if key >> 3 == 2:
                dims.append(_parse_dim(buf[pos:pos + n]))
            pos += n
        elif key & 7 == 0:
            _, pos = _read_varint(buf, pos)
        else:
            raise ValueError("unexpected wire type in TensorShapeProto")
    return tuple(dims)


def parse_tensor(buf: bytes, dtype=None) -> np.ndarray:
    """-> NumPy array; `dtype` (name), when given, must match the stored one (tf.io.parse_tensor's out_type)."""
    pos, dt, shape, content, vals = 0, None, (), None, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
            if field == 1:
                dt = v
            elif field in _VAL_FIELD:                    # unpacked repeated integer value
                vals.append(_signed(v))
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            sub = buf[pos:pos + n]
            pos += n
            if field == 2:
                shape = _parse_shape(sub)
            elif field == 4:
                content = sub
            elif field in _VAL_FIELD:
                if _VAL_FIELD[field]:                    # packed fixed-width values
                    vals.extend(np.frombuffer(sub, dtype=_VAL_FIELD[field]).tolist())
                else:                                    # packed varints
                    p2 = 0
                    while p2 < len(sub):
                        v, p2 = _read_varint(sub, p2)
                        vals.append(_signed(v))
        elif wire == 5:
            vals.append(struct.unpack_from("<f", buf, pos)[0])
            pos += 4
        elif wire == 1:
            vals.append(struct.unpack_from("<d", buf, pos)[0])
            pos += 8
        else:
            raise ValueError("unexpected wire type in TensorProto")
    if dt not in _NP:
        raise TypeError(f"unsupported TensorProto dtype {dt}")
    if dtype is not None and np.dtype(dtype).name != _NP[dt].name:
        raise TypeError(f"stored dtype {_NP[dt].name} does not match the requested {np.dtype(dtype).name}")
    count = int(np.prod(shape)) if shape else 1
    if content:
        arr = np.frombuffer(content, dtype=_NP[dt]).copy()
    else:                                               # value-list form: a single value fills the tensor (TF semantics)
        arr = np.asarray(vals if vals else [0], dtype=_NP[dt])
        if arr.size == 1 and count > 1:
            arr = np.repeat(arr, count)
    return arr.astype(_NP[dt].newbyteorder("=")).reshape(shape)
