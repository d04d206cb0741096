"""Readers for the TensorFlow artefacts the reference ships, without TensorFlow.

The reference keeps its weights in SavedModel directories (TensorBundle
``variables.index`` / ``variables.data-00000-of-00001`` plus ``saved_model.pb``).
The hot path needs three things out of them:

* the tensor table of the YAMNet embedder bundle (names, shapes, byte offsets) —
  ``embedders/yamnet_k2/models/yamnet_wholehop/variables/variables.index``;
* the dense head kernel/bias — ``models/model_general_v3/variables/*``;
* the ``[257, 64]`` mel filterbank that ``tf.signal.linear_to_mel_weight_matrix``
  (``embedders/yamnet/features.py:50-55``) left in the graph as a Const.

Formats (public TensorFlow / LevelDB formats, restated from their specifications):

* ``.index`` is a LevelDB *table*: 48-byte footer = two varint (offset, size)
  block handles + padding + 8-byte magic; every block is a run of
  prefix-compressed entries ``(shared, non_shared, value_len, key_delta, value)``
  followed by a restart array; the index block maps to data blocks.  Each value
  is a ``BundleEntryProto`` {1: dtype, 2: TensorShapeProto, 3: shard_id,
  4: offset, 5: size, 6: crc32c}; the entry with the empty key is the header.
* ``.data-*`` holds the raw little-endian tensors at those offsets.
* ``saved_model.pb`` is ``SavedModel{2: MetaGraphDef{2: GraphDef{1: NodeDef,
  2: FunctionDefLibrary{1: FunctionDef{3: NodeDef}}}}}``; a Const node keeps its
  payload in ``attr["value"].tensor{1: dtype, 2: shape, 4: tensor_content}``.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}


# --------------------------------------------------------------------------- #
# protobuf wire format
# --------------------------------------------------------------------------- #
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) for one message body."""
    pos = 0
    end = len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


def _shape(buf: bytes) -> Tuple[int, ...]:
    dims: List[int] = []
    for fno, _, val in _fields(buf):
        if fno == 2:  # TensorShapeProto.Dim
            size = 0
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


# --------------------------------------------------------------------------- #
# TensorBundle
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class BundleEntry:
    name: str
    dtype: int
    shape: Tuple[int, ...]
    shard: int
    offset: int
    size: int
    crc32c: int

    @property
    def count(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n


def _table_block(buf: bytes, offset: int, size: int) -> List[Tuple[bytes, bytes]]:
    block = buf[offset:offset + size]
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    out: List[Tuple[bytes, bytes]] = []
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_bundle_index(path: str) -> Dict[str, BundleEntry]:
    """Parse a TensorBundle ``variables.index`` into ``{tensor_name: BundleEntry}``."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorBundle index (bad table magic)")
    footer = buf[-48:]
    pos = 0
    _, pos = _varint(footer, pos)      # metaindex handle
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries: Dict[str, BundleEntry] = {}
    for _, handle in _table_block(buf, idx_off, idx_size):
        b_off, p = _varint(handle, 0)
        b_size, _ = _varint(handle, p)
        for key, val in _table_block(buf, b_off, b_size):
            if not key:  # BundleHeaderProto
                continue
            dtype = shard = offset = size = crc = 0
            shape: Tuple[int, ...] = ()
            for fno, wt, v in _fields(val):
                if fno == 1:
                    dtype = v
                elif fno == 2:
                    shape = _shape(v)
                elif fno == 3:
                    shard = v
                elif fno == 4:
                    offset = v
                elif fno == 5:
                    size = v
                elif fno == 6:
                    crc = struct.unpack("<I", v)[0]
            name = key.decode()
            entries[name] = BundleEntry(name, dtype, shape, shard, offset, size, crc)
    return entries


def read_bundle_tensor(data_path: str, entry: BundleEntry) -> np.ndarray:
    dt = _DTYPES[entry.dtype]
    with open(data_path, "rb") as f:
        f.seek(entry.offset)
        raw = f.read(entry.size)
    if len(raw) != entry.size:
        raise ValueError(f"{data_path}: short read for {entry.name}")
    return np.frombuffer(raw, dtype=np.dtype(dt).newbyteorder("<")).reshape(entry.shape).copy()


def bundle_paths(saved_model_dir: str) -> Tuple[str, str]:
    v = os.path.join(saved_model_dir, "variables")
    return os.path.join(v, "variables.index"), os.path.join(v, "variables.data-00000-of-00001")


# --------------------------------------------------------------------------- #
# SavedModel constants
# --------------------------------------------------------------------------- #
def _node_consts(node: bytes) -> Optional[Tuple[str, np.ndarray]]:
    name = ""
    op = ""
    tensor = None
    for fno, _, val in _fields(node):
        if fno == 1:
            name = val.decode()
        elif fno == 2:
            op = val.decode()
        elif fno == 5:  # map<string, AttrValue>
            k = None
            av = None
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    k = v2.decode()
                elif f2 == 2:
                    av = v2
            if k == "value" and av is not None:
                for f3, _, v3 in _fields(av):
                    if f3 == 8:  # AttrValue.tensor
                        tensor = v3
    if op != "Const" or tensor is None:
        return None
    dtype = 0
    shape: Tuple[int, ...] = ()
    content = b""
    floats: List[float] = []
    ints: List[int] = []
    for fno, wt, val in _fields(tensor):
        if fno == 1:
            dtype = val
        elif fno == 2:
            shape = _shape(val)
        elif fno == 4:
            content = val
        elif fno == 5:  # float_val (packed or not)
            if wt == 2:
                floats.extend(struct.unpack(f"<{len(val) // 4}f", val))
            else:
                floats.append(struct.unpack("<f", val)[0])
        elif fno == 7:  # int_val
            if wt == 2:
                p = 0
                while p < len(val):
                    x, p = _varint(val, p)
                    ints.append(x if x < (1 << 63) else x - (1 << 64))
            else:
                ints.append(val if val < (1 << 63) else val - (1 << 64))
        elif fno == 11:  # bool_val
            if wt == 2:
                ints.extend(int(b) for b in val)
            else:
                ints.append(int(val))
    if dtype not in _DTYPES:
        return None
    dt = np.dtype(_DTYPES[dtype])
    count = int(np.prod(shape)) if shape else 1
    if content:
        arr = np.frombuffer(content, dtype=dt.newbyteorder("<")).copy()
    elif floats:
        arr = np.asarray(floats, dtype=dt)
    elif ints:
        arr = np.asarray(ints, dtype=dt)
    else:
        arr = np.zeros(count, dtype=dt)
    if arr.size == 1 and count > 1:
        arr = np.full(count, arr[0], dtype=dt)
    if arr.size != count:
        return None
    return name, arr.reshape(shape)


def _walk_nodes(saved_model: bytes) -> Iterator[Tuple[str, bytes]]:
    """Yield (scope, NodeDef bytes) for the main graph and every library function."""
    for fno, _, mg in _fields(saved_model):
        if fno != 2:
            continue
        for f2, _, gd in _fields(mg):
            if f2 != 2:
                continue
            for f3, _, item in _fields(gd):
                if f3 == 1:
                    yield "", item
                elif f3 == 2:  # FunctionDefLibrary
                    for f4, _, fdef in _fields(item):
                        if f4 != 1:
                            continue
                        fname = ""
                        nodes = []
                        for f5, _, v5 in _fields(fdef):
                            if f5 == 1:  # OpDef signature
                                for f6, _, v6 in _fields(v5):
                                    if f6 == 1:
                                        fname = v6.decode()
                            elif f5 == 3:
                                nodes.append(v5)
                        for n in nodes:
                            yield fname, n


def saved_model_constants(path: str, min_elems: int = 1) -> List[Tuple[str, str, np.ndarray]]:
    """All numeric Const tensors in a ``saved_model.pb`` as (function, node, array)."""
    with open(path, "rb") as f:
        buf = f.read()
    out = []
    for scope, node in _walk_nodes(buf):
        got = _node_consts(node)
        if got is not None and got[1].size >= min_elems:
            out.append((scope, got[0], got[1]))
    return out


def extract_mel_matrix(saved_model_pb: str) -> np.ndarray:
    """The ``[257, 64]`` f32 mel weight matrix baked into a YAMNet SavedModel graph."""
    found = [a for _, _, a in saved_model_constants(saved_model_pb, min_elems=257 * 64)
             if a.shape == (257, 64) and a.dtype == np.float32]
    if not found:
        raise ValueError(f"{saved_model_pb}: no [257,64] float32 Const found")
    first = found[0]
    for other in found[1:]:
        if not np.array_equal(first, other):
            raise ValueError(f"{saved_model_pb}: conflicting [257,64] constants")
    return first


# --------------------------------------------------------------------------- #
# SavedModel graph structure (op types and attributes)
# --------------------------------------------------------------------------- #
def _attr_value(av: bytes):
    """AttrValue -> Python value for the kinds the YAMNet graphs use: s, i, f, b, type, shape, list(i / s / f / b)."""
    for fno, wt, val in _fields(av):
        if fno == 2:
            return val.decode(errors="replace")
        if fno == 3:
            return val if val < (1 << 63) else val - (1 << 64)
        if fno == 4:
            return struct.unpack("<f", val)[0]
        if fno == 5:
            return bool(val)
        if fno == 6:
            return {"dtype": int(val)}
        if fno == 7:
            return {"shape": list(_shape(val))}
        if fno == 8:
            return {"tensor": True}
        if fno == 10:
            return {"func": True}
        if fno == 1:                           # ListValue
            out: List[object] = []
            for f2, w2, v2 in _fields(val):
                if f2 == 2:
                    out.append(v2.decode(errors="replace"))
                elif f2 == 3:
                    if w2 == 2:
                        q = 0
                        while q < len(v2):
                            x, q = _varint(v2, q)
                            out.append(x if x < (1 << 63) else x - (1 << 64))
                    else:
                        out.append(v2 if v2 < (1 << 63) else v2 - (1 << 64))
                elif f2 == 4:
                    if w2 == 2:
                        out.extend(struct.unpack(f"<{len(v2) // 4}f", v2))
                    else:
                        out.append(struct.unpack("<f", v2)[0])
                elif f2 == 5:
                    if w2 == 2:
                        out.extend(bool(b) for b in v2)
                    else:
                        out.append(bool(v2))
            return out
    return None


@dataclass(frozen=True)
class GraphNode:
    function: str                  # "" = the main graph, else the library function that holds the node
    name: str
    op: str
    inputs: Tuple[str, ...]
    attrs: Dict[str, object]
    const: Optional[np.ndarray]    # payload of a numeric Const node


def saved_model_nodes(path: str) -> List[GraphNode]:
    """Every NodeDef of a ``saved_model.pb`` (main graph and function library) with its decoded attributes."""
    with open(path, "rb") as f:
        buf = f.read()
    out: List[GraphNode] = []
    for scope, node in _walk_nodes(buf):
        name = op = ""
        inputs: List[str] = []
        attrs: Dict[str, object] = {}
        for fno, _, val in _fields(node):
            if fno == 1:
                name = val.decode()
            elif fno == 2:
                op = val.decode()
            elif fno == 3:
                inputs.append(val.decode())
            elif fno == 5:
                k = None
                av = b""
                for f2, _, v2 in _fields(val):
                    if f2 == 1:
                        k = v2.decode()
                    elif f2 == 2:
                        av = v2
                if k is not None:
                    attrs[k] = _attr_value(av)
        const = None
        if op == "Const":
            got = _node_consts(node)
            const = got[1] if got is not None else None
        out.append(GraphNode(scope, name, op, tuple(inputs), attrs, const))
    return out
