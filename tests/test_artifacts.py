"""TensorBundle / SavedModel readers against files written by the test itself (the public formats)."""
import struct

import numpy as np
import pytest

from buzzdetect_amd import artifacts as A


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(no, wt, payload):
    key = _varint((no << 3) | wt)
    if wt == 0:
        return key + _varint(payload)
    if wt == 2:
        return key + _varint(len(payload)) + payload
    return key + payload


def _shape(dims):
    return b"".join(_field(2, 2, _field(1, 0, d)) for d in dims)


def _block(entries):
    out = bytearray()
    prev = b""
    for k, v in entries:
        shared = 0
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
            shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    out += struct.pack("<I", 0) + struct.pack("<I", 1)   # one restart at 0
    return bytes(out)


def write_bundle(tmp_path, tensors):
    data = bytearray()
    entries = [(b"", _field(1, 0, 1))]
    for name, arr in sorted(tensors.items()):
        entry = (_field(1, 0, 1) + _field(2, 2, _shape(arr.shape)) + _field(4, 0, len(data)) +
                 _field(5, 0, arr.nbytes) + _field(6, 5, struct.pack("<I", 0)))
        entries.append((name.encode(), entry))
        data += arr.astype("<f4").tobytes()
    blk = _block(entries)
    body = blk + b"\x00" + struct.pack("<I", 0)                      # block trailer (type + crc)
    meta_off = len(body)
    meta = _block([]) + b"\x00" + struct.pack("<I", 0)
    idx_off = meta_off + len(meta)
    index_blk = _block([(b"\xff", _varint(0) + _varint(len(blk)))])
    idx = index_blk + b"\x00" + struct.pack("<I", 0)
    footer = _varint(meta_off) + _varint(len(meta) - 5) + _varint(idx_off) + _varint(len(index_blk))
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    d = tmp_path / "variables"
    d.mkdir()
    (d / "variables.index").write_bytes(body + meta + idx + footer)
    (d / "variables.data-00000-of-00001").write_bytes(bytes(data))
    return str(tmp_path)


def test_bundle_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal((3, 3, 1, 32)).astype(np.float32),
               "layer_with_weights-1/beta/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal(32).astype(np.float32),
               "layer_with_weights-1/moving_mean/.ATTRIBUTES/VARIABLE_VALUE": rng.standard_normal(32).astype(np.float32)}
    idx_p, dat_p = A.bundle_paths(write_bundle(tmp_path, tensors))
    idx = A.read_bundle_index(idx_p)
    assert set(idx) == set(tensors)
    for name, arr in tensors.items():
        assert idx[name].shape == arr.shape
        assert np.array_equal(A.read_bundle_tensor(dat_p, idx[name]), arr)


def test_bad_magic_rejected(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 64)
    with pytest.raises(ValueError, match="bad table magic"):
        A.read_bundle_index(str(p))


def test_saved_model_const_extraction(tmp_path):
    mel = np.arange(257 * 64, dtype=np.float32).reshape(257, 64)
    tensor = _field(1, 0, 1) + _field(2, 2, _shape(mel.shape)) + _field(4, 2, mel.astype("<f4").tobytes())
    attr = _field(1, 2, b"value") + _field(2, 2, _field(8, 2, tensor))
    node = _field(1, 2, b"MatMul/b") + _field(2, 2, b"Const") + _field(5, 2, attr)
    scalar = _field(1, 0, 1) + _field(2, 2, b"") + _field(5, 5, struct.pack("<f", 0.001))
    node2 = (_field(1, 2, b"add/y") + _field(2, 2, b"Const") +
             _field(5, 2, _field(1, 2, b"value") + _field(2, 2, _field(8, 2, scalar))))
    fdef = _field(1, 2, _field(1, 2, b"__inference_fn")) + _field(3, 2, node) + _field(3, 2, node2)
    graph = _field(2, 2, _field(1, 2, fdef))
    saved = _field(2, 2, _field(2, 2, graph))
    p = tmp_path / "saved_model.pb"
    p.write_bytes(saved)
    assert np.array_equal(A.extract_mel_matrix(str(p)), mel)
    consts = {n: a for _, n, a in A.saved_model_constants(str(p))}
    assert consts["add/y"].shape == () and abs(float(consts["add/y"]) - 0.001) < 1e-9
