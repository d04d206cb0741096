"""The C-ABI library: builds, loads, exports every symbol include/buzzdetect_hip.h declares, and its
host-only index arithmetic is bit-exact against the oracle (no device compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from buzzdetect_amd import _lib, build
from oracle import yamnet_oracle as O

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "buzzdetect_hip.h")


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"^BD_API[^;(]*?\b(bd_[a-z_0-9]+)\s*\(", text, flags=re.M)))


def test_header_and_binding_list_the_same_functions():
    assert declared_symbols() == sorted(_lib.PROTOTYPES)
    assert len(declared_symbols()) == 44


def test_library_exports_every_declared_symbol(lib):
    raw = C.CDLL(_lib.library_path())
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} not exported"
    assert lib.bd_abi_version() == 5


def test_header_constants_match_binding():
    text = open(HEADER).read()
    consts = {k: int(v) for k, v in re.findall(r"#define\s+(BD_[A-Z_]+)\s+\(?(-?\d+)\)?", text)}
    assert consts["BD_EMBEDDER_BLOB_FLOATS"] == _lib.EMBEDDER_BLOB_FLOATS == 3217344
    assert consts["BD_NUM_STAGES"] == _lib.NUM_STAGES and consts["BD_PROFILE_SLOTS"] == _lib.PROFILE_SLOTS
    assert consts["BD_MIN_SAMPLES"] == O.MIN_SAMPLES and consts["BD_PATCH_FRAMES"] == O.PATCH_FRAMES
    for code, name in _lib.ERROR_NAMES.items():
        assert consts[name] == code


# (5376, 34), (10752, 67), (14592, 91): framehop_prop 0.35 / 0.7 / 0.95, whose float64 product 0.96 * p * 16000 lies just
# below the integer that tf.cast's float32 tensor rounds to (features.py:99)
@pytest.mark.parametrize("hop,step", [(15360, 96), (7680, 48), (4608, 29), (16000, 100), (333, 2), (5376, 34), (10752, 67),
                                      (14592, 91)])
def test_index_arithmetic_bit_exact_vs_oracle(lib, hop, step):
    rng = np.random.default_rng(hop)
    ns = [0, 1, 399, 400, 15359, 15360, 15599, 15600, 15601, 23360, 3_194_880, 9_600_000, 15_728_640,
          (1 << 24) - 1] + rng.integers(0, 1 << 24, 200).tolist()
    for n in ns:
        assert lib.bd_padded_length(n, hop) == O.padded_length(n, hop)
        assert lib.bd_num_frames(n, hop) == O.num_frames(O.padded_length(n, hop))
        assert lib.bd_num_windows(n, hop, step) == O.num_windows(n, hop, step)


def test_index_arithmetic_errors(lib):
    assert lib.bd_num_windows(-1, 15360, 96) == -1                 # BD_EINVAL
    assert lib.bd_num_windows(100, 0, 96) == -1
    assert lib.bd_num_windows(100, 15360, 0) == -1
    assert lib.bd_num_windows(1 << 24, 15360, 96) == -5           # BD_ERANGE (hazard H2)
    assert b"float32" in lib.bd_last_error()


def test_stage_shapes(lib):
    h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
    expect = {0: (48, 32, 32), 1: (48, 32, 32), 2: (48, 32, 64), 3: (24, 16, 64), 4: (24, 16, 128),
              12: (6, 4, 512), 23: (3, 2, 512), 24: (3, 2, 1024), 26: (3, 2, 1024)}
    for s, shape in expect.items():
        assert lib.bd_stage_shape(s, C.byref(h), C.byref(w), C.byref(c)) == 0
        assert (h.value, w.value, c.value) == shape
    assert lib.bd_stage_shape(27, C.byref(h), C.byref(w), C.byref(c)) == -1


def test_create_rejects_bad_arguments_before_touching_a_device(lib):
    handle = C.c_void_p()
    w = _lib.bd_weights()
    assert lib.bd_create(C.byref(handle), 0, C.byref(w)) == -1          # null blob
    blob = np.zeros(10, np.float32)
    mel = np.zeros(257 * 64, np.float32)
    w.embedder_blob = blob.ctypes.data_as(C.POINTER(C.c_float))
    w.embedder_floats = blob.size
    w.mel = mel.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.bd_create(C.byref(handle), 0, C.byref(w)) == -6          # BD_EWEIGHTS: wrong blob size
    assert handle.value is None


def test_create_without_device_is_enodevice(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from buzzdetect_amd import weights as W
    blob = W.synthetic_embedder_blob()
    mel = W.load_mel()
    w = _lib.bd_weights()
    w.embedder_blob = blob.ctypes.data_as(C.POINTER(C.c_float))
    w.embedder_floats = blob.size
    w.mel = mel.ctypes.data_as(C.POINTER(C.c_float))
    handle = C.c_void_p()
    assert lib.bd_create(C.byref(handle), 0, C.byref(w)) == -2          # BD_ENODEVICE, no CPU fallback
    assert b"no CPU path" in lib.bd_last_error()


def test_round3_entry_points_reject_bad_arguments_without_a_device(lib):
    """bd_predict_chunks / bd_calibrate / bd_get_scales / bd_set_activation_exponents: argument errors are BD_EINVAL with a
    message, before any device call."""
    ptrs = (C.c_void_p * 1)(None)
    lens = (C.c_int64 * 1)(0)
    assert lib.bd_predict_chunks(None, ptrs, lens, 1, 15360, 96, None, 0, None, None, -1, None, None) == -1
    assert b"no output" in lib.bd_last_error()
    dummy = (C.c_float * 4)()
    assert lib.bd_predict_chunks(None, None, lens, 1, 15360, 96, None, 0, None, dummy, -1, None, None) == -1
    assert lib.bd_predict_chunks(None, ptrs, lens, 65, 15360, 96, None, 0, None, dummy, -1, None, None) == -1
    assert b"1..64" in lib.bd_last_error()
    assert lib.bd_predict_chunks(None, ptrs, lens, 1, 15360, 96, None, 0, None, dummy, -1, None, None) == -1
    assert b"null handle" in lib.bd_last_error()
    assert lib.bd_calibrate(None, None, 0, 15360, 96, None, 0, None) == -1
    assert lib.bd_get_scales(None, None, None) == -1
    exps = (C.c_int32 * 13)()
    assert lib.bd_set_activation_exponents(None, exps) == -1


@pytest.mark.gpu
def test_predict_chunks_argument_errors_on_device(engine):
    import torch
    lib_ = _lib.load()
    x = torch.zeros(15600, device=engine.device)
    out = torch.empty((1, engine.n_classes), device=engine.device)
    ws_bytes = _lib.check(lib_.bd_workspace_bytes(engine._handle, x.numel(), 15360, 96))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=engine.device)
    ptrs = (C.c_void_p * 1)(x.data_ptr())
    lens = (C.c_int64 * 1)(x.numel())
    s = torch.cuda.current_stream().cuda_stream

    def call(mode=-1, word=None, p=ptrs):
        return lib_.bd_predict_chunks(engine._handle, p, lens, 1, 15360, 96, ws.data_ptr(), ws.numel(), None, out.data_ptr(),
                                      mode, word, s)

    assert call() == 0
    assert call(mode=3) == -1 and b"mode" in lib_.bd_last_error()
    pageable = np.zeros(1, np.int32)                    # neither device memory nor pinned: the kernels could not write it
    assert call(word=pageable.ctypes.data) == -1 and b"range_word" in lib_.bd_last_error()
    odd = (C.c_void_p * 1)(x.data_ptr() + 2)
    assert call(p=odd) == -1 and b"4-byte" in lib_.bd_last_error()
    pinned = torch.zeros(4, dtype=torch.int32, pin_memory=True)
    assert call(word=pinned[2:].data_ptr()) == 0        # a word in the middle of a pinned block
    torch.cuda.synchronize()
    assert pinned.tolist() == [0, 0, 0, 0]
    bad = (C.c_int32 * 13)(*([0] * 12 + [61]))
    assert lib_.bd_set_activation_exponents(engine._handle, bad) == -1
    exps, _ = engine.scales()
    assert np.all(np.abs(exps) < 60)


def test_stager_rejects_bad_arguments_without_a_device(lib):
    """bd_stager_* (the streamer's way onto the device): argument errors before any device call; no device, no stager."""
    st = C.c_void_p()
    assert lib.bd_stager_create(None, 0, 1 << 20, 2) == -1
    assert lib.bd_stager_create(C.byref(st), 0, 1000, 2) == -1 and b"4096" in lib.bd_last_error()
    assert lib.bd_stager_create(C.byref(st), 0, 1 << 20, 5) == -1
    assert lib.bd_stager_read(None, 0, 0, 16, None, None) == -1
    assert lib.bd_stager_acquire(None, None, None) == -1 and lib.bd_stager_submit(None, 0, 0, None, None) == -1
    assert lib.bd_stager_destroy(None) == 0
    import torch
    if not torch.cuda.is_available():
        assert lib.bd_stager_create(C.byref(st), 0, 1 << 20, 2) == -2 and b"no CPU path" in lib.bd_last_error()


@pytest.mark.gpu
def test_stager_moves_a_file_to_the_device_piece_by_piece(lib, tmp_path):
    """bd_stager_read: file bytes -> device memory through two 64 KB page-locked buffers (dozens of pieces, a ragged last
    one, a read that runs past the end of the file); bd_stager_acquire / bd_stager_submit: the same by hand."""
    import torch
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 64 * 1024 * 37 + 12345, dtype=np.uint8)
    path = tmp_path / "blob.bin"
    path.write_bytes(b"HEAD" * 11 + data.tobytes())
    st = C.c_void_p()
    _lib.check(lib.bd_stager_create(C.byref(st), 0, 64 * 1024, 2))
    fd = os.open(path, os.O_RDONLY)
    try:
        stream = torch.cuda.Stream()
        dev = torch.zeros(data.size + 999, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        got = lib.bd_stager_read(st, fd, 44, data.size + 5000, dev.data_ptr(), stream.cuda_stream)
        assert got == data.size                                            # short at the end of the file
        stream.synchronize()
        assert np.array_equal(dev[: data.size].cpu().numpy(), data) and int(dev[data.size:].sum()) == 0
        assert lib.bd_stager_read(st, fd, 44 + data.size, 10, dev.data_ptr(), stream.cuda_stream) == 0
        # by hand: three pieces written by the caller
        dev.zero_()
        torch.cuda.synchronize()
        at = 0
        for n in (64 * 1024, 1000, 64 * 1024):
            index, host = C.c_int32(), C.c_void_p()
            _lib.check(lib.bd_stager_acquire(st, C.byref(index), C.byref(host)))
            buf = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), shape=(64 * 1024,))
            buf[:n] = data[at:at + n]
            _lib.check(lib.bd_stager_submit(st, index.value, n, dev.data_ptr() + at, stream.cuda_stream))
            at += n
        stream.synchronize()
        assert np.array_equal(dev[:at].cpu().numpy(), data[:at])
        assert lib.bd_stager_submit(st, 0, 64 * 1024 + 1, dev.data_ptr(), stream.cuda_stream) == -1
    finally:
        os.close(fd)
        assert lib.bd_stager_destroy(st) == 0
