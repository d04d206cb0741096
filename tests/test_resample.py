"""Downmix + resample stage (BASELINE config 5's 48 kHz stereo input)."""
import ctypes as C

import numpy as np
import pytest
import scipy.signal

from oracle import resample_oracle as RO


@pytest.mark.parametrize("rate_in,n", [(48000, 4800), (32000, 3001), (44100, 2205), (16000, 500), (8000, 400)])
def test_oracle_is_resample_poly(rate_in, n):
    rng = np.random.default_rng(rate_in)
    x = rng.standard_normal(n).astype(np.float32)
    up, down = RO.ratio(rate_in, 16000)
    want = scipy.signal.resample_poly(x.astype(np.float64), up, down)
    got = RO.resample(x, rate_in)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-12


def test_library_designs_the_same_filter():
    from buzzdetect_amd import _lib
    lib = _lib.load()
    for rate_in in (48000, 32000, 44100, 8000, 22050):
        up, down, half = C.c_int32(), C.c_int32(), C.c_int32()
        n = lib.bd_resample_taps(rate_in, 16000, None, 0, C.byref(up), C.byref(down), C.byref(half))
        buf = np.zeros(n, np.float32)
        assert lib.bd_resample_taps(rate_in, 16000, buf.ctypes.data, n, C.byref(up), C.byref(down), C.byref(half)) == n
        h, hl = RO.taps(*RO.ratio(rate_in, 16000))
        assert (up.value, down.value) == RO.ratio(rate_in, 16000) and half.value == hl and n == h.size
        assert np.abs(buf - h).max() < 1e-7
        for n_in in (0, 1, 160, 48000, 1234567):
            assert lib.bd_resample_length(n_in, rate_in, 16000) == -(-n_in * up.value // down.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,channels,n", [(48000, 2, 48000), (48000, 1, 7001), (32000, 1, 61144 // 2),
                                                 (44100, 2, 22050), (16000, 2, 9999), (48000, 3, 3000)])
def test_device_resample_matches_restatement(engine, rate_in, channels, n):
    rng = np.random.default_rng(n)
    x = (0.5 * rng.standard_normal((n, channels))).astype(np.float32)
    if channels == 1:
        x = x[:, 0]
    got = engine.resample(x, rate_in).cpu().numpy()
    want = RO.resample(x, rate_in)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6


@pytest.mark.gpu
def test_config5_input_48k_stereo_end_to_end(engine, weights_bundle):
    """48 kHz stereo -> device downmix/resample -> predict, against the CPU restatement of the same chain."""
    from oracle import yamnet_oracle as O
    t = np.arange(48000 * 3) / 48000.0
    rng = np.random.default_rng(5)
    left = 0.1 * rng.standard_normal(t.size) + 0.3 * np.sin(2 * np.pi * 220 * t)
    right = 0.1 * rng.standard_normal(t.size)
    x = np.stack([left, right], 1).astype(np.float32)
    mono = engine.resample(x, 48000)
    assert mono.shape[0] == 48000
    got = engine.predict(mono, 0.96).numpy()
    b = weights_bundle
    ref = O.predict(RO.resample(x, 48000).astype(np.float32), b["blob"], b["mel"], b["head_kernel"], b["head_bias"],
                    15360, 96, np.float64)
    assert got.shape == ref.shape == (4, 13)
    assert np.abs(got - ref).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,channels", [(48000, 2), (16000, 1), (32000, 1)])
def test_device_resample_from_s16(engine, rate_in, channels):
    """16-bit PCM in (half the PCIe bytes): same result as converting with value / 32768 on the host first."""
    rng = np.random.default_rng(rate_in + channels)
    q = rng.integers(-32768, 32767, size=(24000, channels), dtype=np.int16)
    if channels == 1:
        q = q[:, 0]
    got = engine.resample(q, rate_in).cpu().numpy()
    via_f32 = engine.resample(q.astype(np.float32) / 32768.0, rate_in).cpu().numpy()
    assert np.array_equal(got, via_f32)
    want = RO.resample(q.astype(np.float32) / 32768.0, rate_in)
    assert np.abs(got - want).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in", [48000, 32000])
@pytest.mark.parametrize("n", [1, 2, 7, 59, 61, 62, 200, 5373, 5374, 5377, 1792 * 3, 1792 * 3 * 2 + 1, 100003])
def test_integer_decimation_edges(engine, rate_in, n):
    """decimate_kernel (filter in scalar registers, seven outputs per thread): inputs shorter than the filter, lengths around
    a workgroup's 1792 outputs, a long odd length; mono float and stereo 16-bit."""
    rng = np.random.default_rng(n + rate_in)
    x = (0.5 * rng.standard_normal(n)).astype(np.float32)
    got = engine.resample(x, rate_in).cpu().numpy()
    want = RO.resample(x, rate_in)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6
    q = rng.integers(-32768, 32767, size=(n, 2), dtype=np.int16)
    got16 = engine.resample(q, rate_in).cpu().numpy()
    want16 = RO.resample(q.astype(np.float32) / 32768.0, rate_in)
    assert got16.shape == want16.shape
    assert np.abs(got16 - want16).max() < 2e-6


@pytest.mark.gpu
def test_config5_chain_s16_stereo_48k_in_both_f16_modes(weights_bundle):
    """BASELINE config 5 as a test: 48 kHz stereo 16-bit PCM -> decimate_kernel (channel mean + 3:1 polyphase) -> hot path,
    against the f64 oracle of the SAME chain (value / 32768, float32 channel mean, resample_poly filter, YAMNet, head).
    Split-f16 arithmetic (the default) stays inside the 1e-4 gate; plain f16 (config 5's arithmetic) is close but outside
    it by design: its error is reported, bounded on both sides, and never passed off as the gate."""
    from buzzdetect_amd.engine import HipEngine
    from oracle import yamnet_oracle as O
    n = 48000 * 12 + 777
    t = np.arange(n) / 48000.0
    rng = np.random.default_rng(55)
    left = 0.1 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 220 * t) * (np.mod(t, 5.0) < 0.5)
    right = 0.05 * rng.standard_normal(n) + 0.2 * np.sin(2 * np.pi * 3100 * t)
    q = (np.clip(np.stack([left, right], 1), -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16)
    b = weights_bundle
    mono_ref = RO.resample(q.astype(np.float32) / 32768.0, 48000)
    ref = O.predict(mono_ref.astype(np.float32), b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    eng = HipEngine()
    try:
        mono = eng.resample(q, 48000)
        assert np.abs(mono.cpu().numpy() - mono_ref).max() < 2e-6
        errs = {}
        for mode in ("f16x3", "f16", "f32"):
            eng.set_pointwise_mode(mode)
            got = eng.predict(eng.resample(q, 48000), 0.96).numpy()
            assert got.shape == ref.shape == (13, 13)
            errs[mode] = float(np.abs(got - ref).max())
        assert errs["f16x3"] < 1e-4 and errs["f32"] < 1e-4, errs
        assert 1e-5 < errs["f16"] < 5e-2, errs
        assert eng.overflow_reruns == 0
    finally:
        eng.close()
