"""Downmix + resample stage (BASELINE config 5's 48 kHz stereo input; src/stream/worker.py:116-128).

The reference resamples with librosa's default soxr_hq.  libsoxr is absent here, so what is pinned is the filter CLASS
(oracle/resample_oracle.py restates libsoxr's published HQ parameters; the response tests below hold the design to them) and
device == oracle; bit-parity with soxr itself stays unpinned."""
import ctypes as C

import numpy as np
import pytest
import scipy.signal

from oracle import resample_oracle as RO

QUALITY_CODE = {"scipy": 0, "hq": 1}


@pytest.mark.parametrize("quality", ["hq", "scipy"])
@pytest.mark.parametrize("rate_in,n", [(48000, 4800), (32000, 3001), (44100, 2205), (16000, 500), (8000, 400), (96000, 3000)])
def test_oracle_is_the_polyphase_definition(rate_in, n, quality):
    """resample() (upfirdn) == resample_direct() (the written-out sum) == scipy.signal.resample_poly with the same taps."""
    rng = np.random.default_rng(rate_in)
    x = rng.standard_normal(n).astype(np.float32)
    up, down = RO.ratio(rate_in, 16000)
    h, _ = RO.taps(up, down, quality=quality)
    want = (scipy.signal.resample_poly(x.astype(np.float64), up, down, window=h / up) if quality == "hq" and up * down > 1
            else scipy.signal.resample_poly(x.astype(np.float64), up, down))
    got = RO.resample(x, rate_in, quality=quality)
    direct = RO.resample_direct(x, rate_in, quality=quality)
    assert got.shape == want.shape == direct.shape
    assert np.abs(got - want).max() < 1e-12
    assert np.abs(got - direct).max() < 1e-12


@pytest.mark.parametrize("rate_in", [48000, 32000, 44100, 96000, 22050, 8000])
def test_hq_filter_has_the_soxr_hq_response(rate_in):
    """libsoxr HQ: pass band to 0.9136 of the lower Nyquist, stop band from that Nyquist, >= 120 dB (20-bit) rejection.
    Asserted on the float32 taps the device runs: ripple <= 0.01 dB up to 0.9125 of the lower Nyquist (7.3 kHz for
    -> 16 kHz), <= -120 dB from the lower Nyquist on, unity gain at DC, linear phase (symmetric taps)."""
    up, down = RO.ratio(rate_in, 16000)
    fp, fs = RO.hq_band_edges()
    assert abs(fp - 0.91363) < 1e-4 and fs == 1.0
    h, half = RO.taps(up, down, np.float32, "hq")
    assert h.size == 2 * half + 1 and np.array_equal(h, h[::-1])
    hd = h.astype(np.float64) / up
    assert abs(hd.sum() - 1.0) < 1e-6
    low_nyquist = min(rate_in, 16000) / 2.0
    w, resp = scipy.signal.freqz(hd, worN=1 << 19, fs=rate_in * up)
    mag = np.abs(resp)
    ripple_db = 20 * np.log10(mag[w <= 0.9125 * low_nyquist])
    assert np.abs(ripple_db).max() <= 0.01
    assert 20 * np.log10(mag[w >= low_nyquist].max()) <= -120.0
    # and the filter it replaces is NOT of that class: -30 dB at 9 kHz, -1.8 dB at 7.5 kHz for 48 -> 16
    if rate_in == 48000:
        h0, _ = RO.taps(up, down, np.float64, "scipy")
        w0, r0 = scipy.signal.freqz(h0, worN=1 << 16, fs=48000)
        assert 20 * np.log10(np.abs(r0[w0 >= 8000]).max()) > -10.0
        assert 20 * np.log10(np.abs(r0[np.argmin(np.abs(w0 - 7500))])) < -1.5


@pytest.mark.parametrize("quality", ["hq", "scipy"])
def test_library_designs_the_same_filter(quality):
    from buzzdetect_amd import _lib
    lib = _lib.load()
    q = QUALITY_CODE[quality]
    for rate_in in (48000, 32000, 44100, 8000, 22050, 96000):
        up, down, half = C.c_int32(), C.c_int32(), C.c_int32()
        n = lib.bd_resample_taps(rate_in, 16000, q, None, 0, C.byref(up), C.byref(down), C.byref(half))
        buf = np.zeros(n, np.float32)
        assert lib.bd_resample_taps(rate_in, 16000, q, buf.ctypes.data, n, C.byref(up), C.byref(down), C.byref(half)) == n
        h, hl = RO.taps(*RO.ratio(rate_in, 16000), quality=quality)
        assert (up.value, down.value) == RO.ratio(rate_in, 16000) and half.value == hl and n == h.size
        assert np.abs(buf - h).max() < 1e-7 * max(1, up.value)
        for n_in in (0, 1, 160, 48000, 1234567):
            assert lib.bd_resample_length(n_in, rate_in, 16000) == -(-n_in * up.value // down.value)
    assert lib.bd_resample_taps(48000, 16000, 7, None, 0, C.byref(up), C.byref(down), C.byref(half)) < 0


def test_ratios_whose_filter_outgrows_the_staged_span_are_refused_not_overrun():
    """with the HQ filter (~189 max(up, down) taps) a ratio that does not fit the matrix kernel goes to
    resample_kernel, which stages one tile's input span in 8192 floats of LDS; beyond down / up ~ 43 not even a one-output
    tile fits and the kernel used to stage past the array.  Such ratios are refused (host-side predicate, no GPU needed);
    everything the tests and the bench use stays supported."""
    from buzzdetect_amd import _lib
    lib = _lib.load()
    for rate_in in (48000, 32000, 44100, 96000, 22050, 24000, 12000, 8000, 16000, 192000, 11025):
        assert lib.bd_resample_supported(rate_in, 16000, QUALITY_CODE["hq"]) == 1, rate_in
        assert lib.bd_resample_supported(rate_in, 16000, QUALITY_CODE["scipy"]) == 1, rate_in
    assert lib.bd_resample_supported(768000, 16000, QUALITY_CODE["hq"]) == 0        # 48 : 1, 18 000 taps: span 18 196 > 8192
    assert lib.bd_resample_supported(1024000, 16000, QUALITY_CODE["hq"]) == 0
    assert lib.bd_resample_supported(768000, 16000, QUALITY_CODE["scipy"]) == 1     # the 61-tap-class filter: 960 taps
    assert lib.bd_resample_supported(16000 * 4099, 16000, QUALITY_CODE["hq"]) == 0  # ratio does not reduce to <= 4096
    assert lib.bd_resample_supported(0, 16000, 1) < 0 and lib.bd_resample_supported(48000, 16000, 7) < 0


def _fir_plan(rate_in):
    from buzzdetect_amd import _lib
    lib = _lib.load()
    geo = np.zeros(14, np.int32)
    if lib.bd_debug_fir_plan(rate_in, 16000, geo.ctypes.data, None, 0, None, 0) != 1:
        return None
    up, down, P, D, NB, kq, mt, contiguous, RS, a_bytes, lds_bytes, half = (int(v) for v in geo[:12])
    boff = np.zeros(NB, np.int32)
    g = np.zeros(NB * 4 * kq * 2 * 64 * 8, np.uint16)
    assert lib.bd_debug_fir_plan(rate_in, 16000, geo.ctypes.data, boff.ctypes.data, NB, g.ctypes.data, g.size) == 1
    return dict(up=up, down=down, P=P, D=D, NB=NB, kq=kq, mt=mt, contiguous=contiguous, RS=RS, a_bytes=a_bytes,
                lds_bytes=lds_bytes, half=half, boff=boff, g=g.view(np.float16).reshape(NB, 4 * kq, 2, 64, 8),
                unscale=geo[12:14].view(np.float32).copy())


@pytest.mark.parametrize("rate_in,n", [(48000, 30000), (32000, 20000), (44100, 30000), (96000, 40000), (8000, 3000),
                                       (22050, 9000), (24000, 9999), (12000, 5000)])
def test_matrix_core_plan_multiplies_out_to_the_oracle(rate_in, n):
    """The Toeplitz product fir_mfma_kernel computes - y[m][b][n] = sum_e x[m D + boff[b] + e] G_b[e][n], G rebuilt from
    the (hi, lo) f16 B fragments in v_mfma_f32_32x32x16_f16 lane order - evaluated in float64 on the CPU: geometry
    (period, phase blocks, offsets, band length), the power-of-two scales and the 22-bit filter halves, with no GPU."""
    p = _fir_plan(rate_in)
    assert p is not None
    assert p["P"] * p["down"] == p["D"] * p["up"] and p["P"] == 32 * p["NB"] and p["D"] % 8 == 0 and p["RS"] % 2 == 1
    assert (p["boff"] % 8 == 0).all() and p["lds_bytes"] <= 160 * 1024
    K = 16 * 4 * p["kq"]
    gf = p["g"].astype(np.float64)
    G = np.zeros((p["NB"], K, 32))
    for ks in range(4 * p["kq"]):
        for lane in range(64):
            G[:, 16 * ks + 8 * (lane >> 5):16 * ks + 8 * (lane >> 5) + 8, lane & 31] = gf[:, ks, 0, lane, :] + gf[:, ks, 1, lane, :]
    rng = np.random.default_rng(rate_in)
    q = rng.integers(-32768, 32767, n).astype(np.int16)
    x = q.astype(np.float32) / 32768.0
    n_out = -(-n * p["up"] // p["down"])
    periods = -(-n_out // p["P"])
    pad = 1 << 17
    xs = np.zeros(periods * p["D"] + K + 2 * pad)
    xs[pad:pad + n] = q.astype(np.float64)                       # 16-bit PCM is staged as the integer (times 2^15)
    y = np.zeros(periods * p["P"])
    for m in range(periods):
        for b in range(p["NB"]):
            s0 = pad + m * p["D"] + int(p["boff"][b])
            y[(m * p["NB"] + b) * 32:(m * p["NB"] + b) * 32 + 32] = xs[s0:s0 + K] @ G[b]
    got = y[:n_out] * float(p["unscale"][0])
    want = RO.resample(x, rate_in)
    assert np.abs(got - want).max() < 2e-7


def test_ratios_beyond_four_waves_registers_stay_on_the_vector_kernel():
    assert _fir_plan(192000) is None         # 2271 taps: 166 k-steps
    assert _fir_plan(16000) is None          # equal rates: no filter


@pytest.fixture()
def engine_q(engine, request):
    """The session engine at the resampler quality the test asks for; back to the default ("hq") afterwards."""
    engine.set_resample_quality(request.param)
    yield engine, request.param
    engine.set_resample_quality("hq")


@pytest.mark.gpu
@pytest.mark.parametrize("engine_q", ["hq", "scipy"], indirect=True)
@pytest.mark.parametrize("rate_in,channels,n", [(48000, 2, 48000), (48000, 1, 7001), (32000, 1, 61144 // 2),
                                                 (44100, 2, 22050), (16000, 2, 9999), (48000, 3, 3000), (96000, 2, 30011),
                                                 (8000, 1, 4000), (22050, 1, 11025), (24000, 2, 5000), (12000, 1, 3001),
                                                 (192000, 1, 40000), (11025, 1, 2000)])
def test_device_resample_matches_restatement(engine_q, rate_in, channels, n):
    """float32 PCM, every kernel of the stage: fir_mfma_kernel with one phase block (48 / 32 / 96 / 8 / 24 / 12 kHz) and with
    many (44.1 / 22.05 / 11.025 kHz), the vector kernel (192 kHz at "hq"; everything but 2:1 and 3:1 at "scipy"),
    decimate_kernel ("scipy" 2:1, 3:1), convert_kernel (16 kHz)."""
    engine, quality = engine_q
    rng = np.random.default_rng(n)
    x = (0.5 * rng.standard_normal((n, channels))).astype(np.float32)
    if channels == 1:
        x = x[:, 0]
    got = engine.resample(x, rate_in).cpu().numpy()
    want = RO.resample(x, rate_in, quality=quality)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < (5e-6 if quality == "hq" else 2e-6)


@pytest.mark.gpu
def test_hq_and_scipy_filters_differ_on_the_device(engine):
    """The quality switch is live: 48 kHz noise through the two filters differs at the 1e-2 level."""
    rng = np.random.default_rng(1)
    x = (0.3 * rng.standard_normal(48000)).astype(np.float32)
    hq = engine.resample(x, 48000).cpu().numpy()
    engine.set_resample_quality("scipy")
    try:
        old = engine.resample(x, 48000).cpu().numpy()
    finally:
        engine.set_resample_quality("hq")
    assert np.abs(hq - old).max() > 1e-2
    with pytest.raises(ValueError):
        engine.set_resample_quality("best")


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1.0, 1e-2, 1e-4, 3e-6])
def test_float_pcm_keeps_its_relative_accuracy_when_quiet(engine, level):
    """Float PCM is staged as hi + lo f16 halves of x * 2^6: 22 bits while lo is a normal f16 (|x| >= 2^-9), below that an
    absolute quantum of 2^-24 / 2^6 = 2^-30 per sample, i.e. an error floor of ~1e-9 (a recording 80 dB below full scale
    still has 1e-5 relative accuracy; an unscaled split would floor at 2^-25 = 3e-8)."""
    rng = np.random.default_rng(7)
    x = (level * 0.5 * rng.standard_normal(48000)).astype(np.float32)
    got = engine.resample(x, 48000).cpu().numpy()
    want = RO.resample(x, 48000)
    assert np.abs(got - want).max() < 5e-6 * level + 1e-9


@pytest.mark.gpu
def test_float_pcm_beyond_the_staging_range_saturates_not_nans(engine):
    """|x| > 1023 (float WAVs may hold anything) saturates at the f16 limit of the staged value; nothing becomes inf / nan,
    and the samples around it are untouched by it beyond the filter's reach."""
    x = np.zeros(48000, np.float32)
    x[24000] = 5e4
    x[100] = 0.25
    got = engine.resample(x, 48000).cpu().numpy()
    assert np.isfinite(got).all()
    want = RO.resample(np.where(np.abs(x) > 1023, np.sign(x) * 65504 / 64, x).astype(np.float32), 48000)
    assert np.abs(got - want).max() < 1e-2 and np.abs(got[:7000] - want[:7000]).max() < 5e-6


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
    assert np.abs(got - want).max() < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("engine_q", ["hq", "scipy"], indirect=True)
@pytest.mark.parametrize("rate_in", [48000, 32000])
@pytest.mark.parametrize("n", [1, 2, 7, 59, 61, 62, 200, 5373, 5374, 5377, 1792 * 3, 1792 * 3 * 2 + 1, 4096 * 3 - 1, 4096 * 3,
                               4096 * 3 + 1, 4096 * 2 * 3 + 95, 100003])
def test_integer_decimation_edges(engine_q, rate_in, n):
    """Inputs shorter than the filter, lengths around a workgroup's outputs (fir_mfma_kernel: 4096; decimate_kernel: 1792),
    a long odd length; mono float and stereo 16-bit."""
    engine, quality = engine_q
    tol = 5e-6 if quality == "hq" else 2e-6
    rng = np.random.default_rng(n + rate_in)
    x = (0.5 * rng.standard_normal(n)).astype(np.float32)
    got = engine.resample(x, rate_in).cpu().numpy()
    want = RO.resample(x, rate_in, quality=quality)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < tol
    q = rng.integers(-32768, 32767, size=(n, 2), dtype=np.int16)
    got16 = engine.resample(q, rate_in).cpu().numpy()
    want16 = RO.resample(q.astype(np.float32) / 32768.0, rate_in, quality=quality)
    assert got16.shape == want16.shape
    assert np.abs(got16 - want16).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,channels", [(48000, 2), (48000, 1), (32000, 2), (44100, 2), (96000, 1)])
def test_s16_pcm_is_exact_in_the_split(engine, rate_in, channels):
    """16-bit PCM (and the half-integer mean of two channels) is hi + lo with no remainder, so full-scale square waves and
    noise come out at the filter's own 22-bit accuracy: within 1e-6 of the oracle."""
    rng = np.random.default_rng(rate_in)
    n = 30000
    q = rng.integers(-32768, 32767, size=(n, channels), dtype=np.int16)
    q[1000:3000] = 32767
    q[3000:5000] = -32768
    if channels == 1:
        q = q[:, 0]
    got = engine.resample(q, rate_in).cpu().numpy()
    want = RO.resample(q.astype(np.float32) / 32768.0, rate_in)
    assert np.abs(got - want).max() < 1.5e-6


@pytest.mark.gpu
def test_config5_chain_s16_stereo_48k_in_both_f16_modes(weights_bundle):
    """BASELINE config 5 as a test: 48 kHz stereo 16-bit PCM -> fir_mfma_kernel (channel mean + 3:1 soxr_hq-class low-pass on
    the matrix cores) -> hot path, against the f64 oracle of the SAME chain (value / 32768, float32 channel mean, HQ filter,
    YAMNet, head).
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
        assert np.abs(mono.cpu().numpy() - mono_ref).max() < 5e-6
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


@pytest.mark.gpu
@pytest.mark.parametrize("rate_in,channels", [(48000, 2), (44100, 1)])
def test_an_hour_of_input_is_indexed_in_64_bits(engine, rate_in, channels):
    """One hour of 16-bit PCM in ONE call (172.8 M frames of 48 kHz stereo: 691 MB, 57.6 M outputs; sample indices beyond 2^27,
    byte offsets beyond 2^29): slices of the output - the start, the end, three places in between - against the oracle run
    on the matching piece of the input with the filter's reach on either side."""
    import torch
    n = rate_in * 3600
    gen = torch.Generator(device="cuda").manual_seed(rate_in)
    q = torch.randint(-20000, 20000, (n, channels), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
    if channels == 1:
        q = q[:, 0].contiguous()
    got = engine.resample(q, rate_in)
    up, down = RO.ratio(rate_in, 16000)
    _, half = RO.taps(up, down)
    n_out = -(-n * up // down)
    assert got.shape[0] == n_out
    reach = half // up + 2                                   # input samples either side of an output's centre
    block = 160 * 20                                         # outputs checked per place: whole periods of either ratio
    for j0 in (0, (n_out // 3) // block * block, (n_out // 2) // block * block, (5 * n_out // 6) // block * block, n_out - block):
        j1 = min(j0 + block, n_out)
        i0 = max(0, j0 * down // up - reach)
        i0 -= i0 % down                                      # keep the piece's phase: i0 a multiple of `down` = whole outputs
        i1 = min(n, j1 * down // up + reach + down)
        piece = q[i0:i1].cpu().numpy().astype(np.float32) / 32768.0
        ref = RO.resample(piece, rate_in)
        off = i0 * up // down                                # output index of the piece's first output
        a, b = j0 - off, j1 - off
        assert np.abs(got[j0:j1].cpu().numpy() - ref[a:b]).max() < 5e-6, (j0, j1)
    del q, got
    torch.cuda.empty_cache()
