"""HIP path vs the CPU oracle, through the C-ABI (run on the GPU box: ``pytest -m gpu``).

Tolerances (north_star: logits within 1e-4 of the CPU path, window indexing bit-exact):
  * logits / embeddings / CNN activations: max |HIP - f64 oracle| <= 1e-4 (absolute; |logit| ~ 10)
  * log-mel: <= 5e-5 absolute on noise-like audio; for pure tones an error model is used (see test)
  * window / frame counts, row order, determinism, group invariance: exact
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import yamnet_oracle as O

pytestmark = pytest.mark.gpu

TOL_LOGITS = 1e-4
HOP, STEP = 15360, 96


def oracle_logits(x, b, hop=HOP, step=STEP, dtype=np.float64):
    return O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], hop, step, dtype)


# --------------------------------------------------------------------------- front end
@pytest.mark.parametrize("n", [0, 1, 399, 400, 15360, 15600, 15601, 23360, 100_001])
def test_frontend_shapes_and_values_ragged_lengths(engine, weights_bundle, n):
    x = O.synthetic_audio(max(n, 1), seed=n)[:n]
    ref = O.log_mel(O.pad_waveform(x, HOP), weights_bundle["mel"], np.float64)
    got = engine.frontend(x if n else np.zeros(0, np.float32), HOP).cpu().numpy()
    assert got.shape == ref.shape == (O.num_frames(O.padded_length(n, HOP)), 64)
    assert np.abs(got - ref).max() < 5e-5


@pytest.mark.parametrize("n", [64 * 160 + 240, 64 * 160 + 241, 3 * 64 * 160 + 17, 300 * 64 * 160 + 5001])
def test_frontend_group_edges_and_persistent_loop(engine, weights_bundle, n):
    """The kernel walks groups of 64 frames with one workgroup per CU: lengths that end just before / after a group
    edge, and a chunk with more groups (300) than the GPU has CUs, so that workgroups loop and prefetch across groups.
    Also: the same input twice gives the same bits, and a chunk that starts at an odd sample offset of a larger buffer
    (how bd_predict_batch hands chunks over) gives the bits of a copy of it."""
    import torch
    x = O.synthetic_audio(n + 3, seed=n % 1000)
    ref = O.log_mel(O.pad_waveform(x[:n], HOP), weights_bundle["mel"], np.float64)
    got = engine.frontend(x[:n], HOP).cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got - ref).max() < 5e-5
    assert np.array_equal(engine.frontend(x[:n], HOP).cpu().numpy(), got)
    dev = torch.from_numpy(x).to(engine.device)
    from buzzdetect_amd import _lib
    out = torch.empty((ref.shape[0], 64), dtype=torch.float32, device=engine.device)
    lib = _lib.load()
    for shift in (0, 4, 8):           # 16-byte aligned views; the unaligned case goes through predict_batch below
        view = dev[shift: shift + n - 8]
        want = engine.frontend(view.clone(), HOP).cpu().numpy()
        o = out[: want.shape[0]]
        _lib.check(lib.bd_frontend(engine._handle, view.data_ptr(), view.numel(), HOP, o.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream))
        assert np.array_equal(o.cpu().numpy(), want), shift


def test_frontend_non_finite_neighbours_do_not_leak(engine, weights_bundle):
    """A frame reads 400 samples; the kernel's packed loads touch up to sample 415 of it.  Whatever lies there (the
    next frames' audio, here NaN and Inf) must not reach this frame: frames that end before the bad samples are clean."""
    x = O.synthetic_audio(HOP + 240, seed=3)
    bad = x.copy()
    bad[4000:4016] = np.nan
    bad[4016:4032] = np.inf
    clean = engine.frontend(x, HOP).cpu().numpy()
    got = engine.frontend(bad, HOP).cpu().numpy()
    last_clean = (4000 - 400) // 160                     # frame f reads samples [160 f, 160 f + 400)
    assert np.array_equal(got[: last_clean + 1], clean[: last_clean + 1])
    assert not np.isfinite(got[last_clean + 1]).all()


def test_frontend_silence_is_the_log_floor(engine):
    # log(0 + 0.001): one constant everywhere, within float32 libm accuracy (1-2 ulp) of ln(0.001f)
    got = engine.frontend(np.zeros(40000, np.float32), HOP).cpu().numpy()
    assert np.all(got == got[0, 0])
    assert abs(float(got[0, 0]) - np.log(np.float64(np.float32(0.001)))) < 1e-6


def test_frontend_full_scale_sine(engine, weights_bundle):
    # A pure tone leaves most bands ~60-100 dB below the peak; float32 FFT round-off (any float32
    # FFT, TensorFlow's included) is relative to the PEAK magnitude, so the error budget per band is
    # eps32 * c * peak / (mel + 0.001) after the log.  Checked in the mel domain instead.
    t = np.arange(48000) / 16000.0
    x = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    mel = weights_bundle["mel"]
    ref = O.log_mel(O.pad_waveform(x, HOP), mel, np.float64)
    got = engine.frontend(x, HOP).cpu().numpy().astype(np.float64)
    peak = np.exp(ref).max()
    err_mel = np.abs(np.exp(got) - np.exp(ref))
    assert err_mel.max() < 64 * np.finfo(np.float32).eps * peak
    loud = ref > np.log(peak) - np.log(1e3)          # bands within 60 dB of the peak
    assert np.abs(got - ref)[loud].max() < 5e-5
    f32 = O.log_mel(O.pad_waveform(x, HOP), mel, np.float32)
    assert np.abs(got - ref).max() <= 4 * max(np.abs(f32 - ref).max(), 1e-5)   # no worse than a CPU f32 FFT


def test_patches_are_views_of_logmel(engine, weights_bundle):
    x = O.synthetic_audio(60000, seed=3)
    lm = engine.frontend(x, 7680)
    p = engine.patches(lm, 48).cpu().numpy()
    assert np.array_equal(p, O.frame_patches(lm.cpu().numpy(), 48))
    assert p.shape == (O.num_windows(60000, 7680), 96, 64)


def test_frontend_matches_golden_rows(engine):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_oracle_f64.npz"))
    x = O.synthetic_audio(int(g["n_samples"]), seed=int(g["seed"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["audio_sha256"])
    got = engine.frontend(x, HOP).cpu().numpy()
    assert np.abs(got[g["logmel_row_index"]] - g["logmel_rows"]).max() < 5e-5


# --------------------------------------------------------------------------- CNN stages
def test_every_cnn_stage_against_oracle(engine_mode, weights_bundle):
    engine = engine_mode
    b = weights_bundle
    x = O.synthetic_audio(HOP * 3 + 500, seed=11)
    taps = []
    lm = O.log_mel(O.pad_waveform(x, HOP), b["mel"], np.float64)
    O.yamnet_body(O.frame_patches(lm, STEP), b["blob"], np.float64, taps)
    assert len(taps) == 27
    for stage, ref in enumerate(taps):
        got = engine.stage_tap(x, HOP, STEP, stage, ref.shape[0]).cpu().numpy()
        assert got.shape == ref.shape, stage
        assert np.abs(got - ref).max() < TOL_LOGITS, f"stage {stage}"


def test_pointwise_gemm_tile_edges(engine_mode, weights_bundle):
    engine = engine_mode
    # 1 window -> the GEMM M dimension (rows = positions) is not a multiple of the 128-row tile for the
    # deep layers (24 and 6 rows): exercises the bounds checks on loads and stores.
    b = weights_bundle
    x = O.synthetic_audio(9000, seed=12)
    taps = []
    lm = O.log_mel(O.pad_waveform(x, HOP), b["mel"], np.float64)
    O.yamnet_body(O.frame_patches(lm, STEP), b["blob"], np.float64, taps)
    for stage in (12, 14, 24, 26):
        got = engine.stage_tap(x, HOP, STEP, stage, 1).cpu().numpy()
        assert np.abs(got - taps[stage]).max() < TOL_LOGITS


# --------------------------------------------------------------------------- whole path
def test_logits_match_golden_fixture(engine_mode):
    engine = engine_mode
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_oracle_f64.npz"))
    x = O.synthetic_audio(int(g["n_samples"]), seed=int(g["seed"]))
    emb, logits = engine.run(x, HOP, STEP, True, True)
    assert np.abs(logits.cpu().numpy() - g["logits_whole"]).max() < TOL_LOGITS
    assert np.abs(emb.cpu().numpy() - g["embeddings_whole"]).max() < TOL_LOGITS
    _, half = engine.run(x, 7680, 48, False, True)
    assert half.shape == g["logits_half"].shape
    assert np.abs(half.cpu().numpy() - g["logits_half"]).max() < TOL_LOGITS


@pytest.mark.parametrize("hop,step,n", [(15360, 96, 15360 * 20 + 17), (7680, 48, 15360 * 9), (4608, 29, 70000)])
def test_logits_vs_oracle_various_hops(engine_mode, weights_bundle, hop, step, n):
    engine = engine_mode
    x = O.synthetic_audio(n, seed=hop)
    ref = oracle_logits(x, weights_bundle, hop, step)
    got = engine.run(x, hop, step, False, True)[1].cpu().numpy()
    assert got.shape == ref.shape == (O.num_windows(n, hop, step), 13)
    assert np.abs(got - ref).max() < TOL_LOGITS
    f32 = oracle_logits(x, weights_bundle, hop, step, np.float32)
    assert np.abs(f32 - ref).max() < TOL_LOGITS          # the CPU f32 path itself sits inside the gate


@pytest.mark.parametrize("n", [0, 1, 15600, 15601])
def test_edge_lengths_whole_path(engine, weights_bundle, n):
    x = O.synthetic_audio(max(n, 1), seed=77)[:n]
    ref = oracle_logits(x, weights_bundle)
    got = engine.predict(x, 0.96).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL_LOGITS


def test_silence_and_full_scale_inputs(engine, weights_bundle):
    # Silence must meet the plain gate.  Full-scale DC / square waves put almost all energy where the
    # mel matrix is zero: what is left in the bands is float32 FFT round-off sitting right at the
    # 0.001 log offset, so ANY float32 implementation (the CPU f32 oracle included) moves by far more
    # than 1e-4 there.  For those inputs HIP must be no worse than a small multiple of the CPU f32
    # path's own distance from the f64 value.
    for x in (np.zeros(50000, np.float32), np.full(50000, 1.0, np.float32),
              np.sign(np.sin(np.arange(50000) * 0.05)).astype(np.float32)):
        ref = oracle_logits(x, weights_bundle)
        got = engine.predict(x, 0.96).numpy()
        assert np.all(np.isfinite(got))
        if not x.any():
            assert np.abs(got - ref).max() < TOL_LOGITS
        else:
            cpu_f32 = np.abs(oracle_logits(x, weights_bundle, dtype=np.float32) - ref).max()
            assert np.abs(got - ref).max() < max(TOL_LOGITS, 8 * cpu_f32)


def test_chunk_edge_zero_padding_hazard_h1(engine, weights_bundle):
    # the last window of a chunk sees 240 zeros, not the next chunk's audio (features.py:92-107)
    x = O.synthetic_audio(HOP * 4, seed=31)
    whole = engine.predict(x, 0.96).numpy()
    first = engine.predict(x[: HOP * 2], 0.96).numpy()
    assert whole.shape == (4, 13) and first.shape == (2, 13)
    assert np.array_equal(whole[0], first[0])                    # interior window: identical bits
    assert not np.array_equal(whole[1], first[1])                # edge window: depends on chunking
    ref = oracle_logits(x[: HOP * 2], weights_bundle)
    assert np.abs(first - ref).max() < TOL_LOGITS


def test_too_long_chunk_is_refused(engine):
    import torch
    from buzzdetect_amd._lib import BuzzdetectHipError
    x = torch.zeros(1 << 24, dtype=torch.float32, device=engine.device)
    with pytest.raises(BuzzdetectHipError, match="BD_ERANGE"):
        engine.predict(x, 0.96)


# --------------------------------------------------------------------------- full-size properties
def test_full_batch_1024_properties(engine_mode, weights_bundle):
    """BASELINE config 2 batch (1024 windows = 15 728 640 samples): properties that need no oracle run
    at full size, plus oracle spot checks on a few windows."""
    import torch
    engine = engine_mode
    n = HOP * 1024
    x = O.synthetic_audio(n, seed=2024)
    xd = torch.from_numpy(x).to(engine.device)
    a = engine.predict(xd, 0.96).numpy()
    assert a.shape == (1024, 13) and np.all(np.isfinite(a))
    # determinism
    assert np.array_equal(a, engine.predict(xd, 0.96).numpy())
    # invariance to the pass size (windows per CNN pass)
    for group in (256, 100):
        engine.set_group_windows(group)
        assert np.array_equal(a, engine.predict(xd, 0.96).numpy()), group
    engine.set_group_windows(0)
    # window independence: row j depends only on samples [j*hop, j*hop + 15600)
    for j in (0, 1, 511, 1022):
        sub = engine.predict(xd[j * HOP: j * HOP + 15600 + HOP], 0.96).numpy()
        assert np.array_equal(sub[0], a[j]), j
    # time shift by one hop shifts rows by one
    shifted = engine.predict(xd[HOP:], 0.96).numpy()
    assert np.array_equal(shifted[:-1], a[1:-1])
    # oracle spot checks (f64) on scattered windows, incl. the zero-padded last one
    for j in (0, 700, 1023):
        seg = x[j * HOP: j * HOP + 15600]
        ref = oracle_logits(seg, weights_bundle)
        assert np.abs(a[j] - ref[0]).max() < TOL_LOGITS, j


def test_one_hour_file_in_batches_of_1024(engine, weights_bundle):
    """Config 2 end to end: 57.6 M samples as 3 x 1024 + 678 windows; counts and row order exact."""
    import torch
    from buzzdetect_amd import framing
    total = 57_600_000
    chunk = framing.round_chunklength(983.04)
    chunks = framing.gaps_to_chunklist([(0, total / 16000)], chunk)
    assert [framing.chunk_sample_range(c, 16000) for c in chunks][0] == (0, 15_728_640)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    audio = (0.1 * torch.randn(total, generator=gen)).clamp_(-1, 1)
    rows = []
    for c in chunks:
        a, b = framing.chunk_sample_range(c, 16000)
        rows.append(engine.predict(audio[a:b].to(engine.device), 0.96).numpy())
    assert [r.shape[0] for r in rows] == [1024, 1024, 1024, 678]
    assert sum(r.shape[0] for r in rows) == 3750
    starts = np.concatenate([framing.window_starts(r.shape[0], c[0], 0.96) for r, c in zip(rows, chunks)])
    assert np.all(np.diff(starts) > 0) and starts[-1] == 3599.04
    seg = audio[15_728_640: 15_728_640 + 15600].numpy()
    assert np.abs(rows[1][0] - oracle_logits(seg, weights_bundle)[0]).max() < TOL_LOGITS


# --------------------------------------------------------------------------- pointwise GEMM in isolation
def _pow2_operands(a, wt):
    """What bd_create / the calibration do to the operands of a 1x1 convolution (engine.hip, SepLayer in bd_internal.h),
    restated: activations times the power of two that puts their largest into [2^8, 2^9), every weight row times the one
    that puts its largest into [2^12, 2^13), f16 hi + lo of the scaled weights, and the epilogue's inverse factors."""
    import torch
    s = 9 - int(torch.frexp(a.abs().max())[1])
    e = 13 - torch.frexp(wt.abs().amax(dim=1))[1].to(torch.int32)
    ws = torch.ldexp(wt, e[:, None])
    whi = ws.to(torch.float16)
    wlo = (ws - whi.float()).to(torch.float16)
    unscale = torch.ldexp(torch.ones_like(e, dtype=torch.float32), -(e + s))
    return torch.ldexp(a, torch.tensor(s, device=a.device)), whi, wlo, unscale


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
@pytest.mark.parametrize("gain", [1.0, 30.0, 300.0, 2.0 ** 20, 0.01, 0.001, 2.0 ** -20])
@pytest.mark.parametrize("m,k,n", [(1, 32, 64), (6, 1024, 1024), (130, 64, 128), (4992, 512, 512), (1000, 256, 256), (19300, 128, 512)])
def test_pointwise_gemm_every_tile_variant(mode, m, k, n, gain):
    """Every tile shape of both GEMM kernels against an f64 product, on ragged M (tile-edge rows), at HOSTILE operand
    scales: activations times `gain`, weights divided by it, rows of the weight matrix a further factor 1..1000 apart
    (what BatchNorm folding does to real weights).  The bound is per output and relative to that output's own sum of
    |a||w| - the f32 kernel's own error class - so the split-f16 path cannot hide a lost low half behind the largest
    output of the matrix.  (19300 x 512: more tiles than the persistent exact-f32 kernel has workgroups - every workgroup
    walks several tiles, the last row tile ragged.)"""
    import torch
    from buzzdetect_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(m * 7 + k + n)
    a = torch.rand((m, k), generator=g, device=dev) * 8.0
    a[:, ::7] *= 1e-4                                  # small activations beside large ones
    a[:, 1::5] = 0.0                                   # ReLU zeros
    a *= gain
    wt = torch.randn((n, k), generator=g, device=dev) * (2.0 / k) ** 0.5 / gain
    wt *= torch.logspace(0, -3, n, device=dev)[torch.randperm(n, generator=g, device=dev)][:, None]
    bias = torch.randn(n, generator=g, device=dev) * 0.1
    ref = torch.relu(a.double() @ wt.double().T + bias.double())
    budget = a.double().abs() @ wt.double().abs().T + bias.double().abs()      # per output
    a16, whi, wlo, unscale = _pow2_operands(a, wt)
    stream = torch.cuda.current_stream().cuda_stream
    ran = 0
    for variant in range(0, 11):
        c = torch.full((m, n), -1.0, device=dev)
        if mode == "f32":
            rc = lib.bd_debug_pointwise(a.data_ptr(), wt.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, variant, stream)
        else:
            rc = lib.bd_debug_pointwise_f16x3(a16.data_ptr(), whi.data_ptr(), wlo.data_ptr(), unscale.data_ptr(),
                                              bias.data_ptr(), c.data_ptr(), m, n, k, variant, stream)
        if rc != 0:
            continue                                    # tile does not divide N
        ran += 1
        torch.cuda.synchronize()
        rel = float(((c.double() - ref).abs() / budget.clamp_min(1e-300)).max())
        assert rel <= 1.5e-6, (mode, variant, rel)
    assert ran >= 3


def test_fused_stem_is_bit_identical_to_unfused(engine, weights_bundle):
    """Layers 1-3 as one kernel (mode 3, the default; mode 5, the form of rounds 2-4) must reproduce conv1 -> depthwise ->
    pointwise -> depthwise -> pointwise launch by launch."""
    x = O.synthetic_audio(HOP * 37 + 1234, seed=55)
    engine.set_pointwise_mode("f16x3")
    try:
        engine.set_fusion(False, False)
        plain_pw2 = engine.stage_tap(x, HOP, STEP, 2, 38).cpu().numpy()
        plain_pw3 = engine.stage_tap(x, HOP, STEP, 4, 38).cpu().numpy()
        plain = engine.predict(x, 0.96).numpy()
        plain_half = engine.predict(x, 0.48).numpy()      # overlapping windows read shared log-mel rows
        for mode in (3, 5):                # 3 (default): the layer-2 tile handed over in registers (stemreg.hip), 5: a workgroup per row
                                           # block, the tile through LDS (stem3_kernel, the default until round 5); a tap inside
                                           # layers 1-3 runs them one kernel per op
            engine.set_fusion(mode, False)
            assert np.array_equal(engine.stage_tap(x, HOP, STEP, 2, 38).cpu().numpy(), plain_pw2), mode
            assert np.array_equal(engine.stage_tap(x, HOP, STEP, 4, 38).cpu().numpy(), plain_pw3), mode
            assert np.array_equal(engine.predict(x, 0.96).numpy(), plain), mode
            assert np.array_equal(engine.predict(x, 0.48).numpy(), plain_half), mode
        # the default stem against the block stem inside the default path, both f16 modes, more windows than resident runs
        y = O.synthetic_audio(HOP * 1050 + 15600, seed=56)      # (1051 windows: more than one pass, under 2^24 samples)
        for pw_mode in ("f16x3", "f16"):
            engine.set_pointwise_mode(pw_mode)
            engine.set_fusion(5, True)
            ref, ref_emb = engine.predict(y, 0.96).numpy(), engine.embed(y, 0.96).numpy()
            for alt in (3,):
                engine.set_fusion(alt, True)
                assert np.array_equal(engine.predict(y, 0.96).numpy(), ref), (pw_mode, alt)
                assert np.array_equal(engine.embed(y, 0.96).numpy(), ref_emb), (pw_mode, alt)
            for windows in (1, 2, 13, 300):                 # runs that start inside a window, single tiles, a partial last pass
                z = y[: HOP * (windows - 1) + 15600]
                engine.set_fusion(5, True)
                refz = engine.predict(z, 0.96).numpy()
                engine.set_fusion(3, True)
                assert np.array_equal(engine.predict(z, 0.96).numpy(), refz), (pw_mode, windows)
        engine.set_pointwise_mode("f16x3")
    finally:
        engine.set_fusion(True, True)


@pytest.mark.parametrize("windows", [1, 3, 37, 300])
def test_fused_separable_layers_bit_identical_to_unfused(engine, windows):
    """Depthwise-inside-GEMM (layers 4, 6, 8-12, 14) vs depthwise kernel + GEMM kernel: every pointwise
    output of those layers, and the logits, must match bit for bit — including partial last tiles."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=windows)
    engine.set_pointwise_mode("f16x3")
    try:
        for variant in (1, 10):                  # 1: the default launch set; 10: layers 5-7 on the four kernels of round 4
            engine.set_fusion(False, False)
            plain = {st: engine.stage_tap(x, HOP, STEP, st, windows).cpu().numpy() for st in (6, 10, 12, 14, 22, 24, 26)}
            plain_logits = engine.predict(x, 0.96).numpy()
            engine.set_fusion(False, variant)
            for st, ref in plain.items():
                got = engine.stage_tap(x, HOP, STEP, st, windows).cpu().numpy()
                assert np.array_equal(got, ref), (variant, st)
            assert np.array_equal(engine.predict(x, 0.96).numpy(), plain_logits)
    finally:
        engine.set_fusion(True, True)


@pytest.mark.parametrize("windows", [1, 4, 5, 131, 1025])
def test_layers_8_to_11_as_one_launch_bit_identical_to_a_launch_each(engine, windows):
    """Default path: every workgroup takes its four windows through layers 8-12 AND the stride-2 depthwise of layer 13 in ONE
    launch with the tiles between the layers kept on the CU (sepchip.hip: accumulators -> depthwise in registers -> LDS ring);
    hook 10: the same with layers 5-7 on their four kernels in front of it.  Against one kernel per op
    (bd_set_fusion 0, 0): the same bits, in both f16 modes, whole and partial tiles, first and last workgroup."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=800 + windows)
    try:
        for mode in ("f16x3", "f16"):
            engine.set_pointwise_mode(mode)
            engine.set_fusion(False, False)
            ref_logits = engine.predict(x, 0.96).numpy()
            ref_emb = engine.embed(x, 0.96).numpy()
            for hook in (10,):                   # 10: layers 5-7 on their four kernels instead of the on-chip launch (sepmid.hip)
                engine.set_fusion(True, hook)
                assert np.array_equal(engine.predict(x, 0.96).numpy(), ref_logits), (mode, hook)
                assert np.array_equal(engine.embed(x, 0.96).numpy(), ref_emb), (mode, hook)
            engine.set_fusion(True, True)
            for _ in range(2):                   # twice: the second pass reads buffers the first one left behind
                assert np.array_equal(engine.predict(x, 0.96).numpy(), ref_logits), mode
            assert np.array_equal(engine.embed(x, 0.96).numpy(), ref_emb), mode
    finally:
        engine.set_pointwise_mode("f16x3")
        engine.set_fusion(True, True)


@pytest.mark.parametrize("windows", [1, 7, 16, 17, 63, 64, 65, 678, 1024, 1031])
def test_tail_on_the_one_wave_per_simd_kernel_bit_identical_to_a_kernel_per_op(engine, windows):
    """Layers 13 / 14 + pool (septail.hip): pointwise 13 reads the depthwise-13 output as f16 hi / lo planes written by the on-chip
    run, applies depthwise 14 to its accumulators and writes planes again; pointwise 14 pools its accumulators.  Against one
    kernel per op: the same logits and embeddings bit for bit, both f16 modes and the exact-f32 mode; row tiles of 16 windows whole and partial, grids
    rounded up to four row tiles (workgroups that leave at once), a second pass of a 1024-window group, twice (the second
    call reads buffers the first left behind)."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=1300 + windows)
    try:
        for mode in ("f16x3", "f16", "f32"):     # f32: the same kernel on v_mfma_f32_32x32x2_f32, its A operand the f32 activation itself
            engine.set_pointwise_mode(mode)
            engine.set_fusion(False, False)
            ref_logits = engine.predict(x, 0.96).numpy()
            ref_emb = engine.embed(x, 0.96).numpy()
            engine.set_fusion(True, True)
            for _ in range(2):
                assert np.array_equal(engine.predict(x, 0.96).numpy(), ref_logits), mode
                assert np.array_equal(engine.embed(x, 0.96).numpy(), ref_emb), mode
            assert ref_emb.shape == (windows, 1024)
    finally:
        engine.set_pointwise_mode("f16x3")
        engine.set_fusion(True, True)


@pytest.mark.parametrize("windows", [1, 3, 37, 130])
def test_pointwise_on_wave_specialised_kernel_bit_identical_to_gemm_kernel(engine, windows):
    """The 1x1 convolutions of layers 5-14 run (unfused path) on the wave-specialised kernel with pass-through
    producers; the plain split-f16 GEMM kernel (tile variant 1) must give the same bits, partial tiles included."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=100 + windows)
    engine.set_pointwise_mode("f16x3")
    stages = (8, 10, 12, 16, 24, 26)
    try:
        engine.set_fusion(False, False)
        for layer in range(5, 15):
            engine.set_pointwise_variant(layer, 1)
        plain = {st: engine.stage_tap(x, HOP, STEP, st, windows).cpu().numpy() for st in stages}
        plain_logits = engine.predict(x, 0.96).numpy()
        for variant in (0, 10, 11):              # auto, the wave-specialised tile kernel, resident weights where they fit
            for layer in range(5, 15):
                engine.set_pointwise_variant(layer, variant)
            for st, ref in plain.items():
                assert np.array_equal(engine.stage_tap(x, HOP, STEP, st, windows).cpu().numpy(), ref), (variant, st)
            assert np.array_equal(engine.predict(x, 0.96).numpy(), plain_logits), variant
        for layer in range(5, 15):
            engine.set_pointwise_variant(layer, 0)
        plain_emb = engine.embed(x, 0.96).numpy()
        engine.set_fusion(True, True)            # default path: layers 5, 7, 13 use it after a depthwise, and the
        assert np.array_equal(engine.predict(x, 0.96).numpy(), plain_logits)    # last layer pools in its epilogue
        assert np.array_equal(engine.embed(x, 0.96).numpy(), plain_emb)
        for st, ref in plain.items():            # ... layers 4, 6, 12 apply the next layer's depthwise in theirs
            assert np.array_equal(engine.stage_tap(x, HOP, STEP, st, windows).cpu().numpy(), ref), ("default", st)
    finally:
        for layer in range(5, 15):
            engine.set_pointwise_variant(layer, 0)
        engine.set_fusion(True, True)


@pytest.mark.parametrize("hop_prop,expect", [(1.0, 625), (0.5, 1249)])
def test_config3_ten_minute_chunk(engine, weights_bundle, hop_prop, expect):
    """BASELINE config 3: a 600.0 s chunk (9 600 000 samples) of a 24 h recording, yamnet_k2 at whole and half hop:
    row count exact, rows independent of the rest of the chunk, oracle spot checks incl. the padded last row."""
    import torch
    from buzzdetect_amd import framing
    assert framing.round_chunklength(600) == 600.0
    n = 9_600_000
    gen = torch.Generator(device="cpu").manual_seed(33)
    x = (0.1 * torch.randn(n, generator=gen)).clamp_(-1, 1)
    xd = x.to(engine.device)
    got = engine.predict(xd, 0.96 * hop_prop).numpy()
    hop = int(15360 * hop_prop)
    assert got.shape == (expect, 13) == (O.num_windows(n, hop), 13)
    assert np.all(np.isfinite(got))
    for j in (0, expect // 2, expect - 1):
        seg = x[j * hop: j * hop + 15600].numpy()
        ref = oracle_logits(seg, weights_bundle)[0]
        assert np.abs(got[j] - ref).max() < TOL_LOGITS, j
    sub = engine.predict(xd[5 * hop: 5 * hop + 15600 + 3 * hop], 0.96 * hop_prop).numpy()
    assert np.array_equal(sub[:3], got[5:8])


@pytest.mark.parametrize("hop_prop", [1.0, 0.5])
def test_predict_batch_equals_per_chunk_calls(engine, hop_prop):
    """bd_predict_batch: windows of several chunks share every launch, each chunk keeps its own zero padding
    (hazard H1) -> rows identical, bit for bit, to one bd_predict per chunk.  Ragged lengths incl. tiny chunks."""
    lengths = [3_194_880, 15_600, 1, 0, 23_360, 100_001, 7, 3_194_879, 15_601, 640_000]
    chunks = [O.synthetic_audio(max(n, 1), seed=900 + i)[:n] for i, n in enumerate(lengths)]
    singles = [engine.predict(c, 0.96 * hop_prop).numpy() for c in chunks]
    batch, embs = engine.predict_batch(chunks, 0.96 * hop_prop, want_embeddings=True)
    assert len(batch) == len(chunks)
    hop = int(15360 * hop_prop)
    for n, one, many in zip(lengths, singles, batch):
        assert many.shape == one.shape == (O.num_windows(n, hop), 13)
        assert np.array_equal(many.numpy(), one)
    assert sum(len(e) for e in embs) == sum(s.shape[0] for s in singles)
    # different pass sizes must not matter either
    engine.set_group_windows(100)
    try:
        again = engine.predict_batch(chunks, 0.96 * hop_prop)
        for one, many in zip(singles, again):
            assert np.array_equal(many.numpy(), one)
    finally:
        engine.set_group_windows(0)


def test_predict_batch_argument_errors(engine):
    from buzzdetect_amd._lib import BuzzdetectHipError
    with pytest.raises(ValueError):
        engine.predict_batch([], 0.96)
    with pytest.raises(ValueError):
        engine.predict_batch([np.zeros(10, np.float32)] * 65, 0.96)
    import torch
    big = torch.zeros(1 << 24, dtype=torch.float32, device=engine.device)
    with pytest.raises(BuzzdetectHipError, match="BD_ERANGE"):
        engine.predict_batch([np.zeros(100, np.float32), big], 0.96)


def test_back_to_back_numpy_inputs_do_not_race(engine):
    """Host staging: many predict() calls on NumPy inputs enqueued without synchronising in between must each
    see their own samples (the pinned staging buffers are reused only after their copy has completed)."""
    xs = [O.synthetic_audio(200_000 + 1000 * i, seed=300 + i) for i in range(9)]
    want = [engine.predict(x, 0.96).numpy() for x in xs]          # synchronised one by one
    pending = [engine.predict(x, 0.96) for x in xs]                # enqueued back to back
    for w, p in zip(want, pending):
        assert np.array_equal(p.numpy(), w)


# --------------------------------------------------------------------------- f16 range guard, plain-f16 mode
def _misscaled(eng, layer=6, shift=14):
    """Exponents that drive the GEMM input of `layer` (2..14) `shift` binary orders above where the calibration put it:
    2^8..2^9 becomes 2^22, far beyond what an f16 'hi' half can hold."""
    exps, _ = eng.scales()
    bad = exps.copy()
    bad[layer - 2] += shift
    return exps, bad


def test_activation_beyond_the_f16_range_is_detected_and_recomputed_in_f32(weights_bundle):
    """Plain ReLU bounds nothing (yamnet.py:36-74).  An activation beyond the headroom of its layer's scale would make
    the split-f16 path produce inf / NaN (or a ReLU-masked zero); the engine must notice, and `.numpy()` must hand back
    the exact-f32 result instead.  (The calibration makes this unreachable from PCM - the CNN's input is a log
    spectrum - so the test drives a layer out of range through the exponent hook.)"""
    from buzzdetect_amd.engine import HipEngine
    b = weights_bundle
    x = O.synthetic_audio(HOP * 6 + 15600, seed=77)
    eng = HipEngine()
    try:
        good, bad = _misscaled(eng)
        eng.set_pointwise_mode("f32")
        exact = eng.predict(x, 0.96).numpy().copy()
        assert np.isfinite(exact).all() and not eng.range_exceeded()
        ref = oracle_logits(x, b)
        assert np.abs(exact - ref).max() < 1e-4
        eng.set_activation_exponents(bad)
        assert np.array_equal(eng.predict(x, 0.96).numpy(), exact)          # exact-f32 mode does not use the scales
        for mode in ("f16x3", "f16"):
            eng.set_pointwise_mode(mode)
            before = eng.overflow_reruns
            raw = eng.predict(x, 0.96)
            torch_rows = raw.tensor.clone()
            got = raw.numpy()                               # reads the result's own range word, recomputes
            assert eng.overflow_reruns == before + 1
            assert np.array_equal(got, exact), mode
            assert not np.array_equal(torch_rows.cpu().numpy(), exact)      # what the f16 path had produced was wrong
            assert not eng.range_exceeded()                 # the per-call word took the flag with it
        # back to the calibrated exponents: no flag, no recomputation, and the gate holds
        eng.set_activation_exponents(good)
        eng.set_pointwise_mode("f16x3")
        before = eng.overflow_reruns
        assert np.abs(eng.predict(x, 0.96).numpy() - ref).max() < 1e-4
        quiet = eng.predict(np.zeros(HOP * 2 + 15600, np.float32), 0.96).numpy()
        assert np.isfinite(quiet).all() and eng.overflow_reruns == before
    finally:
        eng.close()


def test_range_word_belongs_to_the_result_not_to_the_engine(weights_bundle):
    """The reference's call pattern (src/inference/worker.py:71-74, src/write/worker.py:69): the analyzer thread enqueues
    predict(N), predict(N+1), ...; the WRITER thread calls N.numpy(), (N+1).numpy() later.  A chunk that left the f16
    range must be the one that is recomputed - not whichever result happens to be read first - and the repeat must not
    disturb the analyzer thread, which keeps predicting on the same engine meanwhile."""
    import threading
    from buzzdetect_amd.engine import HipEngine
    xs = [O.synthetic_audio(HOP * 3 + 15600, seed=500 + i) for i in range(6)]
    eng = HipEngine()
    try:
        good, bad = _misscaled(eng)
        eng.set_pointwise_mode("f32")
        exact = [eng.predict(x, 0.96).numpy().copy() for x in xs]
        eng.set_pointwise_mode("f16x3")
        sound = [eng.predict(x, 0.96).numpy().copy() for x in xs]
        assert eng.overflow_reruns == 0
        # analyzer: chunks 0, 1 with sound scales, chunk 2 with a layer driven out of range, chunks 3.. sound again
        results = [eng.predict(xs[0], 0.96), eng.predict(xs[1], 0.96)]
        eng.set_activation_exponents(bad)           # (synchronises: the first two are complete, unread)
        results.append(eng.predict(xs[2], 0.96))
        eng.set_activation_exponents(good)
        results += [eng.predict(xs[3], 0.96)]
        got, errors = {}, []
        go_on = threading.Event()

        def writer():
            try:
                for i, r in enumerate(results):
                    got[i] = r.numpy().copy()
                    go_on.set()
            except BaseException as exc:                   # noqa: BLE001
                errors.append(exc)
                go_on.set()

        t = threading.Thread(target=writer)
        t.start()
        go_on.wait(60)
        later = [eng.predict(x, 0.96) for x in xs[4:]]     # the analyzer thread keeps going while the writer reads
        t.join(120)
        assert not t.is_alive() and not errors, errors
        assert eng.overflow_reruns == 1                    # chunk 2, and only chunk 2, was repeated
        for i in (0, 1, 3):
            assert np.array_equal(got[i], sound[i]), i
        assert np.array_equal(got[2], exact[2])
        for x_i, r in zip((4, 5), later):
            assert np.array_equal(r.numpy(), sound[x_i])
        # the same through one launch set: only the flagged set's chunks repeat, each against its own input
        eng.set_activation_exponents(bad)
        batch = eng.predict_batch(xs[:3], 0.96)
        eng.set_activation_exponents(good)
        clean = eng.predict_batch(xs[3:], 0.96)
        before = eng.overflow_reruns
        for i, r in enumerate(clean):
            assert np.array_equal(r.numpy(), sound[3 + i])
        assert eng.overflow_reruns == before
        for i, r in enumerate(batch):
            assert np.array_equal(r.numpy(), exact[i])
        assert eng.overflow_reruns == before + 3
    finally:
        eng.close()


def _scaled_blob(weights_bundle, layer, gain):
    """The synthetic embedder with the GEMM input of `layer` (2..14) scaled by `gain` and its 1x1 kernel by 1 / gain:
    depthwise kernel, BatchNorm mean and beta times gain (its output is the GEMM's input, yamnet.py:55-63), pointwise
    kernel divided by it (yamnet.py:64-70).  The network computes the same function (exactly so when gain is a power
    of two) with one layer's activations and weights at a hostile scale."""
    import json
    from buzzdetect_amd import weights as W
    blob = weights_bundle["blob"].copy()
    with open(os.path.join(os.path.dirname(W.__file__), "data", "embedder_manifest.json")) as f:
        entries = {e["name"]: e for e in json.load(f)["tensors"]}
    k = 2 + 4 * (layer - 2)

    def scale(name, factor):
        e = entries[name]
        off, n = e["offset"] // 4, int(np.prod(e["shape"]))
        blob[off:off + n] *= np.float32(factor)

    scale(f"layer_with_weights-{k}/depthwise_kernel", gain)
    scale(f"layer_with_weights-{k + 1}/moving_mean", gain)
    scale(f"layer_with_weights-{k + 1}/beta", gain)
    scale(f"layer_with_weights-{k + 2}/kernel", 1.0 / gain)
    return blob


@pytest.mark.parametrize("layer,gain", [(6, 30.0), (6, 300.0), (6, 3.0e4), (9, 1 / 100.0), (9, 1 / 1000.0), (2, 300.0),
                                        (3, 1 / 1000.0), (14, 2.0 ** 17), (4, 2.0 ** -15)])
def test_hostile_weight_scales_stay_inside_the_gate(weights_bundle, layer, gain):
    """unscaled hi / lo halves are 22-bit operands only while activations AND folded weights are
    O(0.1 .. 1).  With per-row weight scales and calibrated activation scales the split-f16 path must not care: one
    layer's GEMM input times 30 .. 3e4 (weights divided), or divided by 100 .. 1000 (weights multiplied) - fused and
    one-kernel-per-op paths within max(1e-4 x logit scale, 2 x the exact-f32 path's own error) of the f64 oracle, and
    no chunk repeated in f32."""
    from buzzdetect_amd.engine import HipEngine
    b = weights_bundle
    blob = _scaled_blob(b, layer, gain)
    x = O.synthetic_audio(HOP * 7 + 15600, seed=int(layer * 1000 + abs(np.log2(gain))))
    ref = O.predict(x, blob, b["mel"], b["head_kernel"], b["head_bias"], HOP, STEP, np.float64)
    scale = max(1.0, float(np.abs(ref).max()))
    eng = HipEngine(embedder_blob=blob)
    try:
        exps, maxima = eng.scales()
        assert np.all(maxima > 0) and np.all(np.ldexp(maxima, exps) >= 256.0) and np.all(np.ldexp(maxima, exps) < 512.0)
        eng.set_pointwise_mode("f32")
        err_f32 = np.abs(eng.predict(x, 0.96).numpy() - ref).max()
        assert err_f32 < 1e-4 * scale
        eng.set_pointwise_mode("f16x3")
        bound = max(1e-4 * scale, 2 * err_f32)
        fused = eng.predict(x, 0.96).numpy().copy()
        assert np.abs(fused - ref).max() < bound, (np.abs(fused - ref).max(), err_f32)
        eng.set_fusion(False, False)
        plain = eng.predict(x, 0.96).numpy()
        assert np.array_equal(plain, fused)                # bit-identical paths at any scale
        assert eng.overflow_reruns == 0
        if gain in (2.0 ** 17, 2.0 ** -15):                # a power of two changes nothing at all
            base = HipEngine()
            try:
                assert np.array_equal(base.predict(x, 0.96).numpy(), fused)
            finally:
                base.close()
    finally:
        eng.close()


def _channel_scaled_blob(weights_bundle, layer, exps):
    """As _scaled_blob, but with a power of two PER CHANNEL: input channel c of `layer`'s 1x1 convolution is carried
    2^exps[c] times larger (depthwise kernel, BatchNorm mean and beta of that channel), column c of the 1x1 kernel
    2^-exps[c] times smaller.  The network is exactly the same function."""
    import json
    from buzzdetect_amd import weights as W
    blob = weights_bundle["blob"].copy()
    with open(os.path.join(os.path.dirname(W.__file__), "data", "embedder_manifest.json")) as f:
        entries = {e["name"]: e for e in json.load(f)["tensors"]}
    k = 2 + 4 * (layer - 2)

    def view(name):
        e = entries[name]
        off, n = e["offset"] // 4, int(np.prod(e["shape"]))
        return blob[off:off + n].reshape(e["shape"])

    g = np.ldexp(np.float32(1.0), np.asarray(exps, dtype=np.int32)).astype(np.float32)
    view(f"layer_with_weights-{k}/depthwise_kernel")[:] *= g[None, None, :, None]
    view(f"layer_with_weights-{k + 1}/moving_mean")[:] *= g
    view(f"layer_with_weights-{k + 1}/beta")[:] *= g
    view(f"layer_with_weights-{k + 2}/kernel")[:] /= g[None, None, :, None]
    return blob


@pytest.mark.parametrize("layer,spread", [(6, 8), (9, 10), (12, 12)])
def test_channels_of_one_layer_at_very_different_scales(weights_bundle, layer, spread):
    """What BatchNorm folding does to real weights: the channels of one layer's GEMM input sit at different scales, with
    the 1x1 kernel's columns compensating.  One activation exponent per layer has to serve them all: with the layer's
    largest channel at 2^8..2^9, a channel 2^spread smaller still has a normal low half down to 2^(spread - 11) of ITS
    largest value.  Channels 2^8 .. 2^12 apart (random per channel, an exactly equivalent network) stay inside the gate."""
    from buzzdetect_amd.engine import HipEngine
    b = weights_bundle
    cin = {6: 256, 9: 512, 12: 512}[layer]
    rng = np.random.default_rng(layer * 31 + spread)
    exps = -rng.integers(0, spread + 1, size=cin)
    blob = _channel_scaled_blob(b, layer, exps)
    x = O.synthetic_audio(HOP * 7 + 15600, seed=7000 + layer)
    ref = O.predict(x, blob, b["mel"], b["head_kernel"], b["head_bias"], HOP, STEP, np.float64)
    base = oracle_logits(x, b)
    assert np.abs(ref - base).max() < 1e-9                 # the same function, as constructed
    eng = HipEngine(embedder_blob=blob)
    try:
        eng.set_pointwise_mode("f32")
        err_f32 = np.abs(eng.predict(x, 0.96).numpy() - ref).max()
        eng.set_pointwise_mode("f16x3")
        got = eng.predict(x, 0.96).numpy()
        assert np.abs(got - ref).max() < max(1e-4, 2 * err_f32), (np.abs(got - ref).max(), err_f32)
        assert eng.overflow_reruns == 0
    finally:
        eng.close()


def test_scales_follow_calibration_audio():
    """bd_calibrate only ever widens: a pass over more audio leaves every per-layer maximum at least where it was, the
    exponents follow the maxima, and results stay inside the gate's noise (a power-of-two change of scale moves only
    which values have a subnormal low half)."""
    from buzzdetect_amd.engine import HipEngine
    eng = HipEngine()
    try:
        x = O.synthetic_audio(HOP * 9 + 15600, seed=91)
        before = eng.predict(x, 0.96).numpy().copy()
        exps0, max0 = eng.scales()
        eng.calibrate(8.0 * x)                             # louder than anything in [-1, 1]
        exps1, max1 = eng.scales()
        assert np.all(max1 >= max0) and np.all(exps1 <= exps0)
        assert np.all(np.ldexp(max1, exps1) >= 256.0) and np.all(np.ldexp(max1, exps1) < 512.0)
        after = eng.predict(x, 0.96).numpy()
        if np.array_equal(exps0, exps1):
            assert np.array_equal(after, before)
        else:
            assert np.abs(after - before).max() < 1e-5
        assert eng.overflow_reruns == 0
    finally:
        eng.close()


def test_plain_f16_mode_is_close_but_outside_the_gate(engine, weights_bundle):
    """Mode 'f16' (BASELINE config 5): one MFMA per product.  Reported separately from the 1e-4 gate: the error is
    ~1e-3 of the logit scale, far from garbage, and the default mode is unaffected afterwards."""
    b = weights_bundle
    x = O.synthetic_audio(HOP * 40 + 15600, seed=78)
    ref = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], HOP, STEP, np.float64)
    default = engine.predict(x, 0.96).numpy().copy()
    engine.set_pointwise_mode("f16")
    try:
        plain = engine.predict(x, 0.96).numpy().copy()
    finally:
        engine.set_pointwise_mode("f16x3")
    err_plain, err_default = np.abs(plain - ref).max(), np.abs(default - ref).max()
    assert err_default < 1e-4
    assert 1e-5 < err_plain < 5e-2, err_plain
    assert np.array_equal(engine.predict(x, 0.96).numpy(), default)
    assert engine.overflow_reruns == 0


@pytest.mark.parametrize("windows", [2, 255, 256, 257, 513, 1025])
def test_persistent_kernels_at_grid_boundaries(engine, windows):
    """The layer-4 window kernel and the resident-weights pointwise kernel are persistent (256 workgroups): fewer windows than
    workgroups, exactly as many, one more, two and a bit per workgroup, and a second CNN pass (1025 = 1024 + 1) must all give
    the bits of the one-kernel-per-op path."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=4000 + windows)
    engine.set_pointwise_mode("f16x3")
    try:
        engine.set_fusion(False, False)
        ref = engine.predict(x, 0.96).numpy()
        ref_emb = engine.embed(x, 0.96).numpy()
        engine.set_fusion(True, True)
        assert ref.shape == (windows, 13)
        assert np.array_equal(engine.predict(x, 0.96).numpy(), ref)
        assert np.array_equal(engine.embed(x, 0.96).numpy(), ref_emb)
    finally:
        engine.set_fusion(True, True)


@pytest.mark.parametrize("windows", [1, 3, 37, 300, 678, 1024, 1090])
def test_fused_f32_mode_equals_one_kernel_per_op(engine, windows):
    """Exact-f32 mode (bd_set_pointwise_mode 0) in its fused layouts against the same mode as one kernel per op
    (conv1_kernel, depthwise_kernel, pointwise_kernel): the same chains of IEEE operations, so logits AND embeddings agree
    bit for bit; whole hop and half hop, partial tiles of every layer (windows x positions is not a multiple of the 96 ..
    512-row tiles; windows not a multiple of the 4 / 16 windows of a tile of the depthwise epilogue), two passes (1090).
    Default: layers 1-3 as stem_reg_f32_kernel, layer 4 + depthwise 5 as l4_reg_f32_kernel, the two on-chip runs, layers 13 / 14
    as 1x1 kernels with the next depthwise / the pool in their epilogue (pointwise_kernel<96, 128, 1, 4, NH, NW, NS>)."""
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=windows)
    engine.set_pointwise_mode("f32")
    try:
        engine.set_fusion(False, False)
        ref = engine.predict(x, 0.96).numpy()
        ref_emb = engine.embed(x, 0.96).numpy()
        ref_half = engine.predict(x[: HOP * 40], 0.48).numpy()
        # the default (round 5): layers 1-3 as stem3_f32_kernel, layer 4 + depthwise 5 as l4_f32_kernel, pointwise 5 + layers 6-7
        # and layers 8-12 + depthwise 13 as the two on-chip runs (sepmidf32.hip, sepchipf32.hip), layers 13 / 14 as 1x1 kernels
        # with the next depthwise / the pool in their epilogue; 10 = without the middle run (the chip run then takes the
        # depthwise-8 output)
        for code in (True, 10):
            engine.set_fusion(True, code)
            assert np.array_equal(engine.predict(x, 0.96).numpy(), ref), code
            assert np.array_equal(engine.embed(x, 0.96).numpy(), ref_emb), code
            assert np.array_equal(engine.predict(x[: HOP * 40], 0.48).numpy(), ref_half), code
        assert ref.shape == (windows, 13)
        for gone in ((4, 1), (2, 1), (3, 6), (3, 2), (3, 3), (3, 4), (3, 5), (3, 7), (3, 8), (3, 9), (3, 11), (3, 12)):   # removed in round 6: refused
            with pytest.raises(Exception):
                engine.set_fusion(*gone)
    finally:
        engine.set_fusion(True, True)
        engine.set_pointwise_mode("f16x3")


def test_three_streams_under_load_give_the_idle_gpu_bits(engine):
    """The fused kernels overlay LDS tiles and hand data from layer to layer on the CU inside one launch: an ordering mistake there would only show under load.  240 batches on three analyzer streams
    (an engine each, as bench.py and the pipeline run them), every batch compared on the device with the logits it gave on
    an idle GPU.  One mismatch counter PER STREAM (a shared one would be a non-atomic read-modify-write from three
    streams: a count could be lost), every side stream ordered behind the set-up on the current stream, and a negative
    control: with one reference row corrupted the counters must add up to exactly the batches that use it.
    (tools/stress_identity.py is the long form.)"""
    import torch
    from buzzdetect_amd.engine import HipEngine
    dev = torch.device("cuda", 0)
    engs = [engine, HipEngine(embeddername="yamnet_k2", modelname="model_general_v3"),
            HipEngine(embeddername="yamnet_k2", modelname="model_general_v3")]
    streams = [torch.cuda.Stream(dev) for _ in engs]
    sizes = [1024, 1024, 1024, 678]
    gen = torch.Generator(device="cpu").manual_seed(77)
    parts = [(torch.randn(HOP * (n - 1) + 15600, generator=gen) * 0.1).to(dev) for n in sizes]
    ref = []
    for x, n in zip(parts, sizes):
        out = torch.empty((n, 13), device=dev)
        engine.launch([x], HOP, STEP, False, True, out=out)
        torch.cuda.synchronize()
        ref.append(out)
    ring = [[torch.empty((1024, 13), device=dev) for _ in range(4)] for _ in engs]

    def run(reference):
        bad = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in engs]     # one counter per stream
        torch.cuda.synchronize()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        for k in range(240):
            j, b = k % 3, k % 4
            out = ring[j][(k // 3) % 4][:sizes[b]]
            with torch.cuda.stream(streams[j]):
                engs[j].launch([parts[b]], HOP, STEP, False, True, out=out)
                bad[j] += (out != reference[b]).any().to(torch.int64)
        torch.cuda.synchronize()
        return sum(int(c.item()) for c in bad)

    try:
        assert run(ref) == 0
        broken = [r.clone() for r in ref]
        broken[3][677, 12] += 1.0                      # the last row of the 678-window batch: batches k % 4 == 3
        assert run(broken) == 60
    finally:
        for e in engs[1:]:
            e.close()
