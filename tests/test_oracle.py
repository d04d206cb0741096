"""The CPU oracle against what is pinned: graph constants, closed forms, reference helper vectors,
and an independent torch-CPU restatement."""
import hashlib
import json
import os

import numpy as np
import pytest

from buzzdetect_amd import weights as W
from oracle import yamnet_oracle as O

REF = "/root/reference"


def test_params_match_reference_dataclass(golden_helpers):
    p = golden_helpers["params"]                      # produced by importing embedders/yamnet/params.py
    assert O.SAMPLE_RATE == p["sample_rate"]
    assert O.STFT_WINDOW == int(round(p["sample_rate"] * p["stft_window_seconds"]))
    assert O.STFT_HOP == int(round(p["sample_rate"] * p["stft_hop_seconds"]))
    assert O.FFT_LENGTH == 2 ** int(np.ceil(np.log(O.STFT_WINDOW) / np.log(2.0)))
    assert O.MEL_BANDS == p["mel_bands"] and O.LOG_OFFSET == p["log_offset"]
    assert O.PATCH_FRAMES == p["patch_frames"] and O.BN_EPS == p["batchnorm_epsilon"]
    assert O.MIN_SAMPLES == int((p["patch_window_seconds"] + p["stft_window_seconds"] - p["stft_hop_seconds"])
                                * p["sample_rate"])
    assert p["batchnorm_scale"] is False and p["batchnorm_center"] is True and p["conv_padding"] == "same"


@pytest.mark.parametrize("n,whole,half", [(0, 1, 1), (1, 1, 1), (15360, 1, 1), (15600, 1, 1), (15601, 2, 2),
                                          (23360, 2, 3), (3_194_880, 208, 415), (9_600_000, 625, 1249),
                                          (15_728_640, 1024, 2047)])
def test_window_count_closed_forms(n, whole, half):
    # SURVEY §8a known answers (from the graph constants 15600 / 15360 / 7680)
    assert O.num_windows(n, 15360) == whole
    assert O.num_windows(n, 7680) == half
    for hop, expect in ((15360, whole), (7680, half)):
        closed = 1 if n <= 15600 else 1 + int(np.ceil(np.float32(n - 15600) / np.float32(hop)))
        assert closed == expect


def test_hop_derivation():
    assert O.hop_samples(1.0) == 15360 and O.hop_samples(0.5) == 7680
    assert O.patch_step(1.0) == 96 and O.patch_step(0.5) == 48
    # free hop (Keras-3 yamnet): pad hop and patch step need not agree
    assert O.hop_samples(0.3) == 4608 and O.patch_step(0.3) == 29
    # the product is rounded to float32 before the truncating cast (tf.cast of a Python float):
    # 0.96 * p * 16000 = 5375.999..., 10751.999..., 14591.999... in float64
    assert [O.hop_samples(p) for p in (0.35, 0.7, 0.95)] == [5376, 10752, 14592]
    from buzzdetect_amd import engine as E
    for p in (1.0, 0.5, 0.3, 0.35, 0.7, 0.95, 0.1, 0.25, 0.9):
        assert E.hop_samples(0.96 * p) == O.hop_samples(p) and E.patch_step(0.96 * p) == O.patch_step(p)


def test_mel_constant_is_the_graph_constant():
    man = W.manifest()["sha256"]
    for name, fn in (("yamnet_k2", "mel_yamnet_k2_257x64.f32"), ("yamnet", "mel_yamnet_keras3_257x64.f32")):
        m = W.load_mel(name)
        assert hashlib.sha256(m.astype("<f4").tobytes()).hexdigest() == man[fn]
        assert m.shape == (257, 64) and np.count_nonzero(m) == 461
        assert np.all(m[0] == 0)                       # DC row is zero
        assert abs(float(m.sum(dtype=np.float64)) - 230.9136) < 1e-3
    assert man["mel_yamnet_k2_257x64.f32"].startswith("a77c17ab")
    assert man["mel_yamnet_keras3_257x64.f32"].startswith("3dfc9b74")
    # the Keras-3 graph carries a float-noise variant of the same matrix: close, not bit-equal
    assert 0 < np.abs(W.load_mel("yamnet") - W.load_mel("yamnet_k2")).max() < 1e-5


def test_mel_matrix_matches_htk_formula():
    # tf.signal.linear_to_mel_weight_matrix(64, 257, 16000, 125, 7500) restated in float64
    mel = lambda f: 1127.0 * np.log1p(f / 700.0)  # noqa: E731
    bins = np.linspace(0.0, 8000.0, 257)[1:]
    edges = np.linspace(mel(125.0), mel(7500.0), 66)
    sm = mel(bins)[:, None]
    lo, ce, up = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    w = np.maximum(0.0, np.minimum((sm - lo) / (ce - lo), (up - sm) / (up - ce)))
    w = np.pad(w, ((1, 0), (0, 0)))
    assert np.abs(w - W.load_mel("yamnet_k2")).max() < 2e-5


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_fixtures_regenerate_from_reference():
    from buzzdetect_amd import artifacts as A
    m = A.extract_mel_matrix(os.path.join(REF, "embedders/yamnet_k2/models/yamnet_wholehop/saved_model.pb"))
    assert np.array_equal(m, W.load_mel("yamnet_k2"))
    idx_p, dat_p = A.bundle_paths(os.path.join(REF, "models/model_general_v3"))
    idx = A.read_bundle_index(idx_p)
    k = A.read_bundle_tensor(dat_p, idx["layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"])
    assert np.array_equal(k, W.load_head().kernel)
    eidx = A.read_bundle_index(os.path.join(
        REF, "embedders/yamnet_k2/models/yamnet_wholehop/variables/variables.index"))
    table = {t["name"]: (tuple(t["shape"]), t["offset"]) for t in W.manifest()["tensors"]}
    for e in eidx.values():
        if e.dtype == 1:
            assert table[e.name.replace("/.ATTRIBUTES/VARIABLE_VALUE", "")] == (e.shape, e.offset)


def test_blob_layout_tables_agree():
    assert W.expected_table() == W.blob_table()
    assert W.manifest()["payload_bytes"] == 4 * W.EMBEDDER_BLOB_FLOATS == 12_869_376
    t = O.split_blob(W.synthetic_embedder_blob())
    assert t["layer_with_weights-0/kernel"].shape == (3, 3, 1, 32)
    assert t["layer_with_weights-52/kernel"].shape == (1, 1, 1024, 1024)
    assert len(t) == 108


def test_build_py_smoke_shape(weights_bundle):
    # embedders/yamnet/BUILD.py:20-26: zeros(15360) -> exactly one window
    b = weights_bundle
    out = O.predict(np.zeros(15360, np.float32), b["blob"], b["mel"], b["head_kernel"], b["head_bias"])
    assert out.shape == (1, 13)


def test_silence_hits_the_log_floor(weights_bundle):
    lm = O.log_mel(O.pad_waveform(np.zeros(20000, np.float32), 15360), weights_bundle["mel"])
    assert lm.shape == (192, 64)
    assert np.all(lm == np.float32(np.log(np.float32(0.001))))


def test_f32_oracle_close_to_f64(weights_bundle):
    b = weights_bundle
    x = O.synthetic_audio(15360 * 3 + 1000, seed=5)
    a = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], dtype=np.float32)
    d = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], dtype=np.float64)
    assert a.shape == d.shape == (4, 13)
    assert np.abs(a - d).max() < 1e-4


@pytest.mark.parametrize("hop,step", [(15360, 96), (7680, 48), (4608, 29)])
def test_torch_restatement_agrees_with_numpy_oracle(weights_bundle, hop, step):
    torch = pytest.importorskip("torch")
    from oracle.torch_baseline import TorchYamnet
    b = weights_bundle
    x = O.synthetic_audio(15360 * 2 + 777, seed=9)
    ref = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], hop, step, np.float64)
    tm = TorchYamnet(b["blob"], b["mel"], b["head_kernel"], b["head_bias"], dtype=torch.float64)
    got = tm.predict(x, hop, step)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-9
    t32 = TorchYamnet(b["blob"], b["mel"], b["head_kernel"], b["head_bias"], dtype=torch.float32).predict(x, hop, step)
    assert np.abs(t32 - ref).max() < 1e-4


def _stft_magnitude_by_dft_matrix(wave_padded: np.ndarray, dtype) -> np.ndarray:
    """The reference's OWN TensorFlow-free definition of |STFT| (embedders/yamnet/features.py:111-165, the `tflite_compatible`
    branch), restated in NumPy: Hann sampled as 0.5 - 0.5 cos(2 pi t) on t = arange(0, 1, 1/400) (:115-119), each 400-sample
    frame zero-padded to 512 on BOTH sides (56 + 56, :141-151), multiplied by the first 257 columns of the full
    exp(+2 pi i j k / 512) matrix (:121-136, :152-153), magnitude as sqrt(re^2 + im^2) (:155-156).  The main branch
    (features.py:42-46) pads on the right only and uses exp(-...): a circular shift and a conjugation, neither of which
    changes a magnitude - which is why the two branches of the reference agree, and why this pins the oracle's."""
    t = np.arange(0, 1.0, 1.0 / O.STFT_WINDOW)
    assert t.shape == (O.STFT_WINDOW,)
    window = (0.5 - 0.5 * np.cos(2 * np.pi * t)).astype(dtype)
    n_frames = 1 + (wave_padded.shape[0] - O.STFT_WINDOW) // O.STFT_HOP
    frames = wave_padded.astype(dtype)[(np.arange(n_frames) * O.STFT_HOP)[:, None] + np.arange(O.STFT_WINDOW)[None, :]]
    frames = frames * window[None, :]
    half_pad = (O.FFT_LENGTH - O.STFT_WINDOW) // 2
    frames = np.pad(frames, ((0, 0), (half_pad, O.FFT_LENGTH - O.STFT_WINDOW - half_pad)))
    jk = np.outer(np.arange(O.FFT_LENGTH), np.arange(O.FFT_LENGTH // 2 + 1))
    dft = np.exp(2j * np.pi * jk / float(O.FFT_LENGTH))
    re = frames @ np.real(dft).astype(dtype)
    im = frames @ np.imag(dft).astype(dtype)
    return np.sqrt(re * re + im * im)


def test_front_end_matches_the_references_tf_free_stft_definition(weights_bundle):
    """Row (c) of SURVEY 8: the VALUES of the front end have no reference-held vector, but its DEFINITION does exist in the
    reference as plain NumPy constants + two matmuls (features.py:111-165).  The oracle's rfft-based log-mel must agree with
    that definition to float64 round-off, and in float32 within the bound the device is held to (5e-5)."""
    mel = weights_bundle["mel"]
    x = O.pad_waveform(O.synthetic_audio(15360 * 3, seed=77), 15360)       # 3 windows = 288 frames
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 5e-5)):
        mag = _stft_magnitude_by_dft_matrix(x, dtype)
        assert mag.shape == (288, 257)
        ref = np.log(mag @ mel.astype(dtype) + dtype(O.LOG_OFFSET))
        got = O.log_mel(x, mel, dtype=dtype)
        assert got.shape == ref.shape == (288, 64)
        assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() < tol, dtype
    # the float32 oracle against the float64 DEFINITION (what the device's 5e-5 is measured from)
    ref64 = np.log(_stft_magnitude_by_dft_matrix(x, np.float64) @ mel.astype(np.float64) + O.LOG_OFFSET)
    assert np.abs(O.log_mel(x, mel, dtype=np.float32).astype(np.float64) - ref64).max() < 5e-5
    # the window: arange(0, 1, 1/400) and 2 pi k / 400 are the same periodic Hann to float32 round-off (the oracle evaluates the graph's float32 op chain: two ulps)
    t = np.arange(0, 1.0, 1.0 / O.STFT_WINDOW)
    assert np.abs((0.5 - 0.5 * np.cos(2 * np.pi * t)).astype(np.float32) - O.hann_periodic(np.float32)).max() < 3e-7


def test_torch_second_opinion_on_a_full_batch(weights_bundle):
    """The torch-CPU restatement (F.conv2d, torch.stft) against the NumPy oracle at a full 1024-window batch in float32: two
    independent CPU implementations of the same path agree within the gate the device is held to."""
    torch = pytest.importorskip("torch")
    from oracle.torch_baseline import TorchYamnet
    b = weights_bundle
    x = O.synthetic_audio(15360 * 1023 + 15600, seed=1024)
    tm = TorchYamnet(b["blob"], b["mel"], b["head_kernel"], b["head_bias"], dtype=torch.float32)
    got = tm.predict(x, 15360, 96)
    assert got.shape == (1024, 13)
    idx = np.r_[0:8, 500:508, 1016:1024]                      # the NumPy oracle on the first, middle and last windows
    for w in (0, 500, 1016):
        seg = x[w * 15360: (w + 7) * 15360 + 15600]
        ref = O.predict(seg, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
        assert ref.shape == (8, 13)
        assert np.abs(got[w:w + 8] - ref).max() < 1e-4, w
    assert np.isfinite(got).all() and idx.size == 24


def test_same_padding_is_asymmetric_for_stride_2():
    assert O._same_pad(96, 3, 2) == (48, 0, 1)
    assert O._same_pad(64, 3, 2) == (32, 0, 1)
    assert O._same_pad(6, 3, 2) == (3, 0, 1)
    assert O._same_pad(48, 3, 1) == (48, 1, 1)


def test_golden_hotpath_fixture_reproduces(weights_bundle):
    """tests/golden/hotpath_oracle_f64.npz is the oracle's own output (tools/make_golden_hotpath.py),
    committed so that the GPU box and later rounds compare against identical numbers."""
    b = weights_bundle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_oracle_f64.npz"))
    x = O.synthetic_audio(int(g["n_samples"]), seed=int(g["seed"]))
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["audio_sha256"])
    got = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert np.abs(got - g["logits_whole"]).max() < 1e-10
