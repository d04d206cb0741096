"""The oracle's STRUCTURE against the reference's own graphs.

``tests/golden/graph_facts.json`` is decoded by ``tools/make_fixtures.py`` from the three ``saved_model.pb`` files the
reference ships (``embedders/yamnet_k2/models/yamnet_{wholehop,halfhop}``, ``embedders/yamnet``): op types, attributes and
constants of the inlined inference function.  Everything the oracle "restates from TensorFlow's behaviour" and that a graph
can say is asserted here: strides / SAME / NHWC of the 27 convolutions, inference-mode batch norms with epsilon 1e-4 and a
scale of ones, the zero pad AFTER the 400 samples in front of the RFFT, the periodic Hann computed as
0.5 - 0.5 cos((2 pi k) / 400), |.| -> matmul -> + 0.001 -> log, ReLU (not ReLU6), the mean over axes [1, 2], and the
float32 ceil of pad_waveform.  Attributes the serialiser stripped because they equal TensorFlow's op-definition defaults
come back as null; the defaults are named where they are used.
"""
import json
import os

import numpy as np
import pytest

from oracle import yamnet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# TensorFlow op-definition defaults (tensorflow/core/ops/nn_ops.cc, math_ops.cc) for attributes a SavedModel omits
TF_DEFAULT_BN_EPSILON = 1e-4          # FusedBatchNormV3: "epsilon: float = 0.0001"
TF_DEFAULT_DATA_FORMAT = "NHWC"       # Conv2D / DepthwiseConv2dNative / FusedBatchNormV3
DT_FLOAT, DT_INT32 = 1, 3             # tensorflow/core/framework/types.proto

GRAPHS = {"yamnet_k2_wholehop": 1.0, "yamnet_k2_halfhop": 0.5, "yamnet_keras3": 1.0}


@pytest.fixture(scope="module")
def facts():
    with open(os.path.join(HERE, "golden", "graph_facts.json")) as f:
        return json.load(f)


def _const(g, suffix):
    hits = [v for k, v in g["frontend_consts"].items() if k.endswith(suffix)]
    assert len(hits) == 1, suffix
    return hits[0]


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_front_end_constants(facts, name):
    g, hop_prop = facts[name], GRAPHS[name]
    assert g["inference_functions"] >= 1
    assert _const(g, "stft/frame_length") == O.STFT_WINDOW == 400
    assert _const(g, "stft/frame_step") == O.STFT_HOP == 160
    assert _const(g, "stft/fft_length") == O.FFT_LENGTH == 512
    assert _const(g, "stft/rfft/fft_length") == [O.FFT_LENGTH]
    assert O.N_BINS == O.FFT_LENGTH // 2 + 1
    # zero padding goes AFTER the windowed frame: [[0, 0], [0, 512 - 400]]
    assert _const(g, "rfft/Pad/paddings") == [[0, 0], [0, O.FFT_LENGTH - O.STFT_WINDOW]]
    # patches: 96 frames, step 96 (whole hop) / 48 (half hop), along axis 0; reshaped to (96, 64, 1)
    assert _const(g, "frame/frame_length") == O.PATCH_FRAMES == 96
    assert _const(g, "frame/frame_step") == O.patch_step(hop_prop)
    frame_axes = sorted(v for k, v in g["frontend_consts"].items() if k.endswith("/frame/axis"))
    assert frame_axes == [-1, 0]                      # stft frames the last axis, the patch framing axis 0
    if name != "yamnet_k2_halfhop":                    # (the half-hop graph names its Reshape layer differently)
        assert [_const(g, f"Reshape/shape/{i}") for i in (1, 2, 3)] == [O.PATCH_FRAMES, O.MEL_BANDS, 1]
    # the serving call's arguments: 15600, the hop as int32 AND as float32 (the float32 division of pad_waveform), 0.001
    hop = O.hop_samples(hop_prop)
    consts = [tuple(c) for c in g["main_graph_scalar_consts"]]
    assert ("int32", O.MIN_SAMPLES) in consts and O.MIN_SAMPLES == 15600
    assert ("int32", hop) in consts and ("float32", float(hop)) in consts
    assert ("float32", float(np.float32(O.LOG_OFFSET))) in consts
    assert len(consts) == 5 and ("int32", 0) in consts


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_hann_window_is_periodic_and_float32(facts, name):
    g = facts[name]
    assert _const(g, "hann_window/periodic") is True
    assert _const(g, "hann_window/Const") == float(np.float32(2.0 * np.pi))       # the oracle's dtype(2 pi)
    assert _const(g, "hann_window/mul_2/x") == 0.5 and _const(g, "hann_window/sub_2/x") == 0.5
    # denominator = N + periodic * (1 - N % 2) - 1 = 400 for even N; then 0.5 - 0.5 * cos((2 pi * k) / denominator)
    assert [op for _, op in g["hann_window_ops"]] == [
        "Cast", "FloorMod", "Sub", "Mul", "AddV2", "Sub", "Cast", "Range", "Cast", "Mul", "RealDiv", "Cos", "Mul", "Sub"]
    w = O.hann_periodic(np.float32)
    k = np.arange(400, dtype=np.float32)
    assert np.array_equal(w, np.float32(0.5) - np.float32(0.5) * np.cos((np.float32(2.0 * np.pi) * k) / np.float32(400)))
    assert w[0] == 0.0 and w[200] == 1.0 and w[399] > 0.0                        # periodic: the last sample is not zero


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_front_end_op_chain(facts, name):
    ops = [(op, attrs) for _, op, attrs in facts[name]["frontend_ops"]]
    kinds = [op for op, _ in ops]
    # pad_waveform (features.py:82-108): max / sub / cast to FLOAT32 / divide / ceil / cast back to int32 / ... / Pad
    assert kinds[:16] == ["Shape", "StridedSlice", "Maximum", "Sub", "Cast", "RealDiv", "Ceil", "Cast", "Mul", "Sub", "Maximum",
                          "Sub", "AddV2", "Pack", "Pack", "Pad"]
    casts = [a for op, a in ops if op == "Cast"]
    assert casts[0] == {"DstT": {"dtype": DT_FLOAT}, "SrcT": {"dtype": DT_INT32}}
    assert casts[1] == {"DstT": {"dtype": DT_INT32}, "SrcT": {"dtype": DT_FLOAT}}
    # window multiply, pad, RFFT, magnitude (not power), mel matmul (no transposes), + 0.001, natural log
    assert kinds[16:] == ["Mul", "Pack", "Pad", "RFFT", "ComplexAbs", "MatMul", "AddV2", "Log"]
    matmul = [a for op, a in ops if op == "MatMul"][0]
    assert not matmul.get("transpose_a") and not matmul.get("transpose_b")
    # and the oracle's f32 ceil reproduces the graph's arithmetic on the lengths where float64 would differ
    for n in (15600, 15601, 15600 + 15360, 15600 + 15360 + 1, 3_194_880, 16_777_215):
        hop = 15360
        after = max(n, 15600) - 15600
        hops = int(np.ceil(np.float32(after) / np.float32(hop)))
        assert O.padded_length(n, hop) == n + max(0, 15600 - n) + hop * hops - after


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_cnn_layer_table(facts, name):
    g = facts[name]
    convs, bns, acts = g["convs"], g["batchnorms"], g["activations"]
    assert len(convs) == len(bns) == len(acts) == 27 == 1 + 2 * (len(O.LAYER_DEFS) - 1)
    # layer 1: full 3x3 convolution, stride 2; layers 2..14: depthwise 3x3 (stride from the table) + 1x1 (stride 1)
    expect = [("layer1/conv/Conv2D", "Conv2D", O.LAYER_DEFS[0][0])]
    channels = [O.LAYER_DEFS[0][1]]
    cin = O.LAYER_DEFS[0][1]
    for i, (stride, cout) in enumerate(O.LAYER_DEFS[1:], start=2):
        expect.append((f"layer{i}/depthwise_conv/depthwise", "DepthwiseConv2dNative", stride))
        expect.append((f"layer{i}/pointwise_conv/Conv2D", "Conv2D", 1))
        channels += [cin, cout]
        cin = cout
    for c, (nm, op, stride) in zip(convs, expect):
        assert (c["name"], c["op"]) == (nm, op)
        assert c["strides"] == [1, stride, stride, 1]
        assert c["padding"] == "SAME"                              # _same_pad: total // 2 before, the rest after
        assert (c["data_format"] or TF_DEFAULT_DATA_FORMAT) == "NHWC"
        assert c["dilations"] in (None, [1, 1, 1, 1]) and not c["explicit_paddings"]
    for b, ch in zip(bns, channels):
        assert b["is_training"] is False
        assert (b["epsilon"] if b["epsilon"] is not None else TF_DEFAULT_BN_EPSILON) == pytest.approx(O.BN_EPS)
        assert b["scale_is_const"] and b["scale_all_ones"]         # params.batchnorm_scale = False: gamma == 1
        assert b["channels"] == ch
        assert (b["data_format"] or TF_DEFAULT_DATA_FORMAT) == "NHWC"
    assert {op for _, op in acts} == {"Relu"}                      # plain ReLU, no ReLU6, no sigmoid head
    assert g["pool"]["op"] == "Mean" and g["pool"]["reduction_indices"] == [1, 2] and not g["pool"]["keep_dims"]
    # the stride-2 layers are the ones whose SAME padding is asymmetric (0 before, 1 after) on the even extents
    strided = [i for i, (s, _) in enumerate(O.LAYER_DEFS, start=1) if s == 2]
    assert strided == [1, 3, 5, 7, 13]
    assert O._same_pad(96, 3, 2) == (48, 0, 1) and O._same_pad(48, 3, 1) == (48, 1, 1)


def test_library_constants_match_the_graph(facts):
    import re
    text = open(os.path.join(os.path.dirname(HERE), "include", "buzzdetect_hip.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"#define\s+(BD_[A-Z_]+)\s+\(?(-?\d+)\)?", text)}
    g = facts["yamnet_k2_wholehop"]
    assert consts["BD_STFT_WINDOW"] == _const(g, "stft/frame_length")
    assert consts["BD_STFT_HOP"] == _const(g, "stft/frame_step")
    assert consts["BD_FFT_LENGTH"] == _const(g, "stft/fft_length")
    assert consts["BD_PATCH_FRAMES"] == _const(g, "frame/frame_length")
    assert ["int32", consts["BD_MIN_SAMPLES"]] in g["main_graph_scalar_consts"]
    src = open(os.path.join(os.path.dirname(HERE), "buzzdetect_amd", "csrc", "engine.hip")).read()
    table = re.search(r"kLayerDefs\[14\]\[2\] = \{(.*?)\};", src, flags=re.S).group(1)
    pairs = [tuple(int(x) for x in p) for p in re.findall(r"\{(\d+),\s*(\d+)\}", table)]
    assert tuple(pairs) == O.LAYER_DEFS
    assert "1e-4" in src                                           # fold_bn's epsilon


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_graph_facts_regenerate_from_reference(facts, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    try:
        import make_fixtures
    finally:
        sys.path.pop(0)
    make_fixtures.graph_facts(REF, str(tmp_path))
    with open(tmp_path / "graph_facts.json") as f:
        assert json.load(f) == facts
