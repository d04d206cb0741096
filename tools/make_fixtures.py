#!/usr/bin/env python3
"""Generate the committed data fixtures from the read-only reference checkout.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tools/make_fixtures.py [--reference /root/reference]

Two kinds of output, both *data* (no reference source text is copied):

1. ``buzzdetect_amd/data/`` — constants the product path needs at run time:
   the ``[257,64]`` mel matrices baked into the reference SavedModel graphs, the
   real dense-head weights of ``model_general_v3``, the tensor manifest of the
   YAMNet embedder bundle (names / shapes / offsets — the weights themselves are
   not in the checkout, see ``.MISSING_LARGE_BLOBS``), ``config_model.json`` and
   the model's ``tests/metrics.csv`` threshold table.
2. ``tests/golden/`` — input/output vectors produced by *importing and calling*
   the reference's pure-Python helpers that sit either side of the hot path
   (``src/stream/results_coverage.py``, ``src/write/formatting.py``,
   ``src/write/thresholds.py``, ``embedders/yamnet/params.py``).  The TF-backed
   modules cannot be imported here (``ModuleNotFoundError: tensorflow``), so the
   hot path itself has no reference-generated vectors: parity for it is pinned
   only by the extracted constants ("Oracle").
3. ``tests/golden/graph_facts.json`` — the STRUCTURE of the three SavedModel graphs the
   reference ships, decoded from ``saved_model.pb`` (op types, attributes, scalar
   constants of the inlined inference function): strides / ``SAME`` / ``NHWC`` of the 27
   convolutions, ``epsilon`` / ``is_training`` / ones-scale of the 27 batch norms, the
   ``[[0,0],[0,112]]`` pad in front of the RFFT, frame 400 / 160 / 512, 15600 / 15360 /
   7680 / 0.001, the periodic Hann op chain, ReLU (not ReLU6), Mean over [1, 2].
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from buzzdetect_amd import artifacts  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).astype("<f4").tobytes()).hexdigest()


def product_data(ref: str, out: str) -> None:
    os.makedirs(out, exist_ok=True)
    # -- mel matrices (features.py:50-55 baked as graph Consts) -----------------
    k2 = artifacts.extract_mel_matrix(os.path.join(
        ref, "embedders/yamnet_k2/models/yamnet_wholehop/saved_model.pb"))
    k2h = artifacts.extract_mel_matrix(os.path.join(
        ref, "embedders/yamnet_k2/models/yamnet_halfhop/saved_model.pb"))
    assert np.array_equal(k2, k2h), "wholehop/halfhop mel constants differ"
    k3 = artifacts.extract_mel_matrix(os.path.join(ref, "embedders/yamnet/saved_model.pb"))
    k2.astype("<f4").tofile(os.path.join(out, "mel_yamnet_k2_257x64.f32"))
    k3.astype("<f4").tofile(os.path.join(out, "mel_yamnet_keras3_257x64.f32"))

    # -- dense head (models/model_general_v3/model.py:29) -----------------------
    hdir = os.path.join(ref, "models/model_general_v3")
    idx_p, dat_p = artifacts.bundle_paths(hdir)
    hidx = artifacts.read_bundle_index(idx_p)
    kern = artifacts.read_bundle_tensor(dat_p, hidx["layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"])
    bias = artifacts.read_bundle_tensor(dat_p, hidx["layer_with_weights-0/bias/.ATTRIBUTES/VARIABLE_VALUE"])
    assert kern.shape == (1024, 13) and bias.shape == (13,)
    kern.astype("<f4").tofile(os.path.join(out, "head_model_general_v3_kernel_1024x13.f32"))
    bias.astype("<f4").tofile(os.path.join(out, "head_model_general_v3_bias_13.f32"))
    shutil.copyfile(os.path.join(hdir, "config_model.json"), os.path.join(out, "config_model_general_v3.json"))
    shutil.copyfile(os.path.join(hdir, "tests/metrics.csv"), os.path.join(out, "metrics_model_general_v3.csv"))

    # -- embedder tensor manifest ------------------------------------------------
    eidx = artifacts.read_bundle_index(os.path.join(
        ref, "embedders/yamnet_k2/models/yamnet_wholehop/variables/variables.index"))
    for other in ("embedders/yamnet_k2/models/yamnet_halfhop", "embedders/yamnet"):
        o = artifacts.read_bundle_index(os.path.join(ref, other, "variables/variables.index"))
        for k, e in eidx.items():
            assert (o[k].shape, o[k].offset, o[k].size) == (e.shape, e.offset, e.size), (other, k)
    tensors = [
        {"name": e.name.replace("/.ATTRIBUTES/VARIABLE_VALUE", ""), "shape": list(e.shape),
         "offset": e.offset, "size": e.size}
        for e in sorted(eidx.values(), key=lambda e: e.offset) if e.dtype == 1
    ]
    manifest = {
        "source": "embedders/yamnet_k2/models/yamnet_wholehop/variables/variables.index "
                  "(identical table in yamnet_halfhop and embedders/yamnet)",
        "payload_bytes": sum(t["size"] for t in tensors),
        "tensors": tensors,
        "sha256": {
            "mel_yamnet_k2_257x64.f32": sha(k2),
            "mel_yamnet_keras3_257x64.f32": sha(k3),
            "head_model_general_v3_kernel_1024x13.f32": sha(kern),
            "head_model_general_v3_bias_13.f32": sha(bias),
        },
    }
    with open(os.path.join(out, "embedder_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"product data -> {out}: {len(tensors)} embedder tensors, payload {manifest['payload_bytes']} B")


def golden_vectors(ref: str, out: str) -> None:
    os.makedirs(out, exist_ok=True)
    cwd = os.getcwd()
    sys.path.insert(0, ref)
    os.chdir(ref)  # the reference resolves models/ relative to cwd (src/config.py:23-26)
    try:
        from src.stream.results_coverage import gaps_to_chunklist, get_gaps, melt_coverage, smooth_gaps
        from src.write.formatting import add_time, format_activations, format_detections
        from src.write.thresholds import calculate_threshold
        from embedders.yamnet.params import Params
        import pandas as pd

        g: dict = {}

        # chunk framing (src/stream/results_coverage.py:59-70)
        cases = [([(0, 600.5)], 199.68), ([(0, 3600.0)], 199.68), ([(0, 3600.0)], 600.0),
                 ([(0, 86400.0)], 600.0), ([(0, 3.82)], 199.68), ([(0, 0.5)], 0.96),
                 ([(12.48, 100.0), (250.0, 251.0)], 19.2), ([(0, 1000.32)], 1000.32),
                 ([(0, 4400.0)], 199.68)]
        g["gaps_to_chunklist"] = [
            {"gaps": [list(x) for x in gaps], "chunklength": cl,
             "chunks": [[float(a), float(b)] for a, b in gaps_to_chunklist(gaps, cl)]}
            for gaps, cl in cases]

        # coverage -> gaps (results_coverage.py:4-56), used on resume
        cov_cases = [
            {"starts": [0.0, 0.96, 1.92, 10.56, 11.52], "framelength": 0.96, "range": [0, 30.0], "tol": None},
            {"starts": [5.76, 6.72, 0.0, 0.96], "framelength": 0.96, "range": [0, 7.68], "tol": 0.5},
            {"starts": [0.0, 0.48, 0.96, 3.0], "framelength": 0.96, "range": [0, 20.0], "tol": None},
        ]
        for c in cov_cases:
            df = pd.DataFrame({"start": c["starts"]})
            cov = melt_coverage(df, c["framelength"])
            gaps = get_gaps(tuple(c["range"]), cov)
            sm = smooth_gaps(gaps, tuple(c["range"]), c["framelength"], c["tol"])
            c["coverage"] = [[float(a), float(b)] for a, b in cov]
            c["gaps"] = [[float(a), float(b)] for a, b in gaps]
            c["smoothed"] = [[float(a), float(b)] for a, b in sm]
        g["coverage"] = cov_cases

        # window timestamps (src/write/formatting.py:5-17)
        g["add_time"] = []
        for n, t0, hop in [(208, 199.68, 0.96), (415, 0.48, 0.48), (208, 0, 0.96), (625, 85800.0, 0.96),
                           (4, 4193.28, 0.96)]:
            df = add_time(pd.DataFrame({"x": np.zeros(n)}), t0, hop, 2)
            g["add_time"].append({"n": n, "time_start": t0, "framehop_s": hop, "digits": 2,
                                  "start": [float(v) for v in df["start"]]})

        # result formatting (src/write/formatting.py:20-50)
        with open(os.path.join(ref, "models/model_general_v3/config_model.json")) as f:
            classes = json.load(f)["classes"]
        rng = np.random.default_rng(7)
        res = (rng.standard_normal((5, 13)) * 2.0).astype(np.float32)
        res[0, 8] = -1.2049999
        res[1, 8] = -1.205
        res[2, 8] = 0.125
        fa = format_activations(res, classes, 0.96, 2, time_start=199.68, classes_keep="all", digits_results=2)
        fk = format_activations(res, classes, 0.96, 2, time_start=0,
                                classes_keep=["ins_buzz", "ambient_rain", "mech_auto"], digits_results=2)
        fd = format_detections(res, -1.205, classes, 0.96, 2, 199.68)
        g["formatting"] = {
            "results_f32": [[float(v) for v in r] for r in res], "classes": classes,
            "activations_all_csv": fa.to_csv(index=False),
            "activations_keep": ["ins_buzz", "ambient_rain", "mech_auto"],
            "activations_keep_csv": fk.to_csv(index=False),
            "detections_threshold": -1.205, "detections_csv": fd.to_csv(index=False),
        }

        # thresholds (src/write/thresholds.py:29-41)
        g["calculate_threshold"] = [
            {"precision": p, "threshold": float(calculate_threshold("model_general_v3", p))}
            for p in (0.90, 0.95, 0.99, 0.80, 0.975)]

        # manifest reconciliation (src/pipeline/manifest.py:13-85)
        from src.pipeline.manifest import build_manifest, diff_manifests
        m_act = build_manifest("model_general_v3", 1.0, None, ["ins_buzz", "ambient_rain"])
        m_act2 = build_manifest("model_general_v3", 1.0, None, ["ambient_rain", "ins_buzz"])
        m_act3 = build_manifest("model_general_v3", 0.5, None, ["ins_buzz", "mech_auto"])
        m_det = build_manifest("model_general_v3", 1.0, 0.95, ["ins_buzz"])
        g["manifest"] = {
            "activations": m_act, "activations_reordered": m_act2, "other": m_act3, "detections": m_det,
            "diff_same": diff_manifests(m_act, m_act2), "diff_other": diff_manifests(m_act, m_act3),
            "diff_mode": diff_manifests(m_act, m_det),
        }

        # front-end constants (embedders/yamnet/params.py:26-51)
        p = Params()
        g["params"] = {k: getattr(p, k) for k in (
            "sample_rate", "stft_window_seconds", "stft_hop_seconds", "mel_bands", "mel_min_hz",
            "mel_max_hz", "log_offset", "patch_window_seconds", "patch_hop_seconds", "num_classes",
            "conv_padding", "batchnorm_center", "batchnorm_scale", "batchnorm_epsilon")}
        g["params"]["patch_frames"] = p.patch_frames
        g["params"]["patch_bands"] = p.patch_bands
    finally:
        os.chdir(cwd)
        sys.path.remove(ref)

    with open(os.path.join(out, "reference_helpers.json"), "w") as f:
        json.dump(g, f, indent=1)
    print(f"golden vectors -> {out}/reference_helpers.json")


# --------------------------------------------------------------------------- graph structure
GRAPHS = {
    "yamnet_k2_wholehop": "embedders/yamnet_k2/models/yamnet_wholehop/saved_model.pb",
    "yamnet_k2_halfhop": "embedders/yamnet_k2/models/yamnet_halfhop/saved_model.pb",
    "yamnet_keras3": "embedders/yamnet/saved_model.pb",
}


def _scalar(n):
    return None if n.const is None or n.const.size != 1 else n.const.reshape(-1)[0].item()


def graph_structure(pb_path: str) -> dict:
    """What the SavedModel graph itself says about the hot path (everything TensorFlow decides implicitly and the
    oracle restates): op types and attributes of the inlined inference function, the constants it is called with."""
    nodes = artifacts.saved_model_nodes(pb_path)
    by_fn: dict = {}
    for n in nodes:
        by_fn.setdefault(n.function, []).append(n)

    def is_inference(v):
        ops = [n.op for n in v]
        return (ops.count("RFFT") == 1 and ops.count("Conv2D") == 14 and ops.count("DepthwiseConv2dNative") == 13
                and "AssignVariableOp" not in ops)
    cands = sorted(f for f, v in by_fn.items() if f and is_inference(v))
    assert cands, f"{pb_path}: no inlined inference function found"
    per_fn = []
    for fn in cands:
        v = by_fn[fn]
        byname = {n.name: n for n in v}

        def const_of(ref: str):
            return byname.get(ref.split(":")[0])

        facts: dict = {}
        short = lambda name: name.split("/", 1)[1] if "/" in name else name      # drop the model-name prefix

        # every scalar / small Const of the front end, by node name
        fe = {}
        keep = ("/frame_length", "/frame_step", "/fft_length", "rfft/Pad/paddings", "hann_window/periodic",
                "hann_window/Const", "hann_window/mul_2/x", "hann_window/sub_2/x", "hann_window/mod/y", "/frame/axis")
        for n in v:
            if n.op == "Const" and n.const is not None and n.const.size <= 8 and (
                    n.name.endswith(keep) or "reshape/Reshape/shape/" in n.name):
                fe[short(n.name)] = n.const.tolist()
        facts["frontend_consts"] = fe
        # front-end op chain (everything outside the tf.signal.frame index arithmetic), in graph order
        chain = []
        for n in v:
            if n.op in ("Const", "Identity", "NoOp", "ReadVariableOp") or "/frame/" in n.name or "hann_window" in n.name:
                continue
            if n.name.startswith(v[0].name.split("/")[0] + "/tf.") or "/tf." in n.name:
                attrs = {k: a for k, a in n.attrs.items()
                         if k in ("transpose_a", "transpose_b", "DstT", "SrcT", "shrink_axis_mask", "N")}
                chain.append([short(n.name), n.op, attrs])
        facts["frontend_ops"] = chain
        facts["hann_window_ops"] = [[short(n.name), n.op] for n in v if "hann_window" in n.name and n.op != "Const"]

        convs, bns, relus = [], [], []
        for n in v:
            if n.op in ("Conv2D", "DepthwiseConv2dNative"):
                convs.append({"name": short(n.name), "op": n.op,
                              **{k: n.attrs.get(k) for k in ("strides", "padding", "data_format", "dilations",
                                                             "explicit_paddings")}})
            elif n.op == "FusedBatchNormV3":
                scale = const_of(n.inputs[1])
                bns.append({"name": short(n.name), "epsilon": n.attrs.get("epsilon"),
                            "is_training": n.attrs.get("is_training"), "data_format": n.attrs.get("data_format"),
                            "exponential_avg_factor": n.attrs.get("exponential_avg_factor"),
                            "scale_is_const": scale is not None and scale.op == "Const",
                            "scale_all_ones": bool(scale is not None and scale.const is not None
                                                   and np.all(scale.const == 1.0)),
                            "channels": int(scale.const.size) if scale is not None and scale.const is not None else None})
            elif n.op in ("Relu", "Relu6", "Sigmoid", "Softmax"):
                relus.append([short(n.name), n.op])
        facts["convs"], facts["batchnorms"], facts["activations"] = convs, bns, relus
        mean = [n for n in v if n.op == "Mean"]
        assert len(mean) == 1
        facts["pool"] = {"name": short(mean[0].name), "op": "Mean", "keep_dims": mean[0].attrs.get("keep_dims"),
                         "reduction_indices": const_of(mean[0].inputs[1]).const.tolist()}
        per_fn.append(facts)
    for other in per_fn[1:]:
        assert other == per_fn[0], f"{pb_path}: inference functions disagree"
    out = per_fn[0]
    out["inference_functions"] = len(cands)
    # constants of the main graph: the arguments the serving call passes to the function (pad_waveform's 15600 / hop,
    # the log offset; the [257,64] mel matrix is extracted separately)
    out["main_graph_scalar_consts"] = sorted(
        ([str(n.const.dtype), _scalar(n)] for n in by_fn.get("", [])
         if n.op == "Const" and n.const is not None and n.const.size == 1), key=lambda kv: (kv[0], kv[1]))
    return out


def graph_facts(ref: str, out: str) -> None:
    os.makedirs(out, exist_ok=True)
    facts = {name: graph_structure(os.path.join(ref, rel)) for name, rel in GRAPHS.items()}
    facts["_source"] = {name: rel + " [decoded: function library of the SavedModel, no TensorFlow]" for name, rel in GRAPHS.items()}
    with open(os.path.join(out, "graph_facts.json"), "w") as f:
        json.dump(facts, f, indent=1, sort_keys=True)
    print(f"graph structure -> {out}/graph_facts.json")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    product_data(args.reference, os.path.join(REPO, "buzzdetect_amd", "data"))
    golden_vectors(args.reference, os.path.join(REPO, "tests", "golden"))
    graph_facts(args.reference, os.path.join(REPO, "tests", "golden"))


if __name__ == "__main__":
    main()
