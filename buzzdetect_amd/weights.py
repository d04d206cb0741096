"""Weights for the analyze hot path, kept in the reference's own on-disk layout.

The YAMNet embedder weights are one flat little-endian f32 blob with exactly the
byte layout of the reference TensorBundle payload
(``embedders/yamnet_k2/models/yamnet_wholehop/variables/variables.data-00000-of-00001``,
12 869 376 B; table in ``data/embedder_manifest.json``): conv1 kernel ``[3,3,1,32]``
then its BN ``beta / moving_mean / moving_variance``; then for each of the 13
separable layers (``embedders/yamnet/yamnet.py:77-93``) depthwise kernel
``[3,3,C,1]``, its BN triple, pointwise kernel ``[1,1,Cin,Cout]``, its BN triple.
The C-ABI (``include/buzzdetect_hip.h``) takes that blob unchanged and folds the
BatchNorms itself, so a user who owns the real file can hand it straight in.

That file is NOT part of the reference checkout (``.MISSING_LARGE_BLOBS``), so
``synthetic_embedder_blob`` provides seeded stand-ins in the same layout.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# (stride, filters) of the 14 layers, embedders/yamnet/yamnet.py:77-93
LAYER_DEFS: Tuple[Tuple[int, int], ...] = (
    (2, 32), (1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512),
    (1, 512), (1, 512), (1, 512), (1, 512), (1, 512), (2, 1024), (1, 1024),
)
EMBEDDER_BLOB_FLOATS = 3_217_344
BN_EPSILON = 1e-4  # embedders/yamnet/params.py:48
SYNTHETIC_SEED = 20260723


def manifest() -> dict:
    with open(os.path.join(DATA_DIR, "embedder_manifest.json")) as f:
        return json.load(f)


def blob_table() -> List[Tuple[str, Tuple[int, ...], int]]:
    """(name, shape, float offset) of every embedder tensor, in blob order."""
    return [(t["name"], tuple(t["shape"]), t["offset"] // 4) for t in manifest()["tensors"]]


def expected_table() -> List[Tuple[str, Tuple[int, ...], int]]:
    """The same table derived from LAYER_DEFS alone (what the C++ side assumes)."""
    out = []
    off = 0

    def add(name, shape):
        nonlocal off
        out.append((name, shape, off))
        n = 1
        for d in shape:
            n *= d
        off += n

    def bn(k, c):
        for part in ("beta", "moving_mean", "moving_variance"):
            add(f"layer_with_weights-{k}/{part}", (c,))

    add("layer_with_weights-0/kernel", (3, 3, 1, LAYER_DEFS[0][1]))
    bn(1, LAYER_DEFS[0][1])
    cin = LAYER_DEFS[0][1]
    k = 2
    for _, cout in LAYER_DEFS[1:]:
        add(f"layer_with_weights-{k}/depthwise_kernel", (3, 3, cin, 1))
        bn(k + 1, cin)
        add(f"layer_with_weights-{k + 2}/kernel", (1, 1, cin, cout))
        bn(k + 3, cout)
        k += 4
        cin = cout
    return out


def synthetic_embedder_blob(seed: int = SYNTHETIC_SEED) -> np.ndarray:
    """Seeded stand-in weights in the reference blob layout:
    He-normal kernels, BN mean~N(0,0.1), var~U(0.5,1.5), beta~N(0,0.1)."""
    rng = np.random.default_rng(seed)
    blob = np.empty(EMBEDDER_BLOB_FLOATS, dtype=np.float32)
    for name, shape, off in expected_table():
        n = int(np.prod(shape))
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            fan_in = shape[0] * shape[1] * shape[2]
            v = rng.standard_normal(n) * np.sqrt(2.0 / fan_in)
        elif leaf == "depthwise_kernel":
            v = rng.standard_normal(n) * np.sqrt(2.0 / (shape[0] * shape[1]))
        elif leaf == "moving_variance":
            v = rng.uniform(0.5, 1.5, n)
        else:  # beta, moving_mean
            v = rng.standard_normal(n) * 0.1
        blob[off:off + n] = v.astype(np.float32)
    return blob


def load_embedder_blob(path: Optional[str]) -> np.ndarray:
    """A real ``variables.data-00000-of-00001`` if the user has one, else synthetic."""
    if path is None:
        path = os.environ.get("BUZZDETECT_YAMNET_VARIABLES")
    if path:
        raw = np.fromfile(path, dtype="<f4", count=EMBEDDER_BLOB_FLOATS)
        if raw.size != EMBEDDER_BLOB_FLOATS:
            raise ValueError(f"{path}: expected at least {EMBEDDER_BLOB_FLOATS * 4} bytes of f32 payload")
        return raw.astype(np.float32)
    return synthetic_embedder_blob()


def split_blob(blob: np.ndarray) -> Dict[str, np.ndarray]:
    blob = np.asarray(blob)
    if blob.size != EMBEDDER_BLOB_FLOATS:
        raise ValueError(f"embedder blob has {blob.size} floats, expected {EMBEDDER_BLOB_FLOATS}")
    out = {}
    for name, shape, off in expected_table():
        n = int(np.prod(shape))
        out[name] = blob[off:off + n].reshape(shape)
    return out


def load_mel(embeddername: str = "yamnet_k2") -> np.ndarray:
    """The graph-baked ``[257,64]`` mel matrix (features.py:50-55).  The Keras-3
    ``yamnet`` SavedModel carries a float-noise variant of the yamnet_k2 one."""
    fn = {"yamnet_k2": "mel_yamnet_k2_257x64.f32", "yamnet": "mel_yamnet_keras3_257x64.f32"}[embeddername]
    return np.fromfile(os.path.join(DATA_DIR, fn), dtype="<f4").reshape(257, 64).astype(np.float32)


@dataclass
class HeadWeights:
    kernel: np.ndarray  # [1024, n_classes]
    bias: np.ndarray    # [n_classes]
    classes: List[str]


def load_head(modelname: str = "model_general_v3") -> HeadWeights:
    """Real dense-head weights (models/model_general_v3/variables, model.py:29)."""
    with open(os.path.join(DATA_DIR, f"config_{modelname}.json")) as f:
        cfg = json.load(f)
    n = len(cfg["classes"])
    k = np.fromfile(os.path.join(DATA_DIR, f"head_{modelname}_kernel_1024x{n}.f32"), dtype="<f4")
    b = np.fromfile(os.path.join(DATA_DIR, f"head_{modelname}_bias_{n}.f32"), dtype="<f4")
    return HeadWeights(k.reshape(1024, n).astype(np.float32), b.astype(np.float32), cfg["classes"])
