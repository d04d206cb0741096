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

That file is NOT part of the reference checkout (``.MISSING_LARGE_BLOBS``).  ``load_embedder_blob`` looks for it
where the reference loads its SavedModel - next to the embedder plugin (``embedders/yamnet_k2/embedder.py:14-24``:
``models/yamnet_wholehop`` / ``models/yamnet_halfhop``; ``embedders/yamnet/embedder.py:25-31``: beside ``embedder.py``) -
then in ``BUZZDETECT_YAMNET_VARIABLES``, and FAILS when there is none, as the reference does when its model directory is
missing.  ``synthetic_embedder_blob`` (seeded stand-ins in the same layout) is used only on an explicit opt-in -
``synthetic=True`` or ``BUZZDETECT_SYNTHETIC_WEIGHTS=1``, which the tests, ``bench.py`` and ``smoke()`` set - and says
so with a WARNING on logger ``buzzdetect``: result files written with it are noise in the reference's format.
"""
from __future__ import annotations

import json
import logging
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


VARIABLES_DATA = "variables.data-00000-of-00001"
VARIABLES_ENV = "BUZZDETECT_YAMNET_VARIABLES"
SYNTHETIC_ENV = "BUZZDETECT_SYNTHETIC_WEIGHTS"
PACKAGED_OVERLAY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")
_log = logging.getLogger("buzzdetect")


def plugin_variables(plugin_dir: str, engine_embedder: str, framehop_prop=None) -> List[str]:
    """Where the embedder weights sit relative to an embedder plugin's directory, in the reference's own layout.

    ``yamnet_k2`` (embedders/yamnet_k2/embedder.py:14-24): ``models/yamnet_wholehop`` for framehop_prop 1,
    ``models/yamnet_halfhop`` for 0.5 (the two SavedModels hold the same variables; without a hop both are tried);
    ``yamnet`` (embedders/yamnet/embedder.py:25-31 loads ``yamnet.keras`` beside the plugin, an HDF5 archive this
    package cannot read without h5py): the TensorBundle of the same weights the checkout keeps beside it,
    ``variables/variables.data-00000-of-00001``."""
    if engine_embedder == "yamnet_k2":
        subs = {1: ("yamnet_wholehop",), 0.5: ("yamnet_halfhop",)}.get(framehop_prop, ("yamnet_wholehop", "yamnet_halfhop"))
        return [os.path.join(plugin_dir, "models", sub, "variables", VARIABLES_DATA) for sub in subs]
    return [os.path.join(plugin_dir, "variables", VARIABLES_DATA)]


def default_candidates(engine_embedder: str, framehop_prop=None) -> List[str]:
    """Plugin directories a bare ``HipEngine()`` / ``analyze()`` looks in: ``embedders/<name>`` under the working directory
    (how buzzdetect resolves plugins, src/config.py:23), then the overlay shipped inside this package."""
    out: List[str] = []
    for root in (os.getcwd(), PACKAGED_OVERLAY):
        for c in plugin_variables(os.path.join(root, "embedders", engine_embedder), engine_embedder, framehop_prop):
            if c not in out:
                out.append(c)
    return out


RAW_TAIL_MAX = 1 << 16      # bytes a data shard may carry behind the 54 tensors (the saved object graph)


def read_variables(path: str) -> np.ndarray:
    """The embedder blob from a ``variables.data-00000-of-00001``.  With its ``variables.index`` beside it every tensor is
    looked up by name and checked (shape, float32) and the blob is assembled in the order the C ABI expects, whatever order
    the file keeps; without one the file must BE that layout."""
    index_path = os.path.join(os.path.dirname(path), "variables.index")
    if os.path.exists(index_path):
        from . import artifacts
        index = artifacts.read_bundle_index(index_path)
        blob = np.empty(EMBEDDER_BLOB_FLOATS, dtype=np.float32)
        for name, shape, off in expected_table():
            key = name + "/.ATTRIBUTES/VARIABLE_VALUE"
            entry = index.get(key) or index.get(name)
            if entry is None:
                raise ValueError(f"{index_path}: no tensor {name}")
            if tuple(entry.shape) != tuple(shape) or entry.dtype != 1:
                raise ValueError(f"{index_path}: {name} is {entry.shape} dtype {entry.dtype}, expected float32 {shape}")
            blob[off:off + entry.count] = artifacts.read_bundle_tensor(path, entry).reshape(-1)
        return blob
    # No index: the file is taken to BE the blob in the C ABI's order.  That is an assumption about bytes nobody named, so it
    # is said out loud and checked as far as the layout allows: the size is the payload plus at most a small object-graph
    # tail (a TensorBundle data shard of this model is 12 869 376 bytes + < 64 KiB), every value is finite, and the slices
    # that must be BatchNorm moving variances are all positive - an unrelated or re-ordered bundle fails that at once.
    size = os.path.getsize(path)
    need = EMBEDDER_BLOB_FLOATS * 4
    if size < need or size > need + RAW_TAIL_MAX:
        raise ValueError(f"{path}: {size} bytes without a variables.index beside it; a raw embedder payload is {need} bytes "
                         f"(+ at most {RAW_TAIL_MAX} of object graph)")
    raw = np.fromfile(path, dtype="<f4", count=EMBEDDER_BLOB_FLOATS).astype(np.float32)
    if not np.isfinite(raw).all():
        raise ValueError(f"{path}: non-finite values where the embedder weights should be (no variables.index to look tensors up by)")
    for name, shape, off in expected_table():
        if name.endswith("moving_variance"):
            n = int(np.prod(shape))
            if not (raw[off:off + n] > 0).all():
                raise ValueError(f"{path}: the slice that should be {name} is not all positive - this file is not the embedder "
                                 f"payload in blob order (put its variables.index beside it)")
    _log.warning("embedder weights: %s has no variables.index beside it; read as the raw payload in blob order "
                 "(size, finiteness and BatchNorm-variance slices checked)", path)
    return raw


def synthetic_allowed(synthetic: Optional[bool] = None) -> bool:
    if synthetic is not None:
        return bool(synthetic)
    return os.environ.get(SYNTHETIC_ENV, "").strip().lower() in ("1", "true", "yes")


def load_embedder_blob(path: Optional[str] = None, candidates=(), synthetic: Optional[bool] = None) -> np.ndarray:
    """The embedder weights, in this order: ``path`` (must exist); the first of ``candidates`` that exists (the plugin's
    own directory, ``plugin_variables``); ``$BUZZDETECT_YAMNET_VARIABLES``; seeded synthetic weights ONLY when opted in
    (``synthetic=True`` or ``BUZZDETECT_SYNTHETIC_WEIGHTS=1``), with a WARNING.  Otherwise ``FileNotFoundError`` naming
    every place looked at - never a silent fallback."""
    if path:
        return read_variables(path)
    tried = []
    for c in candidates:
        tried.append(c)
        if os.path.exists(c):
            _log.info(f"embedder weights: {c}")
            return read_variables(c)
    env = os.environ.get(VARIABLES_ENV)
    if env:
        _log.info(f"embedder weights: {env} (${VARIABLES_ENV})")
        return read_variables(env)
    if synthetic_allowed(synthetic):
        _log.warning("SYNTHETIC embedder weights (seeded random stand-ins, seed %d): no %s was found and synthetic weights "
                     "were asked for (synthetic=True or %s=1).  Results are NOT YAMNet's - fit for tests and throughput "
                     "measurements only.", SYNTHETIC_SEED, VARIABLES_DATA, SYNTHETIC_ENV)
        return synthetic_embedder_blob()
    raise FileNotFoundError(
        "YAMNet embedder weights not found.  Looked for " + ", ".join(tried or ["(no plugin directory given)"]) +
        f"; ${VARIABLES_ENV} is not set.  Put the checkout's {VARIABLES_DATA} there (the reference loads its SavedModel from "
        f"the same directory), point ${VARIABLES_ENV} at it, or opt in to seeded synthetic weights with {SYNTHETIC_ENV}=1.")


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
