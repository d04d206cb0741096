"""Where the embedder weights come from: beside the plugin, where the reference loads its SavedModel
(embedders/yamnet_k2/embedder.py:14-24, embedders/yamnet/embedder.py:25-31), then $BUZZDETECT_YAMNET_VARIABLES; seeded
synthetic weights only on an explicit opt-in and with a WARNING; no source -> FileNotFoundError, never a silent fallback."""
import logging
import os
import shutil
import sys

import numpy as np
import pytest

from buzzdetect_amd import weights as W
from conftest import DROPIN
from test_artifacts import write_bundle


@pytest.fixture()
def no_sources(monkeypatch):
    monkeypatch.delenv(W.SYNTHETIC_ENV, raising=False)
    monkeypatch.delenv(W.VARIABLES_ENV, raising=False)


def _overlay(tmp_path, monkeypatch):
    """A private copy of the overlay as working directory, the way buzzdetect resolves plugins (src/config.py)."""
    root = tmp_path / "checkout"
    shutil.copytree(DROPIN, root, ignore=shutil.ignore_patterns("__pycache__"))
    monkeypatch.chdir(root)
    monkeypatch.syspath_prepend(str(root))
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    return root


def _bundle_tensors(blob):
    return {name + "/.ATTRIBUTES/VARIABLE_VALUE": blob[off:off + int(np.prod(shape))].reshape(shape)
            for name, shape, off in W.expected_table()}


def test_no_source_raises_and_names_the_places(no_sources, tmp_path):
    cands = W.plugin_variables(str(tmp_path / "embedders" / "yamnet_k2"), "yamnet_k2", 1)
    assert cands == [str(tmp_path / "embedders" / "yamnet_k2" / "models" / "yamnet_wholehop" / "variables" / W.VARIABLES_DATA)]
    with pytest.raises(FileNotFoundError) as e:
        W.load_embedder_blob(None, cands)
    assert cands[0] in str(e.value) and W.VARIABLES_ENV in str(e.value) and W.SYNTHETIC_ENV in str(e.value)
    assert W.plugin_variables("/p", "yamnet_k2", 0.5) == ["/p/models/yamnet_halfhop/variables/" + W.VARIABLES_DATA]
    assert W.plugin_variables("/p", "yamnet") == ["/p/variables/" + W.VARIABLES_DATA]


def test_synthetic_needs_the_opt_in_and_warns(no_sources, monkeypatch, caplog):
    with pytest.raises(FileNotFoundError):
        W.load_embedder_blob()
    with caplog.at_level(logging.WARNING, logger="buzzdetect"):
        blob = W.load_embedder_blob(synthetic=True)
    assert np.array_equal(blob, W.synthetic_embedder_blob())
    assert any("SYNTHETIC" in r.getMessage() and r.levelno == logging.WARNING for r in caplog.records)
    caplog.clear()
    monkeypatch.setenv(W.SYNTHETIC_ENV, "1")
    with caplog.at_level(logging.WARNING, logger="buzzdetect"):
        W.load_embedder_blob()
    assert any("SYNTHETIC" in r.getMessage() for r in caplog.records)
    monkeypatch.setenv(W.SYNTHETIC_ENV, "0")
    with pytest.raises(FileNotFoundError):
        W.load_embedder_blob()


@pytest.mark.parametrize("plugin,hop", [("yamnet_k2", 1), ("yamnet_k2", 0.5), ("yamnet", 0.3)])
def test_plugin_initialize_without_weights_raises(no_sources, tmp_path, monkeypatch, plugin, hop):
    """The overlay plugin, no weights anywhere, no opt-in: initialize() fails loudly (the reference: TFSMLayer on a missing
    directory, embedders/yamnet_k2/embedder.py:24) - on a box with or without a GPU, before any device is touched."""
    _overlay(tmp_path, monkeypatch)
    from src.inference.embedding import load_embedder
    emb = load_embedder(plugin, framehop_prop=hop, initialize=False)
    with pytest.raises(FileNotFoundError):
        emb.initialize()
    from src.inference.models import load_model
    model = load_model("model_general_v3", framehop_prop=1, initialize=False)
    with pytest.raises(FileNotFoundError):
        model.initialize()


@pytest.mark.parametrize("hop,sub", [(1, "yamnet_wholehop"), (0.5, "yamnet_halfhop")])
def test_bundle_beside_the_plugin_is_picked_up(no_sources, tmp_path, monkeypatch, hop, sub):
    """A TensorBundle (variables.index + variables.data-00000-of-00001) where the reference keeps yamnet_k2's SavedModel is
    what the plugin - and the model plugin that owns it - load; tensors are looked up by name, whatever order the file has."""
    root = _overlay(tmp_path, monkeypatch)
    rng = np.random.default_rng(11)
    blob = rng.standard_normal(W.EMBEDDER_BLOB_FLOATS).astype(np.float32)
    mdir = root / "embedders" / "yamnet_k2" / "models" / sub
    mdir.mkdir(parents=True)
    write_bundle(mdir, _bundle_tensors(blob))                        # sorted by name: NOT the C ABI's order
    vdir = mdir / "variables"
    from src.inference.models import load_model
    model = load_model("model_general_v3", framehop_prop=hop, initialize=False)
    cands = model.embedder.variables_candidates()
    assert [os.path.realpath(c) for c in cands] == [os.path.realpath(vdir / W.VARIABLES_DATA)]
    got = W.load_embedder_blob(None, cands)
    assert np.array_equal(got, blob)
    # a bare engine / analyze() started in that working directory finds the same file
    assert os.path.realpath(vdir / W.VARIABLES_DATA) in [os.path.realpath(c) for c in W.default_candidates("yamnet_k2")]
    # with the weights found, what is missing on a CPU box is the device - not the weights
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            model.initialize()


def test_raw_payload_and_env_variable(no_sources, tmp_path, monkeypatch):
    blob = np.arange(W.EMBEDDER_BLOB_FLOATS, dtype=np.float32)
    path = tmp_path / W.VARIABLES_DATA
    blob.tofile(path)
    assert np.array_equal(W.load_embedder_blob(str(path)), blob)
    monkeypatch.setenv(W.VARIABLES_ENV, str(path))
    assert np.array_equal(W.load_embedder_blob(None, [str(tmp_path / "nothing-here")]), blob)
    short = tmp_path / "short"
    blob[:100].tofile(short)
    with pytest.raises(ValueError):
        W.load_embedder_blob(str(short))


def test_raw_payload_without_an_index_is_checked_and_said_out_loud(no_sources, tmp_path, caplog):
    """without variables.index the file is ASSUMED to be the blob - any 12.9 MB file used to pass.  Now: a WARNING,
    and the file must have the payload's size (+ a small object-graph tail), finite values and positive BatchNorm variances."""
    good = W.synthetic_embedder_blob(seed=7)
    path = tmp_path / W.VARIABLES_DATA
    with open(path, "wb") as f:
        f.write(good.astype("<f4").tobytes() + b"\0" * 4096)                # payload + a tail, as a real shard has
    with caplog.at_level(logging.WARNING, logger="buzzdetect"):
        assert np.array_equal(W.read_variables(str(path)), good)
    assert any("no variables.index" in r.getMessage() and r.levelno == logging.WARNING for r in caplog.records)
    big = tmp_path / "unrelated.bin"
    with open(big, "wb") as f:
        f.write(good.astype("<f4").tobytes() + b"\0" * (W.RAW_TAIL_MAX + 4))
    with pytest.raises(ValueError, match="without a variables.index"):
        W.read_variables(str(big))
    name, shape, off = next(t for t in W.expected_table() if t[0].endswith("moving_variance"))
    bad = good.copy()
    bad[off + 3] = -0.25                                                    # e.g. a re-ordered bundle: a beta where a variance belongs
    bad.astype("<f4").tofile(tmp_path / "reordered.bin")
    with pytest.raises(ValueError, match="moving_variance"):
        W.read_variables(str(tmp_path / "reordered.bin"))
    bad = good.copy()
    bad[12345] = np.nan
    bad.astype("<f4").tofile(tmp_path / "nan.bin")
    with pytest.raises(ValueError, match="non-finite"):
        W.read_variables(str(tmp_path / "nan.bin"))


def test_index_with_a_wrong_shape_is_refused(no_sources, tmp_path):
    t = _bundle_tensors(np.zeros(W.EMBEDDER_BLOB_FLOATS, np.float32))
    key = "layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"
    t[key] = np.zeros((3, 3, 1, 16), np.float32)
    write_bundle(tmp_path, t)
    with pytest.raises(ValueError, match="layer_with_weights-0/kernel"):
        W.read_variables(str(tmp_path / "variables" / W.VARIABLES_DATA))


@pytest.mark.gpu
def test_engine_runs_on_the_weights_beside_the_plugin(no_sources, tmp_path, monkeypatch):
    """GPU: the plugin path end to end - model.initialize() with a bundle in the overlay gives the logits of an engine that
    was handed the same blob, and differs from the synthetic stand-ins."""
    root = _overlay(tmp_path, monkeypatch)
    blob = W.synthetic_embedder_blob(seed=4242)
    mdir = root / "embedders" / "yamnet_k2" / "models" / "yamnet_wholehop"
    mdir.mkdir(parents=True)
    write_bundle(mdir, _bundle_tensors(blob))
    from buzzdetect_amd.engine import HipEngine
    from src.inference.models import load_model
    model = load_model("model_general_v3", framehop_prop=1, initialize=True)
    x = (0.1 * np.random.default_rng(3).standard_normal(15360 * 3)).astype(np.float32)
    got = model.predict(x).numpy()
    ref_eng = HipEngine(embedder_blob=blob)
    here = HipEngine(synthetic_weights=True)          # the opt-in is a LAST resort: in this working directory the bundle still wins
    syn_eng = HipEngine(embedder_blob=W.synthetic_embedder_blob())
    try:
        assert np.array_equal(got, ref_eng.predict(x, 0.96).numpy())
        assert np.array_equal(got, here.predict(x, 0.96).numpy())
        assert np.abs(got - syn_eng.predict(x, 0.96).numpy()).max() > 1e-3
    finally:
        ref_eng.close()
        syn_eng.close()
        here.close()
        model.model.close()
