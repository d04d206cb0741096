"""The plugin surface: same names, attributes and error behaviour as the reference's
src/inference/embedding.py:8-79, src/inference/models.py:12-79 and the three plugin files."""
import numpy as np
import pytest


def test_load_model_uninitialised_exposes_reference_attributes(dropin_cwd):
    from src.inference.models import BaseModel, load_model
    model = load_model("model_general_v3", framehop_prop=1.0, initialize=False)
    assert isinstance(model, BaseModel)
    assert (model.modelname, model.embeddername, model.digits_results) == ("model_general_v3", "yamnet_k2", 2)
    assert model.model is None
    e = model.embedder
    # attributes callers touch without initialising (src/analyze.py:105-110, src/stream/worker.py:32-33)
    assert (e.framelength_s, e.digits_time, e.samplerate, e.n_embeddings, e.dtype_in) == (0.96, 2, 16000, 1024, "float32")
    assert e.embeddername == "yamnet"            # YamnetK2 reports "yamnet" (embedders/yamnet_k2/embedder.py:7)
    assert e.framehop_prop == 1.0 and e.framehop_s == 0.96 and e.model is None
    assert model.config["classes"][8] == "ins_buzz" and len(model.config["classes"]) == 13
    assert model.config["digits_results"] == 2 and model.config["embeddername"] == "yamnet_k2"


def test_framehop_s_is_product_of_length_and_prop(dropin_cwd):
    from src.inference.embedding import load_embedder
    e = load_embedder("yamnet_k2", framehop_prop=0.5, initialize=False)
    assert e.framehop_s == 0.96 * 0.5 == 0.48
    e3 = load_embedder("yamnet", framehop_prop=0.3, initialize=False)
    assert type(e3).__name__ == "EmbedderYamnet" and e3.framehop_s == 0.96 * 0.3


def test_unknown_plugins_raise_value_error(dropin_cwd):
    from src.inference.embedding import load_embedder
    from src.inference.models import load_model
    with pytest.raises(ValueError, match="Embedder 'nope' not found in embedders"):
        load_embedder("nope", 1.0, False)
    with pytest.raises(ValueError, match="model 'nope' not found in models"):
        load_model("nope", 1.0, False)


def test_directory_without_subclass_raises(dropin_cwd, tmp_path, monkeypatch):
    from src import config as cfg
    from src.inference.embedding import load_embedder
    d = tmp_path / "embedders" / "empty"
    d.mkdir(parents=True)
    (d / "embedder.py").write_text("x = 1\n")
    monkeypatch.setattr(cfg, "DIR_EMBEDDERS", str(tmp_path / "embedders"))
    with pytest.raises(ValueError, match="No BaseEmbedder subclass found in empty/embedder.py"):
        load_embedder("empty", 1.0, False)


def test_yamnet_k2_rejects_other_hops(dropin_cwd):
    from src.inference.embedding import load_embedder
    e = load_embedder("yamnet_k2", framehop_prop=0.3, initialize=False)     # construction is allowed
    with pytest.raises(ValueError, match="For Keras 2 YAMNet, framehop_prop must be 1 or 0.5"):
        e.initialize()


def test_plugin_class_is_first_alphabetical_subclass(dropin_cwd, tmp_path, monkeypatch):
    from src import config as cfg
    from src.inference.embedding import load_embedder
    d = tmp_path / "embedders" / "two"
    d.mkdir(parents=True)
    (d / "embedder.py").write_text(
        "from src.inference.embedding import BaseEmbedder\n"
        "class Zed(BaseEmbedder):\n    framelength_s = 1.0\n    def initialize(self): pass\n    def embed(self, s): return 'z'\n"
        "class Alpha(BaseEmbedder):\n    framelength_s = 2.0\n    def initialize(self): pass\n    def embed(self, s): return 'a'\n")
    monkeypatch.setattr(cfg, "DIR_EMBEDDERS", str(tmp_path / "embedders"))
    assert type(load_embedder("two", 0.5, True)).__name__ == "Alpha"


def test_initialize_without_gpu_fails_loudly(dropin_cwd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the failure path is for CPU-only hosts")
    from src.inference.models import load_model
    model = load_model("model_general_v3", 1.0, initialize=False)
    with pytest.raises(Exception) as ei:
        model.initialize()
    assert "CPU" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_imports_the_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "buzzdetect_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f"{f} imports the oracle"


@pytest.mark.gpu
def test_plugin_predict_on_gpu_matches_oracle(dropin_cwd, weights_bundle):
    from oracle import yamnet_oracle as O
    from src.inference.models import load_model
    b = weights_bundle
    model = load_model("model_general_v3", framehop_prop=1.0, initialize=True)
    x = O.synthetic_audio(15360 * 4 + 300, seed=21)
    res = model.predict(x)
    got = res.numpy()                             # the one method the writer calls (src/write/worker.py:69)
    ref = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert got.shape == ref.shape == (5, 13) and got.dtype == np.float32
    assert np.abs(got - ref).max() < 1e-4
    emb = model.embedder.embed(x).numpy()         # embedder shares the model's engine
    assert emb.shape == (5, 1024)
    assert np.abs(emb - O.embed(x, b["blob"], b["mel"], 15360, 96, np.float64)).max() < 1e-4


@pytest.mark.gpu
def test_plugin_halfhop_and_free_hop_on_gpu(dropin_cwd, weights_bundle):
    from oracle import yamnet_oracle as O
    from src.inference.embedding import load_embedder
    from src.inference.models import load_model
    b = weights_bundle
    x = O.synthetic_audio(50000, seed=22)
    half = load_model("model_general_v3", framehop_prop=0.5, initialize=True).predict(x).numpy()
    ref = O.predict(x, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 7680, 48, np.float64)
    assert half.shape == ref.shape and np.abs(half - ref).max() < 1e-4
    e = load_embedder("yamnet", framehop_prop=0.3, initialize=True)   # Keras-3 embedder: any hop
    emb = e.embed(x).numpy()
    ref3 = O.embed(x, b["blob"], b["mel_keras3"], O.hop_samples(0.3), O.patch_step(0.3), np.float64)
    assert emb.shape == ref3.shape and np.abs(emb - ref3).max() < 1e-4
    # hops whose sample count tf.cast rounds UP through float32 (5376, not int(5375.999...) = 5375): one sample per hop
    # of padding decides the frame count of some chunk lengths: five hops after the first patch end on a multiple of 160
    for prop, hop in ((0.35, 5376), (0.7, 10752), (0.95, 14592)):
        e = load_embedder("yamnet", framehop_prop=prop, initialize=True)
        n = 15600 + 4 * hop + 1
        xs = O.synthetic_audio(n, seed=int(prop * 100))
        assert O.hop_samples(prop) == hop
        got = e.embed(xs).numpy()
        ref = O.embed(xs, b["blob"], b["mel_keras3"], hop, O.patch_step(prop), np.float64)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-4, prop
        assert O.num_frames(O.padded_length(n, hop)) != O.num_frames(O.padded_length(n, hop - 1)), "the case must tell the hops apart"


@pytest.mark.gpu
def test_worker_thread_contract_two_analyzers_on_one_device(dropin_cwd, weights_bundle):
    """src/inference/worker.py:21,78 and docs/source/tuning.rst:111: every analyzer THREAD constructs its own model
    (initialize=False), initialises it in-thread, and only ever calls predict from that thread; two of them share a GPU;
    the results are read on the WRITER thread.  Rows must be bit-equal to what a single thread computes, on every repetition."""
    import threading
    from oracle import yamnet_oracle as O
    from src.inference.models import load_model
    chunks = [O.synthetic_audio(15360 * (3 + i) + 100 * i, seed=40 + i) for i in range(6)]
    solo = load_model("model_general_v3", framehop_prop=1.0, initialize=True)
    want = [solo.predict(c).numpy().copy() for c in chunks]
    got, errors = {}, []
    start = threading.Barrier(2)

    # the analyzers only ENQUEUE (process_chunk: predict, then coordinator.put_write, src/inference/worker.py:71-74); ONE
    # writer thread calls .numpy() on the results later (src/write/worker.py:69), while the analyzers keep predicting
    import queue
    q_write: "queue.Queue" = queue.Queue()

    def worker(wid):
        try:
            model = load_model("model_general_v3", framehop_prop=1.0, initialize=False)      # WorkerInferer.__init__
            start.wait(30)
            model.initialize()                                                                # WorkerInferer.run, in-thread
            for rep in range(5):
                for i, c in enumerate(chunks):
                    q_write.put(((wid, rep, i), model.predict(c)))
        except BaseException as exc:              # noqa: BLE001
            errors.append(exc)
        finally:
            q_write.put(None)

    def writer():
        done = 0
        try:
            while done < 2:
                item = q_write.get(timeout=120)
                if item is None:
                    done += 1
                    continue
                key, res = item
                got[key] = res.numpy().copy()
        except BaseException as exc:              # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(2)] + [threading.Thread(target=writer)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    assert len(got) == 2 * 5 * len(chunks)
    for (wid, rep, i), rows in got.items():
        assert np.array_equal(rows, want[i]), (wid, rep, i)
