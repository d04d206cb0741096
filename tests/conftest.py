import json
import os
import sys

import numpy as np
import pytest

# The checkout has no YAMNet weights (.MISSING_LARGE_BLOBS): the suite runs on the seeded stand-ins, which the product only
# uses on this explicit opt-in (buzzdetect_amd/weights.py; tests/test_weights_source.py checks the refusal without it).
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, "buzzdetect_amd", "dropin")
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_helpers():
    with open(os.path.join(GOLDEN, "reference_helpers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def weights_bundle():
    from buzzdetect_amd import weights as W
    head = W.load_head()
    return {"blob": W.synthetic_embedder_blob(), "mel": W.load_mel("yamnet_k2"),
            "mel_keras3": W.load_mel("yamnet"), "head_kernel": head.kernel, "head_bias": head.bias,
            "classes": head.classes}


@pytest.fixture()
def dropin_cwd(monkeypatch):
    """Run with the overlay as working directory, the way buzzdetect resolves plugins (src/config.py)."""
    monkeypatch.chdir(DROPIN)
    monkeypatch.syspath_prepend(DROPIN)
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    yield DROPIN
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(name, None)


@pytest.fixture(scope="session")
def engine():
    """One HIP engine (yamnet_k2 + model_general_v3 head) for the GPU parity tests."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    from buzzdetect_amd.engine import HipEngine
    eng = HipEngine(embeddername="yamnet_k2", modelname="model_general_v3")
    yield eng
    eng.close()


@pytest.fixture(params=["f16x3", "f32"])
def engine_mode(engine, request):
    """The same engine with the 1x1 convolutions on split-f16 MFMA (default) or exact-f32 MFMA."""
    engine.set_pointwise_mode(request.param)
    yield engine
    engine.set_pointwise_mode("f16x3")
