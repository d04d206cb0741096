"""Model plugin ABC and loader (reference contract: src/inference/models.py:12-79).

A model owns an *uninitialised* embedder from construction on (callers read
``model.embedder.framelength_s`` etc. without ever initialising: src/analyze.py:105-110,
src/stream/worker.py:32-33) and its ``config_model.json`` as ``self.config`` (keys ``classes`` and
``digits_results`` are used at src/analyze.py:117,205,210).
"""
import json
import os
from abc import ABC, abstractmethod
from pathlib import Path

from src import config as cfg
from src.inference._discovery import first_subclass, load_plugin_module
from src.inference.embedding import BaseEmbedder, load_embedder


class BaseModel(ABC):
    modelname: str = None
    embeddername: str = None
    digits_results: int = None   # decimals kept in result files
    dtype_in: str = None

    def __init__(self, framehop_prop):
        self.model = None
        self.embedder: BaseEmbedder = load_embedder(
            embeddername=self.embeddername, framehop_prop=framehop_prop, initialize=False)
        with open(os.path.join(cfg.DIR_MODELS, self.modelname, 'config_model.json'), 'r') as f:
            self.config = json.load(f)

    @abstractmethod
    def initialize(self):
        """Create the compute engine (called inside the analyzer thread, src/inference/worker.py:78)."""

    @abstractmethod
    def predict(self, audiosamples):
        """1-D audio at ``embedder.samplerate`` -> ``[n_windows, n_classes]`` with a ``.numpy()``."""


def load_model(modelname: str, framehop_prop: float, initialize: bool):
    """Find ``models/<modelname>/model.py`` and build its BaseModel subclass."""
    if not (Path(cfg.DIR_MODELS) / modelname).exists():
        raise ValueError(f"model '{modelname}' not found in {cfg.DIR_MODELS}")

    module = load_plugin_module(cfg.DIR_MODELS, modelname, "model.py", "model")
    plugin = first_subclass(module, BaseModel)
    if plugin is None:
        raise ValueError(f"No BaseModel subclass found in {modelname}/model.py")

    model = plugin(framehop_prop=framehop_prop)
    if initialize:
        model.initialize()
    return model
