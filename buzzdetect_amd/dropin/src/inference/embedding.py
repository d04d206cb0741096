"""Embedder plugin ABC and loader — the drop-in boundary of the analyze hot path.

Same contract as the reference's src/inference/embedding.py:8-79: class attributes
``embeddername, samplerate, framelength_s, n_embeddings, digits_time, dtype_in``; instances carry
``framehop_prop``, ``framehop_s = framelength_s * framehop_prop`` and ``model`` (None until
``initialize()``); ``embed(samples)`` returns ``[n_windows, n_embeddings]``.
"""
from abc import ABC, abstractmethod
from pathlib import Path

from src import config as cfg
from src.inference._discovery import first_subclass, load_plugin_module


class BaseEmbedder(ABC):
    embeddername: str = None
    samplerate: int = None        # Hz
    framelength_s: float = None   # seconds of audio per embedding frame
    n_embeddings: int = None
    digits_time: int = None       # decimals for timestamps (matches framelength_s)
    dtype_in: str = None

    def __init__(self, framehop_prop):
        self.framehop_prop = framehop_prop
        self.framehop_s = self.framelength_s * framehop_prop
        self.model = None

    @abstractmethod
    def initialize(self):
        """Create the compute engine; construction alone must stay cheap (attribute access only)."""

    @abstractmethod
    def embed(self, samples):
        """1-D audio at ``samplerate`` -> ``[n_windows, n_embeddings]``."""


def load_embedder(embeddername: str, framehop_prop: float, initialize: bool):
    """Find ``embedders/<embeddername>/embedder.py`` and build its BaseEmbedder subclass."""
    if not (Path(cfg.DIR_EMBEDDERS) / embeddername).exists():
        raise ValueError(f"Embedder '{embeddername}' not found in {cfg.DIR_EMBEDDERS}")

    module = load_plugin_module(cfg.DIR_EMBEDDERS, embeddername, "embedder.py", "embedder")
    plugin = first_subclass(module, BaseEmbedder)
    if plugin is None:
        raise ValueError(f"No BaseEmbedder subclass found in {embeddername}/embedder.py")

    embedder = plugin(framehop_prop=framehop_prop)
    if initialize:
        embedder.initialize()
    return embedder
