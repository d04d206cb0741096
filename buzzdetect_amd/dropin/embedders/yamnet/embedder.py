"""`yamnet` (Keras-3, free hop) embedder on the MI355X engine
(reference: embedders/yamnet/embedder.py:14-44).

The reference sets ``patch_hop_seconds = framehop_s`` on the loaded model (embedder.py:30); the pad
hop ``int(hop_s * 16000)`` (features.py:99) and the patch step ``round(100 * hop_s)``
(features.py:70-71) are then derived independently, which is what the engine does.
"""
import os

from src.inference.embedding import BaseEmbedder


class EmbedderYamnet(BaseEmbedder):
    embeddername = "yamnet"
    framelength_s = 0.96
    digits_time = 2
    samplerate = 16000
    n_embeddings = 1024
    dtype_in = 'float32'

    engine_embedder = "yamnet"

    def variables_candidates(self):
        """The reference loads yamnet.keras beside this file (embedder.py:25-31); the engine reads the TensorBundle of the
        same weights the checkout keeps there, variables/variables.data-00000-of-00001."""
        from buzzdetect_amd import weights
        return weights.plugin_variables(os.path.dirname(os.path.realpath(__file__)), self.engine_embedder)

    def attach(self, engine):
        self.model = engine

    def initialize(self):
        from buzzdetect_amd.engine import HipEngine
        self.model = HipEngine(embeddername=self.engine_embedder, modelname=None,
                               variables_candidates=self.variables_candidates())

    def embed(self, audio):
        """1-D float32 audio at 16 kHz -> [n_windows, 1024] embeddings."""
        return self.model.embed(audio, self.framehop_s)
