"""`yamnet_k2` embedder on the MI355X engine (reference: embedders/yamnet_k2/embedder.py:5-37).

The reference loads one of two Keras-2 SavedModels (whole hop / half hop) and returns the
``global_average_pooling2d`` output; here both hops run through the same HIP kernels with
hop = 15360 / 7680 samples and patch step 96 / 48 frames, the constants baked into those graphs.
"""
import os

from src.inference.embedding import BaseEmbedder


class YamnetK2(BaseEmbedder):
    embeddername = "yamnet"   # sic — the reference class reports "yamnet" (embedder.py:7)
    framelength_s = 0.96
    digits_time = 2
    samplerate = 16000
    n_embeddings = 1024
    dtype_in = 'float32'

    engine_embedder = "yamnet_k2"   # which graph constants (mel matrix) the engine loads

    def _check_hop(self):
        if not (self.framehop_prop == 1 or self.framehop_prop == 0.5):
            raise ValueError('For Keras 2 YAMNet, framehop_prop must be 1 or 0.5')

    def variables_candidates(self):
        """Where the reference loads this embedder's SavedModel (embedder.py:14-24): models/yamnet_wholehop or
        models/yamnet_halfhop beside this file; the engine reads variables/variables.data-00000-of-00001 there."""
        from buzzdetect_amd import weights
        return weights.plugin_variables(os.path.dirname(os.path.realpath(__file__)), self.engine_embedder, self.framehop_prop)

    def attach(self, engine):
        """Share an engine that already holds the embedder weights (used by model plugins)."""
        self._check_hop()
        self.model = engine

    def initialize(self):
        self._check_hop()
        from buzzdetect_amd.engine import HipEngine
        self.model = HipEngine(embeddername=self.engine_embedder, modelname=None,
                               variables_candidates=self.variables_candidates())

    def embed(self, audiosamples):
        """1-D float32 audio at 16 kHz -> [n_windows, 1024] embeddings (device-resident, has .numpy())."""
        return self.model.embed(audiosamples, self.framehop_s)
