"""`model_general_v3` on the MI355X engine (reference: models/model_general_v3/model.py:6-31).

The reference chains ``embedder.embed`` and a Dense(1024 -> 13) SavedModel; the engine runs the
same chain in one enqueue (front end, 14 conv layers, pooling, dense head) and returns raw logits.
"""
from src.inference.models import BaseModel


class ModelGeneralV3(BaseModel):
    modelname = "model_general_v3"
    embeddername = 'yamnet_k2'
    digits_results = 2

    def initialize(self):
        from buzzdetect_amd.engine import HipEngine
        engine_embedder = getattr(self.embedder, "engine_embedder", None)
        if engine_embedder is None:
            raise RuntimeError(f"embedder plugin '{self.embeddername}' is not backed by the HIP engine")
        # the embedder's weights come from beside the EMBEDDER plugin, where the reference's embedder.initialize() loads them
        self.model = HipEngine(embeddername=engine_embedder, modelname=self.modelname,
                               variables_candidates=self.embedder.variables_candidates())
        self.embedder.attach(self.model)   # one set of weights serves embed() and predict()

    def predict(self, audiosamples):
        """1-D float32 audio at 16 kHz -> [n_windows, 13] logits (device-resident, has .numpy())."""
        return self.model.predict(audiosamples, self.embedder.framehop_s)
