#!/usr/bin/env python3
"""sha256 of the logits and embeddings of a fixed synthetic input, all three arithmetic modes: run before and after a change
that must not move a bit (kernel rewrites of the same arithmetic).

    python tools/logits_digest.py
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import numpy as np

from buzzdetect_amd.engine import HipEngine
from oracle import yamnet_oracle as O


def main():
    eng = HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=0)
    x = O.synthetic_audio(15360 * 299 + 15600, seed=7)
    for mode in ("f16x3", "f16", "f32"):
        eng.set_pointwise_mode(mode)
        logits = np.ascontiguousarray(eng.predict(x, 0.96).numpy())
        emb = np.ascontiguousarray(eng.embed(x, 0.96).numpy())
        half = np.ascontiguousarray(eng.predict(x, 0.48).numpy())
        print(mode, logits.shape, hashlib.sha256(logits.tobytes()).hexdigest()[:16], hashlib.sha256(emb.tobytes()).hexdigest()[:16],
              hashlib.sha256(half.tobytes()).hexdigest()[:16], flush=True)


if __name__ == "__main__":
    main()
