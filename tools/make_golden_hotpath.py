#!/usr/bin/env python3
"""Write tests/golden/hotpath_oracle_f64.npz: seeded inputs -> float64 oracle outputs for the hot path.

These vectors come from THIS repository's oracle (oracle/yamnet_oracle.py), not from the reference:
the reference cannot produce hot-path outputs (no TensorFlow here, embedder weights not in the
checkout).  They freeze the oracle so the GPU box and later rounds test against identical numbers.
"""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from buzzdetect_amd import weights as W  # noqa: E402
from oracle import yamnet_oracle as O  # noqa: E402


def main():
    blob, mel, head = W.synthetic_embedder_blob(), W.load_mel("yamnet_k2"), W.load_head()
    seed, n = 1234, 15360 * 6 + 4321
    x = O.synthetic_audio(n, seed)
    lm = O.log_mel(O.pad_waveform(x, 15360), mel, np.float64)
    emb = O.embed(x, blob, mel, 15360, 96, np.float64)
    whole = O.dense_head(emb, head.kernel, head.bias, np.float64)
    half = O.predict(x, blob, mel, head.kernel, head.bias, 7680, 48, np.float64)
    out = os.path.join(REPO, "tests", "golden", "hotpath_oracle_f64.npz")
    np.savez_compressed(out, seed=seed, n_samples=n, audio_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                        blob_sha256=hashlib.sha256(blob.tobytes()).hexdigest(),
                        logmel_rows=lm[[0, 1, 95, 96, 300, lm.shape[0] - 1]], logmel_row_index=[0, 1, 95, 96, 300, lm.shape[0] - 1],
                        embeddings_whole=emb, logits_whole=whole, logits_half=half)
    print(out, whole.shape, half.shape)


if __name__ == "__main__":
    main()
