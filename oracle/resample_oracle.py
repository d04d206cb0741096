"""CPU restatement of the streamer's downmix + resample stage — TEST INFRASTRUCTURE ONLY.

Reference: ``np.mean(samples, axis=1)`` then ``librosa.resample(y, orig_sr, target_sr)``
(src/stream/worker.py:116-128); librosa's default backend is soxr_hq, which is neither vendored nor
installed here, and its filter is not specified by the reference => **parity unpinned** for this stage.
The product defines the stage as the scipy.signal.resample_poly design (Kaiser 5.0 windowed sinc,
20*max(up,down)+1 taps); this module restates that definition directly:

    y[j] = sum_i mono[i] * h[j*down - i*up + half]
"""
from math import gcd

import numpy as np
import scipy.signal


def ratio(rate_in: int, rate_out: int):
    g = gcd(int(rate_in), int(rate_out))
    return rate_out // g, rate_in // g


def taps(up: int, down: int, dtype=np.float64):
    max_rate = max(up, down)
    half = 10 * max_rate
    if max_rate == 1:                      # equal rates: resample_poly returns the input; the filter is a delta
        h = np.zeros(2 * half + 1)
        h[half] = 1.0
        return h.astype(dtype), half
    h = scipy.signal.firwin(2 * half + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    return h.astype(dtype), half


def downmix(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    return x if x.ndim == 1 else np.mean(x, axis=1)          # float32 mean, as the reference computes it


def resample(x: np.ndarray, rate_in: int, rate_out: int = 16000, dtype=np.float64) -> np.ndarray:
    mono = downmix(x).astype(dtype)
    up, down = ratio(rate_in, rate_out)
    h, half = taps(up, down, dtype)
    n_in = mono.shape[0]
    n_out = -(-n_in * up // down)
    out = np.zeros(n_out, dtype=dtype)
    for j in range(n_out):
        c = j * down
        i0 = max(0, -((half - c) // up) if c - half < 0 else -(-(c - half) // up))
        i1 = min(n_in - 1, (c + half) // up)
        if i1 >= i0:
            i = np.arange(i0, i1 + 1)
            out[j] = np.dot(mono[i], h[c - i * up + half])
    return out
