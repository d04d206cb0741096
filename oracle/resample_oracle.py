"""CPU restatement of the streamer's downmix + resample stage — TEST INFRASTRUCTURE ONLY.

Reference: ``np.mean(samples, axis=1)`` then ``librosa.resample(y, orig_sr, target_sr)``
(src/stream/worker.py:116-128).  librosa's default ``res_type`` is ``soxr_hq`` (environment.yml:9 pulls librosa, which
pulls soxr); neither librosa nor libsoxr is vendored or installed here and the reference holds no vector for the stage, so
bit-parity with soxr stays **unpinned**.  What CAN be restated is the filter CLASS libsoxr publishes for its HQ recipe
(soxr.h / soxr.c ``soxr_quality_spec``; restated from the published parameters, the source is absent):

    precision      20 bits            -> rejection  rej = 20 * 20 log10(2) = 120.41 dB
    passband_end   1 - 0.05 / TO_3dB(rej),  TO_3dB(a) = (1.6e-6 a - 7.5e-4) a + 0.646       = 0.91363 of the LOWER Nyquist
    stopband_begin 1.0 of the lower Nyquist (no aliasing into the band), linear phase, unity gain, zero net delay,
    output length  ceil(n * rate_out / rate_in)    (librosa fixes the length to this, librosa.resample ``fix=True``)

``quality="hq"`` (the default) is a single linear-phase low-pass of that class at the up-sampled rate rate_in * up: a
Kaiser-windowed sinc whose -6 dB point sits midway between the two band edges (as libsoxr's ``lsx_design_lpf`` places
it), beta = 0.1102 (A - 8.7) and length from Kaiser's estimate N = (A - 7.95) / (2.285 dw) with the design attenuation
A = 125 dB (libsoxr's 120.41 dB plus a margin that keeps the realised stop band under -120 dB once the taps are rounded
to float32).  libsoxr itself realises the response as a cascade (half-band stages + a polyphase / DFT stage); a cascade
and this single stage agree to within the stop-band leakage and pass-band ripple of either (~1e-6), not bit for bit.
tests/test_resample.py asserts the response: ripple <= 0.01 dB up to 0.9125 of the lower Nyquist, <= -120 dB from it on.

``quality="scipy"`` is round 1-3's filter, kept selectable (``bd_set_resample_quality(h, 0)``): the
scipy.signal.resample_poly default (Kaiser 5.0 windowed sinc, 20*max(up,down)+1 taps) - about -30 dB one kHz into the
stop band and -1.8 dB at 7.5 kHz for 48 -> 16 kHz, i.e. NOT the reference's filter class.

Both are applied the same way:

    y[j] = sum_i mono[i] * h[j*down - i*up + half]
"""
from math import ceil, gcd, pi

import numpy as np
import scipy.signal

HQ_PRECISION_BITS = 20
HQ_DESIGN_ATTENUATION_DB = 125.0


def ratio(rate_in: int, rate_out: int):
    g = gcd(int(rate_in), int(rate_out))
    return rate_out // g, rate_in // g


def hq_band_edges():
    """(passband_end, stopband_begin) as fractions of the lower of the two Nyquist frequencies (libsoxr's HQ recipe)."""
    rej = HQ_PRECISION_BITS * 20.0 * np.log10(2.0)
    to_3db = (1.6e-6 * rej - 7.5e-4) * rej + 0.646
    return 1.0 - 0.05 / to_3db, 1.0


def kaiser_lowpass(half: int, cutoff: float, beta: float):
    """Unity-DC-gain windowed sinc with 2*half+1 taps; cutoff (the -6 dB point) as a fraction of Nyquist."""
    m = np.arange(-half, half + 1, dtype=np.float64)
    r = m / half
    w = np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / np.i0(beta)
    h = cutoff * np.sinc(cutoff * m) * w
    return h / h.sum()


def taps(up: int, down: int, dtype=np.float64, quality: str = "hq"):
    max_rate = max(up, down)
    if quality == "scipy":
        half = 10 * max_rate
        if max_rate == 1:                  # equal rates: resample_poly returns the input; the filter is a delta
            h = np.zeros(2 * half + 1)
            h[half] = 1.0
            return h.astype(dtype), half
        h = scipy.signal.firwin(2 * half + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
        return h.astype(dtype), half
    if quality != "hq":
        raise ValueError(f"quality {quality!r}")
    if max_rate == 1:                      # equal rates: soxr copies the input
        return np.ones(1, dtype), 0
    fp, fs = (e / max_rate for e in hq_band_edges())      # fractions of the Nyquist of the up-sampled rate
    a = HQ_DESIGN_ATTENUATION_DB
    beta = 0.1102 * (a - 8.7)
    n = int(ceil((a - 7.95) / (2.285 * pi * (fs - fp)))) + 1
    half = n // 2
    h = kaiser_lowpass(half, 0.5 * (fp + fs), beta) * up
    return h.astype(dtype), half


def downmix(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    return x if x.ndim == 1 else np.mean(x, axis=1)          # float32 mean, as the reference computes it


def resample_direct(x: np.ndarray, rate_in: int, rate_out: int = 16000, dtype=np.float64, quality: str = "hq") -> np.ndarray:
    """The definition, one output at a time (small inputs)."""
    mono = downmix(x).astype(dtype)
    up, down = ratio(rate_in, rate_out)
    h, half = taps(up, down, dtype, quality)
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


def resample(x: np.ndarray, rate_in: int, rate_out: int = 16000, dtype=np.float64, quality: str = "hq") -> np.ndarray:
    """The same sums through scipy.signal.upfirdn (polyphase; tests/test_resample.py holds it to resample_direct)."""
    mono = downmix(x).astype(dtype)
    up, down = ratio(rate_in, rate_out)
    h, half = taps(up, down, dtype, quality)
    n_in = mono.shape[0]
    n_out = -(-n_in * up // down)
    if n_in == 0:
        return np.zeros(0, dtype)
    # upfirdn output k = sum_i mono[i] h[k*down - i*up]; y[j] wants tap index j*down - i*up + half: prepend zeros so that
    # the filter's centre lands on a multiple of `down`
    pre = (-half) % down
    full = scipy.signal.upfirdn(np.concatenate([np.zeros(pre, dtype), h]), mono, up, down)
    first = (half + pre) // down
    out = np.zeros(n_out, dtype)
    got = full[first:first + n_out]
    out[:got.shape[0]] = got
    return out
