"""Result rows as CSV text without a DataFrame in the way.

The reference formats every chunk with pandas (src/write/formatting.py:31-50 + ``DataFrame.to_csv``,
src/write/worker.py:67-87): values rounded to two decimals, written with the shortest float repr
("-1.28", "0.5", "3.0", "-0.0"), ``start`` likewise, ``\\n`` line ends.  That costs ~20 us per row; at the
engine's rate a GPU produces a row every microsecond.  Here the same bytes are assembled with NumPy:
every cell is a fixed-size record filled from lookup tables, unused positions are NUL and squeezed out at the
end.  Activations below 100 in magnitude (the usual case: logits are O(10)) are ONE 8-byte word each,
",-dd.ff", looked up by their value in hundredths; larger ones take a 10-byte record
[sign][5 integer digits][.][2 fraction digits][separator].  Anything neither form can express (non-finite
values, |x| >= 100 000) goes through pandas, so the output is always what the reference would have written
(``tests/test_fastcsv.py`` compares the bytes).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

_LIMIT = 100_000                      # integer parts have at most five digits in the fast form


def _tables():
    """ints[i] = the decimal digits of i, left-aligned in five bytes (NUL behind them); frac[f] = the two digits of f / 100
    without a trailing zero.  Built with array arithmetic (a Python loop over the 100 000 entries took 70 ms of every first
    call); tests/test_fastcsv.py compares every entry with str()."""
    i = np.arange(_LIMIT)
    digits = np.ones(_LIMIT, np.int64)
    for p in (10, 100, 1000, 10000):
        digits += i >= p
    ints = np.zeros((_LIMIT, 5), np.uint8)
    for k in range(5):
        p = digits - 1 - k
        ints[:, k] = np.where(p >= 0, (i // 10 ** np.maximum(p, 0)) % 10 + 48, 0)
    f = np.arange(100)
    frac = np.zeros((100, 2), np.uint8)
    frac[:, 0] = f // 10 + 48
    frac[:, 1] = np.where(f % 10 == 0, 0, f % 10 + 48)
    return ints, frac


_INT_TAB, _FRAC_TAB = None, None


def _cells(hundredths: np.ndarray, negative: np.ndarray, out: np.ndarray) -> None:
    """out[..., 0:9] <- text of hundredths / 100 (shortest repr), sign from ``negative``."""
    global _INT_TAB, _FRAC_TAB
    if _INT_TAB is None:
        _INT_TAB, _FRAC_TAB = _tables()
    h = np.abs(hundredths)
    out[..., 0] = np.where(negative, ord("-"), 0)
    out[..., 1:6] = _INT_TAB[h // 100]
    out[..., 6] = ord(".")
    out[..., 7:9] = _FRAC_TAB[h % 100]


_SMALL = 10_000                       # |value| < 100 <=> |hundredths| < 10 000: one 8-byte word per cell
_WORD_TAB = None


def _word_table() -> np.ndarray:
    """uint64 per value: the bytes of ',' + shortest repr of k / 100, for k = -9999..9999 at index k + 10 000;
    index 0 is ',-0.0' (a negative value that rounded to zero)."""
    k = np.arange(-_SMALL + 1, _SMALL)
    a = np.abs(k)
    whole, fr = a // 100, a % 100
    n = k.size
    pieces = np.stack([np.full(n, ord(","), np.uint8),
                       np.where(k < 0, ord("-"), 0).astype(np.uint8),
                       np.where(whole >= 10, whole // 10 + 48, 0).astype(np.uint8),
                       (whole % 10 + 48).astype(np.uint8),
                       np.full(n, ord("."), np.uint8),
                       (fr // 10 + 48).astype(np.uint8),
                       np.where(fr % 10 == 0, 0, fr % 10 + 48).astype(np.uint8)], 1)
    packed = np.take_along_axis(pieces, np.argsort(pieces == 0, axis=1, kind="stable"), 1)     # NULs to the end
    tab = np.zeros((2 * _SMALL, 8), np.uint8)
    tab[k + _SMALL, :7] = packed
    tab[0, :5] = np.frombuffer(b",-0.0", np.uint8)
    return tab.view(np.uint64).reshape(-1)


def _rows_small(start: np.ndarray, values: np.ndarray) -> Optional[bytes]:
    """Fast form for float activations with |x| < 100: [start: 2 words][one word per value][newline word]."""
    global _WORD_TAB
    if _WORD_TAB is None:
        _WORD_TAB = _word_table()
    n, cols = values.shape
    v32 = values.astype(np.float32, copy=False)
    if not (np.abs(v32) < np.float32(99.99)).all():              # false for NaN / inf as well
        return None
    k = np.rint(v32 * np.float32(100.0)).astype(np.int32)        # exact: x is the float32 next to k / 100
    idx = k + _SMALL
    idx[(k == 0) & np.signbit(v32)] = 0
    words = np.zeros((n, cols + 3), np.uint64)
    hs = np.rint(start * 100.0).astype(np.int64)
    if not np.isfinite(start).all() or (np.abs(hs) >= _LIMIT * 100).any():
        return None
    cell = words[:, :2].view(np.uint8).reshape(n, 16)
    _cells(hs, np.signbit(start), cell[:, :10])                  # bytes 0..8 text, 9 stays NUL
    words[:, 2:cols + 2] = _WORD_TAB[idx]
    words[:, cols + 2] = ord("\n")
    flat = words.view(np.uint8).reshape(-1)
    return flat[flat != 0].tobytes()


def rows(start: np.ndarray, values: np.ndarray) -> Optional[bytes]:
    """CSV text of ``start`` (float64, already rounded to 2 decimals) followed by the columns of ``values`` (float32
    or float64, already rounded to 2 decimals; or an integer array, written as integers).  ``None`` if a value does not
    fit the fast form (the caller then formats with pandas)."""
    start = np.asarray(start, dtype=np.float64)
    values = np.asarray(values)
    n = start.shape[0]
    if values.ndim != 2 or values.shape[0] != n:
        raise ValueError("values must be [rows, columns]")
    if n == 0:
        return b""
    cols = values.shape[1]
    if np.issubdtype(values.dtype, np.floating):
        out = _rows_small(start, values)
        if out is not None:
            return out
    buf = np.zeros((n, cols + 1, 10), np.uint8)
    if not np.isfinite(start).all():
        return None
    hs = np.rint(start * 100.0).astype(np.int64)
    if (np.abs(hs) >= _LIMIT * 100).any():
        return None
    _cells(hs, np.signbit(start), buf[:, 0, :])
    if np.issubdtype(values.dtype, np.integer):
        if (values < 0).any() or (values >= _LIMIT).any():
            return None
        global _INT_TAB, _FRAC_TAB
        if _INT_TAB is None:
            _INT_TAB, _FRAC_TAB = _tables()
        buf[:, 1:, 1:6] = _INT_TAB[values]
    else:
        if not np.isfinite(values).all():
            return None
        hv = np.rint(values.astype(np.float64) * 100.0).astype(np.int64)
        if (np.abs(hv) >= _LIMIT * 100).any():
            return None
        _cells(hv, np.signbit(values), buf[:, 1:, :])
    buf[:, :, 9] = ord(",")
    buf[:, cols, 9] = ord("\n")
    flat = buf.reshape(-1)
    return flat[flat != 0].tobytes()


_NATIVE = None          # the C-ABI library's bd_format_rows, False once it turned out to be unavailable
BD_ERANGE = -5              # include/buzzdetect_hip.h


def rows_native(start: np.ndarray, raw: np.ndarray, keep: Optional[Sequence[int]] = None) -> Optional[bytes]:
    """The bytes of ``rows(start, raw.round(2)[:, keep])`` from the library's host-side formatter (``bd_format_rows``,
    csrc/rowfmt.hip: a plain C loop, ~30 x the NumPy assembly above, GIL released).  ``raw`` are the UNROUNDED float32
    logits ``[rows, classes]`` (any row stride); ``None`` when the library is not there or a value does not fit the fixed form
    - the caller then goes through ``rows`` / pandas, which is what decides the bytes in that case."""
    global _NATIVE
    if _NATIVE is None:
        try:
            from . import _lib
            _NATIVE = _lib.load(build_if_missing=False).bd_format_rows
        except Exception:                                   # noqa: BLE001 - formatting must never depend on the library
            _NATIVE = False
    if _NATIVE is False:
        return None
    raw = np.asarray(raw)
    start = np.ascontiguousarray(start, dtype=np.float64)
    if raw.ndim != 2 or raw.dtype != np.float32 or raw.shape[0] != start.shape[0]:
        return None
    n, c = raw.shape
    if n == 0:
        return b""
    if raw.strides[1] != 4 or raw.strides[0] % 4:
        return None
    idx = None if keep is None else np.ascontiguousarray(keep, dtype=np.int32)
    cols = c if idx is None else int(idx.size)
    out = np.empty(n * (10 * cols + 12) + 8, np.uint8)
    got = _NATIVE(raw.ctypes.data, n, c, raw.strides[0] // 4, None if idx is None else idx.ctypes.data, 0 if idx is None else cols,
                  start.ctypes.data, out.ctypes.data, out.size)
    if got < 0:
        if got == BD_ERANGE:
            return None
        raise ValueError(f"bd_format_rows failed ({got})")
    return out[:got].tobytes()


def header(columns: Sequence[str]) -> bytes:
    return (",".join(columns) + "\n").encode()
