"""Chunk and window index arithmetic either side of the hot path, in float64 exactly as the
reference writes it (hazards H1-H4 of SURVEY §8a are properties of these formulas).

    round_chunklength   Analyzer._setup_chunklength      src/analyze.py:102-111
    gaps_to_chunklist   gaps_to_chunklist                src/stream/results_coverage.py:59-70
    chunk_sample_range  WorkerStreamer.queue_chunk       src/stream/worker.py:110-112
    window_starts       add_time                         src/write/formatting.py:5-17
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import numpy as np


def round_chunklength(chunklength: float, framelength_s: float = 0.96, digits_time: int = 2) -> float:
    """Nearest whole number of frames, rounded to ``digits_time``, at least one frame
    (200 -> 199.68, 600 -> 600.0, 1000 -> 1000.32)."""
    out = round(chunklength / framelength_s) * framelength_s
    out = round(out, digits_time)
    return framelength_s if out < framelength_s else out


def gaps_to_chunklist(gaps_in: Iterable[Sequence[float]], chunklength: float, decimals: int = 2
                      ) -> List[Tuple[float, float]]:
    """Cut every (start, end) gap into chunks of ``chunklength`` seconds; the last chunk of a gap is
    whatever remains.  Edges come from ``np.arange`` and are rounded to ``decimals``."""
    chunks: List[Tuple[float, float]] = []
    for start, end in gaps_in:
        edges = np.arange(start, end, chunklength).tolist()
        edges.append(end)                      # arange excludes the right edge even when it aligns
        edges = np.round(edges, decimals)
        chunks.extend(zip(edges[:-1], edges[1:]))
    return chunks


def chunk_sample_range(chunk: Sequence[float], samplerate: int) -> Tuple[int, int]:
    """``int(edge * samplerate)`` truncation for both chunk edges (hazard H3 lives here)."""
    return int(chunk[0] * samplerate), int(chunk[1] * samplerate)


def window_starts(n_windows: int, time_start: float, framehop_s: float, digits_time: int = 2) -> np.ndarray:
    """``start`` column of a chunk's result rows: round(i * framehop_s + time_start, digits)."""
    start = np.arange(n_windows, dtype=np.int64) * framehop_s
    if time_start != 0:
        start = start + time_start
    return np.round(start, digits_time)
