"""Chunk / window index arithmetic against vectors produced by
importing the reference's own helpers (tests/golden/reference_helpers.json, tools/make_fixtures.py)."""
import numpy as np
import pytest

from buzzdetect_amd import framing


def test_gaps_to_chunklist_matches_reference(golden_helpers):
    for case in golden_helpers["gaps_to_chunklist"]:
        got = framing.gaps_to_chunklist([tuple(g) for g in case["gaps"]], case["chunklength"])
        assert [[float(a), float(b)] for a, b in got] == case["chunks"]


def test_known_answer_600s_file():
    got = framing.gaps_to_chunklist([(0, 600.5)], 199.68)
    assert [tuple(map(float, c)) for c in got] == [(0.0, 199.68), (199.68, 399.36), (399.36, 599.04), (599.04, 600.5)]


def test_window_starts_match_reference_add_time(golden_helpers):
    for case in golden_helpers["add_time"]:
        got = framing.window_starts(case["n"], case["time_start"], case["framehop_s"], case["digits"])
        assert got.tolist() == case["start"]


@pytest.mark.parametrize("asked,expect", [(200, 199.68), (600, 600.0), (1000, 1000.32), (0.2, 0.96), (0.96, 0.96)])
def test_round_chunklength_known_answers(asked, expect):
    # src/analyze.py:102-111 cannot be imported (TensorFlow); known answers from SURVEY §8a row a12
    assert framing.round_chunklength(asked) == expect


def test_int_truncation_hazard_h3():
    # 4193.28 * 16000 = 67092479.99999999 -> one sample short
    assert framing.chunk_sample_range((4193.28, 4392.96), 16000)[0] == 67092479
    assert framing.chunk_sample_range((0.0, 199.68), 16000) == (0, 3194880)


def test_chunk_sample_counts_give_expected_windows():
    from oracle import yamnet_oracle as O
    chunks = framing.gaps_to_chunklist([(0, 3600.0)], framing.round_chunklength(200))
    counts = [framing.chunk_sample_range(c, 16000) for c in chunks]
    sizes = [b - a for a, b in counts]
    assert len(chunks) == 19 and sum(sizes) <= 57_600_000
    full = [s for s in sizes[:-1]]
    # full chunks are 3 194 880 samples +-1 (H3) -> 208 or 209 windows, never fewer
    assert all(abs(s - 3_194_880) <= 1 for s in full)
    assert all(O.num_windows(s, 15360) in (208, 209) for s in full)
