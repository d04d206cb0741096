"""fastcsv / activation_csv / detection_csv must produce the bytes of the reference's pandas formatting."""
import numpy as np
import pandas as pd
import pytest

from buzzdetect_amd import fastcsv, results as R

CLASSES = ["ambient_background", "ambient_rain", "ins_buzz", "mech_plane"]


def pandas_bytes(table: pd.DataFrame) -> bytes:
    return table.to_csv(index=False).encode()


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (1, 8.0), (2, 300.0), (3, 0.004), (4, 30000.0)])
def test_activation_rows_equal_pandas(seed, scale):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((977, len(CLASSES))) * scale).astype(np.float32)
    x[3, 1] = -0.001            # rounds to -0.0: pandas writes "-0.0"
    x[4, 2] = 0.004999
    x[5, 0] = 12.5
    x[6, 3] = -7.0
    for start, hop in ((0.0, 0.96), (199.68, 0.96), (86000.0, 0.48), (0.48, 0.48)):
        head, body = R.activation_csv(x, CLASSES, hop, 2, start, "all", 2)
        ref = pandas_bytes(R.activation_table(x, CLASSES, hop, 2, start, "all", 2))
        assert head + body == ref
        head, body = R.activation_csv(x, CLASSES, hop, 2, start, ["ins_buzz", "ambient_rain"], 2)
        assert head + body == pandas_bytes(R.activation_table(x, CLASSES, hop, 2, start, ["ins_buzz", "ambient_rain"], 2))


def test_values_outside_the_fast_form_fall_back_to_pandas():
    x = np.zeros((5, len(CLASSES)), np.float32)
    for bad in (np.nan, np.inf, -np.inf, 2.5e6, -1e9):
        y = x.copy()
        y[2, 1] = bad
        assert fastcsv.rows(np.arange(5) * 0.96, y.round(2)) is None
        head, body = R.activation_csv(y, CLASSES, 0.96, 2, 0.0)
        assert head + body == pandas_bytes(R.activation_table(y, CLASSES, 0.96, 2, 0.0))
    # a start beyond 100 000 s (27.8 h into a recording) also goes through pandas
    head, body = R.activation_csv(x, CLASSES, 0.96, 2, 123456.0)
    assert head + body == pandas_bytes(R.activation_table(x, CLASSES, 0.96, 2, 123456.0))
    assert fastcsv.rows(np.zeros(0), np.zeros((0, 3), np.float32)) == b""


def test_native_formatter_writes_fastcsv_bytes():
    """bd_format_rows (csrc/rowfmt.hip, host code of the C-ABI library) against fastcsv.rows on the ROUNDED values, which the
    tests above hold to pandas: random logits at five scales, every two-decimal tie and boundary, negative zero, a strided
    view (the writer formats slices of a batch's pinned block), column subsets in model order."""
    assert fastcsv.rows_native(np.zeros(1), np.zeros((1, 2), np.float32)) is not None, "the library must be loadable on a CPU box"
    edge = np.array([0.0, -0.0, 0.004, 0.005, 0.0050001, 0.015, 0.025, 0.035, -0.004, -0.005, -0.0051, 0.994, 0.995, 0.996, 9.995,
                     99.98, 99.985, 99.99, 99.994, 99.995, 99.996, 100.0, 100.004, 100.005, 123.456, 999.995, 1000.0, 12345.678,
                     65535.996, 99999.0, 99999.98, 99999.99, -99999.99, -100.0, -99.995, 1e-30, -1e-30, 0.1, 0.7, 2.675, 1.005],
                    np.float32)
    for seed, scale in [(0, 1.0), (1, 8.0), (2, 300.0), (3, 0.004), (4, 30000.0)]:
        rng = np.random.default_rng(seed)
        x = (rng.standard_normal((977, 13)) * scale).astype(np.float32)
        x[:edge.size, 5] = edge
        x[:edge.size, 0] = -edge
        for start, hop in ((0.0, 0.96), (199.68, 0.96), (86000.0, 0.48), (99990.0, 0.96)):
            starts = np.round(np.arange(x.shape[0]) * hop + start, 2)
            if (starts >= 100000).any():
                assert fastcsv.rows_native(starts, x) is None and fastcsv.rows(starts, x.round(2)) is None
                continue
            want = fastcsv.rows(starts, x.round(2))
            if scale > 1000:                                                # some |x| >= 1e5: both refuse, pandas decides
                assert want is None and fastcsv.rows_native(starts, x) is None
                small = np.clip(x, -99999.0, 99999.0)
                assert fastcsv.rows_native(starts, small) == fastcsv.rows(starts, small.round(2)) != None      # noqa: E711
                continue
            assert want is not None
            assert fastcsv.rows_native(starts, x) == want, (seed, start)
            keep = [2, 5, 11]
            assert fastcsv.rows_native(starts, x, keep) == fastcsv.rows(starts, x.round(2)[:, keep])
            big = np.zeros((x.shape[0] + 7, 16), np.float32)               # a strided view: rows 3.. of a wider block
            big[3:3 + x.shape[0], :13] = x
            assert fastcsv.rows_native(starts, big[3:3 + x.shape[0], :13]) == want
    for bad in (np.nan, np.inf, -np.inf, 2.5e6, -1e9, 1e5, -100000.0):
        y = np.zeros((5, 4), np.float32)
        y[2, 1] = bad
        assert fastcsv.rows_native(np.arange(5) * 0.96, y) is None, bad
        assert fastcsv.rows(np.arange(5) * 0.96, y.round(2)) is None, bad
    assert fastcsv.rows_native(np.zeros(0), np.zeros((0, 3), np.float32)) == b""
    assert fastcsv.rows_native(np.zeros(3), np.zeros((3, 3), np.float64)) is None       # float32 only: others take the NumPy path


def test_detection_rows_equal_pandas():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((300, len(CLASSES))).astype(np.float32) * 2
    for thr in (-1.205, 0.0, 5.0):
        head, body = R.detection_csv(x, thr, CLASSES, 0.96, 2, 600.0)
        assert head + body == pandas_bytes(R.detection_table(x, thr, CLASSES, 0.96, 2, 600.0))


def test_sorted_partial_file_survives_the_finalize_round_trip(tmp_path):
    """finalize() = read partial, sort by start, write complete.  For rows appended in start order that is the
    identity on the bytes, which is what lets the writer rename instead (ResultFile.finalize_sorted)."""
    rng = np.random.default_rng(11)
    a, b = R.ResultFile(str(tmp_path / "a")), R.ResultFile(str(tmp_path / "b"))
    for chunk in range(4):
        x = (rng.standard_normal((208, len(CLASSES))) * 6).astype(np.float32)
        x[0, 0] = -0.001
        head, body = R.activation_csv(x, CLASSES, 0.96, 2, round(chunk * 199.68, 2))
        a.append_text(head, body)
        b.append_text(head, body)
    a.finalize()
    b.finalize_sorted()
    assert open(a.path_complete, "rb").read() == open(b.path_complete, "rb").read()
    assert not (tmp_path / "a_buzzpart.csv").exists() and not (tmp_path / "b_buzzpart.csv").exists()


def test_lookup_tables_entry_by_entry():
    """The tables are built with array arithmetic; every entry against Python's own formatting."""
    ints, frac = fastcsv._tables()
    for i in list(range(0, 1200)) + list(range(9990, 10011)) + list(range(99900, 100000)) + [12345, 54321, 70007]:
        assert bytes(ints[i]).rstrip(b"\0") == str(i).encode(), i
    for f in range(100):
        assert bytes(frac[f]).rstrip(b"\0") == (f"{f:02d}".rstrip("0") or "0").encode(), f
    words = fastcsv._word_table().view(np.uint8).reshape(-1, 8)
    for k in range(-9999, 10000):
        text = f",{'-' if k < 0 else ''}{abs(k) // 100}.{(f'{abs(k) % 100:02d}'.rstrip('0') or '0')}".encode()
        assert bytes(words[k + 10000]).rstrip(b"\0") == text, k
    assert bytes(words[0]).rstrip(b"\0") == b",-0.0"
