"""analyze() end to end on a GPU box: WAV in (16 kHz mono and 48 kHz stereo), reference-format CSV out,
resume after an interrupted run, manifest lock."""
import os
import wave

import numpy as np
import pytest

from oracle import yamnet_oracle as O
from oracle import resample_oracle as RO


def write_wav(path, x, rate):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[1])
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.clip(x, -1, 1 - 2 ** -15) * 32768.0).round().astype("<i2").tobytes())


def test_wav_reader_roundtrip(tmp_path):
    from buzzdetect_amd.analyze import WavTrack, build_ident
    x = (np.arange(2000).reshape(1000, 2) % 97 - 48) / 64.0
    write_wav(tmp_path / "a.wav", x, 32000)
    t = WavTrack(str(tmp_path / "a.wav"))
    assert (t.samplerate, t.channels, t.frames) == (32000, 2, 1000) and t.duration == 1000 / 32000
    t.seek(10)
    got = t.read(5)
    assert got.shape == (5, 2) and np.allclose(got, x[10:15], atol=1 / 32768)
    t.seek(998)
    assert t.read(10).shape == (2, 2)              # short read at the end of the file
    t.close()
    assert build_ident("/data/audio/site1/rec.wav", "/data/audio") == "site1/rec"


@pytest.mark.gpu
def test_analyze_writes_reference_format_and_resumes(engine, weights_bundle, tmp_path):
    import pandas as pd
    from buzzdetect_amd.analyze import analyze
    from buzzdetect_amd import results as R
    audio, out = tmp_path / "audio", tmp_path / "out"
    (audio / "site").mkdir(parents=True)
    x16 = O.synthetic_audio(16000 * 9 + 123, seed=8)
    write_wav(audio / "site" / "mono16.wav", x16, 16000)
    t = np.arange(48000 * 5) / 48000.0
    st = np.stack([0.3 * np.sin(2 * np.pi * 300 * t), 0.2 * np.sin(2 * np.pi * 1500 * t)], 1)
    write_wav(audio / "stereo48.wav", st, 48000)

    rep = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], framehop_prop=1, chunklength=3,
                  dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep.files_done == 2 and rep.files_total == 2
    a = pd.read_csv(out / "site" / "mono16_buzzdetect.csv")
    assert list(a.columns) == ["start", "activation_ambient_rain", "activation_ins_buzz"]   # model order (H5)
    assert not (out / "site" / "mono16_buzzpart.csv").exists()
    # chunklength 3 -> 2.88 s = 3 frames; 9.0077 s -> chunks [0,2.88) [2.88,5.76) [5.76,8.64) [8.64,9.01)
    assert a["start"].tolist() == [0.0, 0.96, 1.92, 2.88, 3.84, 4.8, 5.76, 6.72, 7.68, 8.64]
    # values: the 16-bit samples the reader produced, chunked the same way, through the f64 oracle
    b = weights_bundle
    x_q = (np.clip(x16, -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16).astype(np.float32) / 32768.0
    ref = O.predict(x_q[: int(2.88 * 16000)], b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert np.abs(a["activation_ins_buzz"].to_numpy()[:3] - ref[:, 8].round(2)).max() <= 0.011
    s = pd.read_csv(out / "stereo48_buzzdetect.csv")
    assert s["start"].tolist() == [0.0, 0.96, 1.92, 2.88, 3.84, 4.8]        # chunks [0,2.88) + [2.88,5.0): 3 + 3 rows
    q = (np.clip(st, -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16).astype(np.float32) / 32768.0
    mono = RO.resample(q[: int(2.88 * 48000)], 48000).astype(np.float32)
    ref48 = O.predict(mono, b["blob"], b["mel"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert np.abs(s["activation_ins_buzz"].to_numpy()[:3] - ref48[:, 8].round(2)).max() <= 0.011

    # second run: nothing to do; a different setting in the same folder is refused by the manifest
    rep2 = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], chunklength=3,
                   dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep2.files_done == 0 and rep2.files_skipped == 2
    with pytest.raises(RuntimeError, match="different settings"):
        analyze("model_general_v3", precision=0.95, chunklength=3, dir_audio=str(audio), dir_out=str(out), engine=engine)

    # interrupted run: keep only the rows of chunks 0 and 2 as a partial file -> only the gaps are redone
    os.remove(out / "site" / "mono16_buzzdetect.csv")
    a.iloc[[0, 1, 2, 6, 7, 8]].to_csv(out / "site" / "mono16_buzzpart.csv", index=False)
    rep3 = analyze("model_general_v3", classes_out=["ins_buzz", "ambient_rain"], chunklength=3,
                   dir_audio=str(audio), dir_out=str(out), engine=engine)
    # gaps are (2.88, 5.76) and (8.64, 9.0077); like the reference (results_coverage.py:45-46) a gap that
    # starts within one frame of the end is ignored, so one chunk is redone and the 8.64 row stays missing
    assert rep3.files_done == 1 and rep3.chunks == 1
    a3 = pd.read_csv(out / "site" / "mono16_buzzdetect.csv")
    assert a3.equals(a.iloc[:9].reset_index(drop=True))

    # detections mode in a fresh folder
    out2 = tmp_path / "out_det"
    analyze("model_general_v3", precision=0.95, chunklength=200, dir_audio=str(audio), dir_out=str(out2), engine=engine)
    d = pd.read_csv(out2 / "site" / "mono16_buzzdetect.csv")
    assert list(d.columns) == ["start", "detections_ins_buzz"] and set(d["detections_ins_buzz"]) <= {0, 1}
    assert R.threshold_for_precision("model_general_v3", 0.95) == pytest.approx(-1.205)


# --------------------------------------------------------------------------------------------------- WAV parsing (CPU)
def _riff(fmt_chunk: bytes, data: bytes, extra: bytes = b"") -> bytes:
    import struct
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + extra + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_parser_formats_the_stdlib_reader_rejects(tmp_path):
    """WAVE_FORMAT_EXTENSIBLE, IEEE float and 24-bit files (python's wave module refuses the first two), an odd-sized
    LIST chunk before the data, and a data chunk whose count runs past the end of the file (recorder died)."""
    import struct
    from buzzdetect_amd.wavio import WavFormatError, WavTrack
    x = (np.arange(64, dtype=np.float32).reshape(32, 2) - 30) / 40
    # IEEE float 32, stereo, 48 kHz
    fmt = struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32)
    (tmp_path / "f32.wav").write_bytes(_riff(fmt, x.astype("<f4").tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\0"))
    t = WavTrack(str(tmp_path / "f32.wav"))
    assert (t.samplerate, t.channels, t.frames, t.is_s16) == (48000, 2, 32, False)
    assert np.array_equal(t.read(32), x)
    # extensible PCM 16-bit: sub-format tag 1 at offset 24 of the fmt chunk
    s = (x * 20000).astype("<i2")
    fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 16000, 16000 * 4, 4, 16) + struct.pack("<HHI", 22, 16, 3) + struct.pack("<H", 1) + b"\0" * 14
    (tmp_path / "ext.wav").write_bytes(_riff(fmt, s.tobytes()))
    t = WavTrack(str(tmp_path / "ext.wav"))
    assert t.is_s16 and t.frames == 32
    t.seek(3)
    assert np.array_equal(t.read_s16(4), s[3:7])
    t.seek(30)
    assert t.read(10).shape == (2, 2)                       # short read at the end
    assert np.allclose(t.read(0), np.zeros((0, 2)))
    # 24-bit mono
    v = np.array([0, 1, -1, 8388607, -8388608, 4194304], dtype=np.int64)
    raw = b"".join(int(i & 0xFFFFFF).to_bytes(3, "little") for i in v)
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 48000, 3, 24)
    (tmp_path / "p24.wav").write_bytes(_riff(fmt, raw))
    got = WavTrack(str(tmp_path / "p24.wav")).read(6)[:, 0]
    assert np.allclose(got, v / 8388608.0)
    # data count beyond the file end: frames follow the bytes that exist
    blob = bytearray(_riff(struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16), s.tobytes()))
    blob[blob.index(b"data") + 4: blob.index(b"data") + 8] = struct.pack("<I", 10_000_000)
    (tmp_path / "cut.wav").write_bytes(bytes(blob))
    cut = WavTrack(str(tmp_path / "cut.wav"))
    assert cut.frames == 64 and cut.frames_declared == 5_000_000           # what can be read / what the header promises
    assert cut.duration == 5_000_000 / 16000 and cut.duration_readable == 64 / 16000
    # 0xFFFFFFFF: "length unknown" (streaming writers): the file's own length is all there is
    blob[blob.index(b"data") + 4: blob.index(b"data") + 8] = struct.pack("<I", 0xFFFFFFFF)
    (tmp_path / "stream.wav").write_bytes(bytes(blob))
    t = WavTrack(str(tmp_path / "stream.wav"))
    assert t.frames == t.frames_declared == 64
    for bad in (b"RIFF\0\0\0\0WAVX", _riff(struct.pack("<HHIIHH", 85, 1, 16000, 2000, 1, 0), b"\0" * 64)):   # not WAVE; MP3-in-WAV
        (tmp_path / "bad.wav").write_bytes(bad)
        with pytest.raises(WavFormatError):
            WavTrack(str(tmp_path / "bad.wav"))


# --------------------------------------------------------------------------------------------------- pipeline (GPU)
def _three_recordings(audio):
    (audio / "a").mkdir(parents=True)
    write_wav(audio / "a" / "one.wav", O.synthetic_audio(16000 * 31 + 77, seed=1), 16000)
    write_wav(audio / "two.wav", O.synthetic_audio(16000 * 12, seed=2), 16000)
    t = np.arange(32000 * 9) / 32000.0
    write_wav(audio / "three32k.wav", 0.4 * np.sin(2 * np.pi * 440 * t), 32000)


def _cut_short(path, keep_seconds: float, rate: int = 16000):
    """Truncate a WAV written by write_wav behind `keep_seconds` of audio: the header keeps the full count (a recorder
    whose battery died, src/stream/worker.py:41-44)."""
    with open(path, "r+b") as f:
        f.truncate(44 + int(keep_seconds * rate) * 2)


def _reader_stage(tmp_path, chunklength):
    """A Pipeline whose planner and reader stages run without a device (pageable staging buffers)."""
    from buzzdetect_amd import pipeline as P, results as R
    pipe = P.Pipeline(make_engine=None, classes=["a"], framehop_s=0.96, hop=15360, step=96, chunklength=chunklength,
                      framelength_s=0.96, digits_time=2, digits_results=2, classes_out="all", threshold=None, readers=1,
                      analyzers=1, pin_memory=False, stream_buffer_depth=64)
    def job(name):
        return P.FileJob(str(tmp_path / name), name[:-4], name, R.ResultFile(str(tmp_path / "out" / name[:-4])))
    def drain(q):
        got = []
        while not q.empty():
            got.append(q.get())
        return got
    return pipe, job, drain


def test_a_file_cut_short_is_said_once_truncated_and_stopped(tmp_path, caplog):
    """The reference's only failure handling on this path (src/stream/worker.py:41-59, 119-127; src/config.py:18): a read
    that returns fewer frames than the header promised -> "Unreadable audio at <x>s out of <y>s for <file>." - WARNING (+
    "Aborting early ...") when more than 1 % of the file is missing, DEBUG (+ "Bad audio is near file end ...") otherwise -
    the chunk ends at round(start + got / rate, 1) and the file ends there.  Reader stage only: no device needed."""
    import logging
    x = O.synthetic_audio(16000 * 100, seed=3)
    write_wav(tmp_path / "dead.wav", x, 16000)
    _cut_short(tmp_path / "dead.wav", 60.3)
    write_wav(tmp_path / "tail.wav", x, 16000)
    _cut_short(tmp_path / "tail.wav", 99.5)
    write_wav(tmp_path / "whole.wav", x, 16000)
    (tmp_path / "out").mkdir()
    pipe, job, drain = _reader_stage(tmp_path, 19.2)
    with caplog.at_level(logging.DEBUG, logger="buzzdetect"):
        for name in ("dead.wav", "tail.wav", "whole.wav"):
            j = job(name)
            pipe._plan_file(j)
            units = drain(pipe.q_units)
            assert [u.chunk for u in units] == [(0.0, 19.2), (19.2, 38.4), (38.4, 57.6), (57.6, 76.8), (76.8, 96.0), (96.0, 100.0)]
            for u in units:
                pipe._read_unit(u)
            tasks = [t for t in drain(pipe.q_analyze)]
            finished = [w for w in drain(pipe.q_write)]
            if name == "dead.wav":
                assert [t.chunk for t in tasks] == [(0.0, 19.2), (19.2, 38.4), (38.4, 57.6), (57.6, 60.3)]
                assert tasks[-1].frames == int(60.3 * 16000) - int(57.6 * 16000)
                assert j.outstanding == 4 and j.bad_read and not finished          # two chunks dropped, four to analyze
            elif name == "tail.wav":
                assert [t.chunk for t in tasks][-1] == (96.0, 99.5) and len(tasks) == 6 and j.bad_read
            else:
                assert [t.chunk for t in tasks][-1] == (96.0, 100.0) and not j.bad_read
    said = [r for r in caplog.records if "Unreadable audio" in r.getMessage()]
    assert len(said) == 2                                                           # once per file, not once per chunk
    dead = next(r for r in said if "dead.wav" in r.getMessage())
    assert dead.levelno == logging.WARNING
    assert "Unreadable audio at 60.3s out of 100.0s for dead.wav." in dead.getMessage()
    assert "Aborting early due to corrupt audio data." in dead.getMessage()
    tail = next(r for r in said if "tail.wav" in r.getMessage())
    assert tail.levelno == logging.DEBUG and "Unreadable audio at 99.5s out of 100.0s for tail.wav." in tail.getMessage()
    assert "Bad audio is near file end, results should be mostly unaffected." in tail.getMessage()
    assert any("dead.wav" in m for m in pipe.report.messages)


@pytest.mark.gpu
def test_analyze_of_a_file_cut_short_writes_what_could_be_read(engine, tmp_path, caplog):
    """End to end on the device: rows up to the truncation point, the complete file written (a re-run skips it), the
    reference's WARNING on the log."""
    import logging
    import pandas as pd
    from buzzdetect_amd.analyze import analyze
    audio, out = tmp_path / "audio", tmp_path / "out"
    audio.mkdir()
    x = O.synthetic_audio(16000 * 100, seed=3)
    write_wav(audio / "dead.wav", x, 16000)
    _cut_short(audio / "dead.wav", 60.3)
    with caplog.at_level(logging.DEBUG, logger="buzzdetect"):
        rep = analyze("model_general_v3", framehop_prop=1, chunklength=19.2, dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep.files_done == 1
    a = pd.read_csv(out / "dead_buzzdetect.csv")
    assert a["start"].iloc[0] == 0.0 and 59.0 < a["start"].iloc[-1] <= 60.3 and a["start"].is_monotonic_increasing
    # chunk (57.6, 60.3): 2.7 s = 43 200 samples -> 1 + ceil((43 200 - 15 600) / 15 360) = 3 windows at 57.6, 58.56, 59.52
    assert a["start"].tolist()[-3:] == [57.6, 58.56, 59.52] and len(a) == 3 * 20 + 3
    assert sum("Unreadable audio at 60.3s out of 100.0s" in r.getMessage() for r in caplog.records) == 1
    rep2 = analyze("model_general_v3", framehop_prop=1, chunklength=19.2, dir_audio=str(audio), dir_out=str(out), engine=engine)
    assert rep2.files_skipped == 1 and rep2.files_done == 0


@pytest.mark.gpu
def test_pipeline_two_analyzers_equals_one_and_logs_the_reference_lines(engine, tmp_path, caplog, monkeypatch):
    """Two analyzer threads, each with its own engine and stream, fed by four readers, write exactly the files a single
    analyzer writes; the log carries the reference's PROGRESS rate line for every chunk (src/inference/worker.py:54-64)."""
    import logging
    import re
    from buzzdetect_amd.analyze import analyze
    from buzzdetect_amd.pipeline import PROGRESS
    audio = tmp_path / "audio"
    _three_recordings(audio)
    ref = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "one"), engine=engine)
    # slow readers (as decoding compressed audio is for the reference): the analyzers must report that they starve
    import time
    from buzzdetect_amd import pipeline as P
    fast_read = P.ReaderStage.read                      # (16-bit PCM goes file -> device through the native stager)
    monkeypatch.setattr(P.ReaderStage, "read", lambda self, fd, off, n, dev: (time.sleep(0.05), fast_read(self, fd, off, n, dev))[1])
    with caplog.at_level(logging.DEBUG, logger="buzzdetect"):
        rep = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "two"),
                      analyzers_gpu=2, n_streamers=4)
    assert (rep.files_done, rep.chunks, rep.windows) == (ref.files_done, ref.chunks, ref.windows) == (3, rep.chunks, rep.windows)
    assert rep.chunks == 7 + 3 + 2 and rep.files_done == 3
    for rel in ("a/one", "two", "three32k"):
        a = (tmp_path / "one" / f"{rel}_buzzdetect.csv").read_bytes()
        b = (tmp_path / "two" / f"{rel}_buzzdetect.csv").read_bytes()
        assert a == b and a.count(b"\n") > 3
        assert not (tmp_path / "two" / f"{rel}_buzzpart.csv").exists()
    progress = [r.getMessage() for r in caplog.records if r.levelno == PROGRESS]
    assert len(progress) == rep.chunks
    pat = re.compile(r"^analyzer [01]: analyzed (a/one|two|three32k)\.wav, chunk \(\d+\.\d\d, \d+\.\d\d\) in \d+\.\d\ds \(rate: \d+\.\d\)$")
    assert all(pat.match(m) for m in progress), progress[:3]
    assert any("analyzed a/one.wav, chunk (4.80, 9.60)" in m for m in progress)
    msgs = [r.getMessage() for r in caplog.records]
    assert sum("launching" in m and m.startswith("analyzer") for m in msgs) == 2
    assert any(re.match(r"^analyzer \d: BUFFER BOTTLENECK: analyzer \d received assignment after \d+\.\ds$", m) for m in msgs)


@pytest.mark.gpu
def test_overflow_repeat_of_unresampled_float_chunks_reads_its_own_audio(tmp_path, monkeypatch):
    """float32 mono 16 kHz needs no resampling: the kernels read such chunks straight out of the analyzer's raw arena.
    When batch n - 1 is repeated in exact f32 (its range word was raised) behind batch n, batch n + 1's upload must not
    reuse that half of the arena before the repeat has read it.  Every batch is driven out of range here, many
    small batches are in flight, and each row must equal what the exact-f32 mode gives for ITS chunk."""
    import struct
    from buzzdetect_amd import pipeline as P
    from buzzdetect_amd.analyze import analyze
    from buzzdetect_amd.engine import HipEngine
    audio = tmp_path / "audio"
    audio.mkdir()
    # distinct content per chunk, so that rows computed from another chunk's audio cannot pass
    x = O.synthetic_audio(16000 * 400, seed=91).astype("<f4")
    x *= np.repeat(np.linspace(0.2, 1.0, 400, dtype=np.float32), 16000)
    fmt = struct.pack("<HHIIHH", 3, 1, 16000, 16000 * 4, 4, 32)
    (audio / "f32mono.wav").write_bytes(_riff(fmt, x.tobytes()))
    monkeypatch.setattr(P, "BATCH_WINDOWS", 12)                 # a batch = two 9.6 s chunks: ~ 21 batches back to back
    eng = HipEngine()
    try:
        exps, _ = eng.scales()
        bad = exps.copy()
        bad[6 - 2] += 14
        eng.set_pointwise_mode("f32")
        ref = analyze("model_general_v3", chunklength=9.6, dir_audio=str(audio), dir_out=str(tmp_path / "exact"), engine=eng)
        assert eng.overflow_reruns == 0 and ref.chunks == 42
        eng.set_pointwise_mode("f16x3")
        eng.set_activation_exponents(bad)
        rep = analyze("model_general_v3", chunklength=9.6, dir_audio=str(audio), dir_out=str(tmp_path / "repeated"), engine=eng)
        assert rep.chunks == ref.chunks and rep.windows == ref.windows == 42 * 10 - 10 + 7
        assert eng.overflow_reruns >= 10                        # (one per batch; the batching depends on the readers' pace)
        a = (tmp_path / "exact" / "f32mono_buzzdetect.csv").read_bytes()
        b = (tmp_path / "repeated" / "f32mono_buzzdetect.csv").read_bytes()
        assert a == b and a.count(b"\n") == rep.windows + 1
    finally:
        eng.close()


@pytest.mark.gpu
def test_pipeline_failure_in_any_stage_stops_everything(engine, tmp_path, monkeypatch):
    """A stage that raises poisons the pipeline: analyze() re-raises the first exception instead of hanging with the other
    stages blocked on their queues (the reference's known hole); an unreadable recording is skipped, not fatal."""
    import threading
    from buzzdetect_amd import results as R
    from buzzdetect_amd.analyze import analyze
    audio = tmp_path / "audio"
    _three_recordings(audio)
    (audio / "broken.wav").write_bytes(b"RIFF" + b"\0" * 6000)
    rep = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "ok"), engine=engine)
    assert rep.files_done == 3 and rep.files_skipped == 1 and any("broken.wav" in m for m in rep.messages)

    calls = {"n": 0}
    real = R.activation_csv

    def boom(*a, **k):
        calls["n"] += 1
        if calls["n"] == 3:
            raise OSError("disk full (injected)")
        return real(*a, **k)

    monkeypatch.setattr(R, "activation_csv", boom)
    out = {}

    def run():
        try:
            analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "bad"), engine=engine)
        except BaseException as exc:          # noqa: BLE001
            out["exc"] = exc

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "analyze() hung after a stage failed"
    assert isinstance(out.get("exc"), OSError) and "disk full" in str(out["exc"])
    monkeypatch.setattr(R, "activation_csv", real)

    def bad_engine():
        raise RuntimeError("no such device (injected)")

    from buzzdetect_amd.pipeline import FileJob, Pipeline
    pipe = Pipeline(make_engine=bad_engine, classes=engine.classes, framehop_s=0.96, hop=15360, step=96, chunklength=4.8,
                    framelength_s=0.96, digits_time=2, digits_results=2, classes_out="all", threshold=None, readers=2, analyzers=2)
    jobs = [FileJob(path=str(audio / "two.wav"), ident="two", shortpath="two.wav", rf=R.ResultFile(str(tmp_path / "x" / "two")))]
    with pytest.raises(RuntimeError, match="no such device"):
        pipe.run(jobs)


def _gather_rank(rank, world, port, audio, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the box; the collective runs over gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from buzzdetect_amd.analyze import analyze
        rep = analyze("model_general_v3", chunklength=5, dir_audio=audio, dir_out=out, gather_logits=True, analyzers_gpu=1)
        assert rep.files_total == 3
        assert rep.files_done == (2 if rank == 0 else 1)        # round-robin: recordings 0 and 2 / recording 1
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_analyze_gathers_logits_to_rank_0_world_size_2(engine, tmp_path):
    """BASELINE config 4 in miniature: two ranks, recordings dealt round-robin, one gather per round, rank 0 writes every
    result file - the same bytes a single process writes."""
    import socket
    import torch.multiprocessing as mp
    from buzzdetect_amd.analyze import analyze
    audio = tmp_path / "audio"
    _three_recordings(audio)
    analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "solo"), engine=engine)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gather_rank, args=(2, port, str(audio), str(tmp_path / "gathered")), nprocs=2, join=True)
    for rel in ("a/one", "two", "three32k"):
        a = (tmp_path / "solo" / f"{rel}_buzzdetect.csv").read_bytes()
        b = (tmp_path / "gathered" / f"{rel}_buzzdetect.csv").read_bytes()
        assert a == b
        assert not (tmp_path / "gathered" / f"{rel}_buzzpart.csv").exists()


@pytest.mark.gpu
def test_stop_event_ends_the_run_and_a_rerun_completes_it(engine, tmp_path, monkeypatch):
    """The reference's early exit (src/pipeline/coordination.py:182-188, analyze(event_stopanalysis=...)): setting the
    event releases every queue like a failure does, but analyze() RETURNS - end_reason "interrupted" - and what was
    written stays in the partial files, whole chunks only; the next run resumes from their coverage and ends with the
    bytes of an uninterrupted run."""
    import threading
    import time
    from buzzdetect_amd import results as R
    from buzzdetect_amd import wavio
    from buzzdetect_amd.analyze import analyze
    audio = tmp_path / "audio"
    _three_recordings(audio)
    ref = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "ref"), engine=engine,
                  stream_buffer_depth=1, analyzers_cpu=1)
    assert ref.end_reason == "completed" and ref.files_done == 3 and ref.chunks == 12

    stop = threading.Event()
    written = {"n": 0}
    real_append = R.ResultFile.append_text

    def counting_append(self, head, body):
        real_append(self, head, body)
        written["n"] += 1
        if written["n"] == 2:
            stop.set()                                  # "after the second chunk"

    from buzzdetect_amd import pipeline as P
    fast_read = P.ReaderStage.read                      # (16-bit PCM goes file -> device through the native stager)
    monkeypatch.setattr(R.ResultFile, "append_text", counting_append)
    monkeypatch.setattr(P.ReaderStage, "read", lambda self, fd, off, n, dev: (time.sleep(0.1), fast_read(self, fd, off, n, dev))[1])
    out = {}

    def run():
        try:
            out["rep"] = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "cut"),
                                 engine=engine, n_streamers=1, stream_buffer_depth=1, event_stopanalysis=stop)
        except BaseException as exc:                    # noqa: BLE001
            out["exc"] = exc

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "analyze() did not return after the stop event"
    assert "exc" not in out, out.get("exc")
    rep = out["rep"]
    assert rep.end_reason == "interrupted" and rep.files_done < 3
    assert 2 <= written["n"] < 12
    parts = [p for p in (tmp_path / "cut").rglob("*_buzzpart.csv")]
    assert parts, "nothing left to resume from"
    monkeypatch.setattr(R.ResultFile, "append_text", real_append)
    monkeypatch.setattr(P.ReaderStage, "read", fast_read)
    again = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "cut"), engine=engine)
    assert again.end_reason == "completed" and again.chunks == 12 - written["n"]
    for rel in ("a/one", "two", "three32k"):
        assert (tmp_path / "cut" / f"{rel}_buzzdetect.csv").read_bytes() == (tmp_path / "ref" / f"{rel}_buzzdetect.csv").read_bytes()
        assert not (tmp_path / "cut" / f"{rel}_buzzpart.csv").exists()
    # a stop event that is already set: nothing is analysed, nothing hangs
    early = analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "none"), engine=engine,
                    event_stopanalysis=stop)
    assert early.end_reason == "interrupted" and early.files_done == 0


@pytest.mark.gpu
def test_reference_logging_arguments(engine, tmp_path):
    """verbosity_log / log_progress (src/pipeline/logger.py:23-57): a time-stamped .log file in the output folder,
    PROGRESS lines only on request."""
    import re
    from buzzdetect_amd.analyze import analyze
    audio = tmp_path / "audio"
    _three_recordings(audio)
    for progress in (False, True):
        out = tmp_path / f"out{int(progress)}"
        analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(out), engine=engine,
                verbosity_log="DEBUG", log_progress=progress)
        logs = list(out.glob("*.log"))
        assert len(logs) == 1 and re.match(r"\d{4}-\d\d-\d\d_\d{6}\.log$", logs[0].name)
        text = logs[0].read_text()
        assert re.search(r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d\.\d{3} \[INFO\] planner: buffering", text, flags=re.M)
        assert ("[PROGRESS] analyzer 0: analyzed" in text) == progress
    import logging
    assert not logging.getLogger("buzzdetect").handlers           # the handlers are gone after the call


def _gather_rank_resume(rank, world, port, audio, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from buzzdetect_amd.analyze import analyze
        # rank 1 is given an output folder of its own in which NOTHING is complete: were every rank to plan for itself
        # (round 2's code), the two plans would differ (2 vs 3 recordings) and the gathers would mismatch or hang
        rep = analyze("model_general_v3", chunklength=5, dir_audio=audio, dir_out=out if rank == 0 else out + "_rank1",
                      gather_logits=True, analyzers_gpu=1)
        assert rep.files_done == 1                      # the two recordings left, one per rank
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gather_mode_resumes_with_rank_0s_plan(engine, tmp_path):
    import socket
    import torch.multiprocessing as mp
    from buzzdetect_amd.analyze import analyze
    audio = tmp_path / "audio"
    _three_recordings(audio)
    analyze("model_general_v3", chunklength=5, dir_audio=str(audio), dir_out=str(tmp_path / "solo"), engine=engine)
    out = tmp_path / "gathered"
    (out / "a").mkdir(parents=True)
    done = (tmp_path / "solo" / "two_buzzdetect.csv").read_bytes()
    (out / "two_buzzdetect.csv").write_bytes(done)                 # an earlier run finished this recording
    import shutil
    shutil.copy(tmp_path / "solo" / "buzzdetect_manifest.json", out / "buzzdetect_manifest.json")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gather_rank_resume, args=(2, port, str(audio), str(out)), nprocs=2, join=True)
    for rel in ("a/one", "two", "three32k"):
        assert (out / f"{rel}_buzzdetect.csv").read_bytes() == (tmp_path / "solo" / f"{rel}_buzzdetect.csv").read_bytes()


@pytest.mark.gpu
def test_config1_shape_on_the_device(weights_bundle, tmp_path):
    """BASELINE config 1 as far as this build can take it: the reference's getting-started run analyses
    audio_in/testbuzz.mp3 - 32 kHz mono, 3.82 s (docs/source/getting_started.rst:60-65) - with
    embedders/yamnet + model_general_v3 and ONE worker on the CPU.  MP3 decoding (PyAV / libsndfile) is out of scope and
    there is no CPU path, so the same shape runs here as a 32 kHz mono 16-bit WAV of that length through analyze() with the
    Keras-3 'yamnet' embedder's mel constant, one analyzer, one streamer, default chunk length: 4 rows (hazard H4: the last
    one mostly zero padding), the reference's columns, values against the f64 oracle of the same chain."""
    import pandas as pd
    from buzzdetect_amd.analyze import analyze
    rate, n = 32000, int(round(61144 * 8 / 128000 * 32000))          # 61 144 bytes at 128 kbit/s
    t = np.arange(n) / rate
    x = 0.25 * np.sin(2 * np.pi * 230 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * np.random.default_rng(4).standard_normal(n)
    audio = tmp_path / "audio_in"
    audio.mkdir()
    write_wav(audio / "testbuzz.wav", x, rate)
    rep = analyze("model_general_v3", classes_out="all", framehop_prop=1, chunklength=200, dir_audio=str(audio),
                  dir_out=str(tmp_path / "out"), embeddername="yamnet", analyzers_gpu=1, n_streamers=1, analyzers_cpu=1)
    assert (rep.files_done, rep.chunks, rep.windows) == (1, 1, 4)
    df = pd.read_csv(tmp_path / "out" / "testbuzz_buzzdetect.csv")
    assert df["start"].tolist() == [0.0, 0.96, 1.92, 2.88]
    assert list(df.columns) == ["start"] + [f"activation_{c}" for c in weights_bundle["classes"]]
    b = weights_bundle
    q = (np.clip(x, -1, 1 - 2 ** -15) * 32768.0).round().astype(np.int16).astype(np.float32) / 32768.0
    # the reference rounds chunk edges to two decimals (results_coverage.py:59-70) and truncates to samples
    # (src/stream/worker.py:110-111): the chunk is (0, 3.82) = samples [0, 122 240), 48 short of the file (hazard H3's kin)
    from buzzdetect_amd import framing
    (chunk,) = framing.gaps_to_chunklist([(0, n / rate)], framing.round_chunklength(200))
    a0, a1 = framing.chunk_sample_range(chunk, rate)
    assert (tuple(float(c) for c in chunk), a0, a1) == ((0.0, 3.82), 0, 122240)
    mono = RO.resample(q[a0:a1], rate).astype(np.float32)
    ref = O.predict(mono, b["blob"], b["mel_keras3"], b["head_kernel"], b["head_bias"], 15360, 96, np.float64)
    assert ref.shape == (4, 13)
    got = df[[f"activation_{c}" for c in b["classes"]]].to_numpy()
    assert np.abs(got - ref.round(2)).max() <= 0.011
