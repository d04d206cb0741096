"""`analyze()` on the MI355X engine: recordings in, reference-format result CSVs out.

Mirrors the call surface and on-disk behaviour of the reference's orchestration for the part that
surrounds the hot path:

    analyze(modelname, classes_out, precision, framehop_prop, chunklength, dir_audio, dir_out, ...)
                                                              src/analyze.py:387-492
    chunk length rounding, file idents, skip rules             src/analyze.py:102-111, :273-326
    chunk read, downmix, resample                              src/stream/worker.py:109-135
    result append / finalise, resume from coverage             src/write/worker.py:67-87, src/stream/worker.py:61-107
    output-folder manifest                                     src/pipeline/manifest.py:62-85

What differs by design: no worker threads and queues — one process per GPU walks its share of the
recordings (round-robin, ``sharding.shard_indices``) and keeps one chunk in flight on the device while the
previous chunk's rows are written; compressed formats are not decoded here (the reference uses
libsndfile / PyAV on the CPU, out of scope): inputs are PCM ``.wav`` files.
"""
from __future__ import annotations

import os
import re
import wave
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import framing, results, sharding

FILE_SIZE_MINIMUM = 5000          # src/config.py:20
BATCH_WINDOWS = 1024              # windows gathered per launch set
EXTENSIONS = (".wav",)


@dataclass
class AnalyzeReport:
    files_total: int = 0
    files_done: int = 0
    files_skipped: int = 0
    chunks: int = 0
    windows: int = 0
    audio_seconds: float = 0.0
    messages: List[str] = field(default_factory=list)


def build_ident(path: str, root_dir: str) -> str:
    """Path relative to the audio root, without extension (src/utils.py:51-62)."""
    ident = re.sub(re.escape(root_dir), "", path) if root_dir else path
    ident = os.path.splitext(ident)[0]
    return re.sub("^/", "", ident)


def search_audio(dir_audio: str) -> List[str]:
    out = []
    for root, _, files in os.walk(dir_audio):
        for f in files:
            if f.lower().endswith(EXTENSIONS):
                out.append(os.path.join(root, f))
    return sorted(out)


class WavTrack:
    """Frame-accurate reader with the soundfile-style contract the streamer relies on
    (seek(frame), read(n, float32) -> [n, channels] in [-1, 1), src/stream/audio.py:24-44)."""

    def __init__(self, path: str):
        self._w = wave.open(path, "rb")
        self.samplerate = self._w.getframerate()
        self.channels = self._w.getnchannels()
        self.frames = self._w.getnframes()
        self._width = self._w.getsampwidth()
        if self._width not in (1, 2, 3, 4):
            raise ValueError(f"{path}: unsupported sample width {self._width}")

    @property
    def duration(self) -> float:
        return self.frames / self.samplerate

    def seek(self, frame: int) -> None:
        self._w.setpos(min(max(frame, 0), self.frames))

    def read(self, n: int, keep_s16: bool = False) -> np.ndarray:
        """[frames, channels] float32 in [-1, 1); with ``keep_s16`` 16-bit files come back as int16 (the device
        stage scales by 1/32768 itself, halving the host-to-device bytes)."""
        raw = self._w.readframes(max(n, 0))
        if self._width == 2 and keep_s16:
            return np.frombuffer(raw, dtype="<i2").reshape(-1, self.channels)
        if self._width == 2:
            a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif self._width == 4:
            a = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif self._width == 1:
            a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            a = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        return a.reshape(-1, self.channels)

    def close(self) -> None:
        self._w.close()


def analyze(modelname: str = "model_general_v3", classes_out="all", precision: Optional[float] = None,
            framehop_prop: float = 1, chunklength: float = 200, dir_audio: str = "audio_in",
            dir_out: Optional[str] = None, embeddername: str = "yamnet_k2", engine=None,
            rank: Optional[int] = None, world_size: Optional[int] = None) -> AnalyzeReport:
    """Analyse every ``.wav`` under ``dir_audio``; write ``<ident>_buzzdetect.csv`` under ``dir_out``.

    ``classes_out`` / ``precision`` choose activations vs detections exactly as in the reference;
    ``rank`` / ``world_size`` default to the torch.distributed environment (one process per GPU)."""
    from .engine import HipEngine, hop_samples, patch_step   # device code is only needed once there is work to do

    report = AnalyzeReport()
    if rank is None or world_size is None:
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank, world_size = dist.get_rank(), dist.get_world_size()
        except ImportError:
            pass
    rank = rank or 0
    world_size = world_size or 1
    dir_out = dir_out or os.path.join("models", modelname, "output")

    eng = engine or HipEngine(embeddername=embeddername, modelname=modelname)
    classes = eng.classes
    framelength_s, digits_time, digits_results = 0.96, 2, 2
    framehop_s = framelength_s * framehop_prop
    chunklength = framing.round_chunklength(chunklength, framelength_s, digits_time)
    if classes_out == "all":
        classes_out = list(classes)
    threshold = None if precision is None else results.threshold_for_precision(modelname, precision)

    manifest = results.build_manifest(modelname, framehop_prop, precision, classes_out)
    ok, msg = (True, None)
    if rank == 0:
        ok, msg = results.check_or_write_manifest(dir_out, manifest)
    if not ok:
        raise RuntimeError(msg)

    paths = search_audio(dir_audio)
    idents = [build_ident(p, dir_audio) for p in paths]
    conflicting = {i for i in idents if idents.count(i) > 1}
    todo = [(p, i) for p, i in zip(paths, idents) if i not in conflicting]
    report.files_total = len(todo)
    mine = [todo[k] for k in sharding.shard_indices(len(todo), rank, world_size)]

    def table_for(logits: np.ndarray, time_start: float):
        if threshold is None:
            return results.activation_table(logits, classes, framehop_s, digits_time, time_start, classes_out,
                                            digits_results)
        return results.detection_table(logits, threshold, classes, framehop_s, digits_time, time_start)

    for path, ident in mine:
        rf = results.ResultFile(os.path.join(dir_out, ident))
        if rf.complete or os.path.getsize(path) < FILE_SIZE_MINIMUM:
            report.files_skipped += 1
            continue
        track = WavTrack(path)
        try:
            chunks = rf.pending_chunks(track.duration, chunklength, framelength_s)
            # Chunks are gathered into batches of up to ~BATCH_WINDOWS windows (<= 64 chunks) and go through
            # ONE launch set (bd_predict_batch); every chunk keeps its own end-of-chunk padding, so the rows are
            # those of one predict() per chunk.  The previous batch's rows are written while this one runs.
            batch, batch_windows = [], 0
            in_flight = None                                  # (list of DeviceResult, list of chunks)

            def flush():
                nonlocal batch, batch_windows, in_flight
                if not batch:
                    return
                res = eng.predict_batch([pcm for _, pcm in batch], framehop_s)
                if in_flight is not None:
                    for r, c in zip(*in_flight):
                        rf.append(table_for(r.numpy(), c[0]))
                in_flight = (res, [c for c, _ in batch])
                report.windows += sum(len(r) for r in res)
                batch, batch_windows = [], 0

            stop = False
            for chunk in chunks:
                a, b = framing.chunk_sample_range(chunk, track.samplerate)
                track.seek(a)
                samples = track.read(b - a, keep_s16=True)
                if samples.shape[0] < b - a:                 # short read: truncate the chunk, finish the file
                    chunk = (chunk[0], round(chunk[0] + samples.shape[0] / track.samplerate, 1))
                    stop = True
                if samples.shape[0] == 0:
                    break
                if track.samplerate != 16000 or track.channels > 1 or samples.dtype == np.int16:
                    pcm = eng.resample(samples, track.samplerate, 16000)     # also the s16 -> f32 conversion
                else:
                    pcm = samples[:, 0]
                batch.append((chunk, pcm))
                batch_windows += eng.num_windows(len(pcm), hop_samples(framehop_s), patch_step(framehop_s))
                report.chunks += 1
                report.audio_seconds += float(chunk[1] - chunk[0])
                if batch_windows >= BATCH_WINDOWS or len(batch) == 64:
                    flush()
                if stop:
                    break
            flush()
            if in_flight is not None:
                for r, c in zip(*in_flight):
                    rf.append(table_for(r.numpy(), c[0]))
            if os.path.exists(rf.path_partial):
                rf.finalize()
            report.files_done += 1
        finally:
            track.close()
    for ident in sorted(conflicting):
        report.messages.append(f"conflicting names, skipped: {ident}")
    return report
