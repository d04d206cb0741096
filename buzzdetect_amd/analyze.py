"""`analyze()` on the MI355X engine: recordings in, reference-format result CSVs out.

Mirrors the call surface and on-disk behaviour of the reference's orchestration for the part that
surrounds the hot path:

    analyze(modelname, classes_out, precision, framehop_prop, chunklength, dir_audio, dir_out, ...)
                                                              src/analyze.py:387-492
    chunk length rounding, file idents, skip rules             src/analyze.py:102-111, :273-326
    chunk read, downmix, resample                              src/stream/worker.py:109-135
    result append / finalise, resume from coverage             src/write/worker.py:67-87, src/stream/worker.py:61-107
    output-folder manifest                                     src/pipeline/manifest.py:62-85
    streamers -> bounded queue -> analyzers -> writer          src/pipeline/coordination.py:26-194  (buzzdetect_amd/pipeline.py)

One process per GPU walks its share of the recordings (round-robin, ``sharding.shard_indices``).  Compressed formats
are not decoded here (the reference uses libsndfile / PyAV on the CPU, out of scope): inputs are uncompressed ``.wav``.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional

from . import framing, results, sharding
from .pipeline import FILE_SIZE_MINIMUM, FileJob, Pipeline, Report
from .pipeline import log as _log
from .wavio import WavTrack  # noqa: F401  (re-exported: the streamer's reader)

AnalyzeReport = Report
EXTENSIONS = (".wav",)
STREAMERS_PER_ANALYZER = 3        # the reference runs 8 decoding streamers per GPU analyzer (coordination.py:129-138);
                                  # reading uncompressed PCM (~8 GB/s per thread) needs fewer


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


def analyze(modelname: str = "model_general_v3", classes_out="all", precision: Optional[float] = None,
            framehop_prop: float = 1, chunklength: float = 200, dir_audio: str = "audio_in",
            dir_out: Optional[str] = None, embeddername: str = "yamnet_k2", engine=None,
            rank: Optional[int] = None, world_size: Optional[int] = None, analyzers_gpu: int = 2,
            n_streamers: Optional[int] = None, engines: Optional[list] = None,
            gather_logits: bool = False, analyzers_cpu: int = 0, stream_buffer_depth: Optional[int] = None,
            verbosity_print: Optional[str] = None, verbosity_log: Optional[str] = None, log_progress: bool = False,
            event_stopanalysis=None) -> AnalyzeReport:
    """Analyse every ``.wav`` under ``dir_audio``; write ``<ident>_buzzdetect.csv`` under ``dir_out``.

    ``classes_out`` / ``precision`` choose activations vs detections exactly as in the reference;
    ``rank`` / ``world_size`` default to the torch.distributed environment (one process per GPU);
    ``analyzers_gpu`` analyzer threads (each with its own engine and HIP stream) are fed by ``n_streamers`` reader
    threads.  ``engine``: use this engine on ONE analyzer thread instead (tests, embedding in another loop);
    ``engines``: prebuilt engines, one per analyzer thread (a caller that analyses folder after folder keeps them: building
    an engine folds and uploads the weights, ~0.1 s).
    ``gather_logits`` (several ranks): BASELINE config 4 - every rank analyses its recordings, the per-window logits
    are gathered to rank 0 over RCCL once per round of ``world_size`` recordings, and rank 0 alone writes the result
    files (for output folders only rank 0 can reach; without it every rank writes its own files).
    The remaining keyword arguments are the reference's (src/analyze.py:387-404): ``stream_buffer_depth`` chunks may wait
    between streamers and analyzers (default 2 x streamers); ``event_stopanalysis`` (anything with ``is_set()``) ends the
    analysis early - every queue is released, the report comes back with ``end_reason == "interrupted"`` and the partial
    result files are left for the next run to resume; ``verbosity_print`` / ``verbosity_log`` / ``log_progress`` attach
    the reference's console and ``<dir_out>/<timestamp>.log`` handlers to logger ``buzzdetect`` for the duration of the
    call (None: leave logging to the caller); ``analyzers_cpu`` is accepted and ignored - this engine has no CPU path."""
    from .engine import HipEngine, hop_samples, patch_step   # device code is only needed once there is work to do

    dist = None
    if rank is None or world_size is None:
        try:
            import torch.distributed as dist_mod
            if dist_mod.is_available() and dist_mod.is_initialized():
                dist = dist_mod
                rank, world_size = dist.get_rank(), dist.get_world_size()
        except ImportError:
            pass
    rank = rank or 0
    world_size = world_size or 1
    dir_out = dir_out or os.path.join("models", modelname, "output")
    handlers = _attach_log_handlers(dir_out, verbosity_print, verbosity_log, log_progress)
    try:
        return _analyze(modelname, classes_out, precision, framehop_prop, chunklength, dir_audio, dir_out, embeddername,
                        engine, rank, world_size, dist, analyzers_gpu, n_streamers, engines, gather_logits, analyzers_cpu,
                        stream_buffer_depth, event_stopanalysis)
    finally:
        for h in handlers:
            _log.removeHandler(h)
            h.close()


def _attach_log_handlers(dir_out: str, verbosity_print, verbosity_log, log_progress: bool) -> list:
    """The reference's logger set-up (src/pipeline/logger.py:23-57, src/analyze.py:127-136) on logger ``buzzdetect``."""
    import logging
    import time
    from .pipeline import PROGRESS
    levels = {"NOTSET": logging.NOTSET, "DEBUG": logging.DEBUG, "PROGRESS": PROGRESS, "INFO": logging.INFO,
              "WARNING": logging.WARNING, "ERROR": logging.ERROR, "CRITICAL": logging.CRITICAL}

    class PeriodFormatter(logging.Formatter):
        def formatTime(self, record, datefmt=None):
            return f"{time.strftime('%Y-%m-%d %H:%M:%S', self.converter(record.created))}.{int(record.msecs):03d}"

    out = []
    fmt = "%(asctime)s [%(levelname)s] %(message)s"
    if verbosity_log is not None:
        os.makedirs(dir_out, exist_ok=True)
        h = logging.FileHandler(os.path.join(dir_out, time.strftime("%Y-%m-%d_%H%M%S") + ".log"))
        h.setLevel(levels[verbosity_log])
        if not log_progress:
            h.addFilter(lambda record: record.levelno != PROGRESS)
        h.setFormatter(PeriodFormatter(fmt))
        out.append(h)
    if verbosity_print is not None:
        h = logging.StreamHandler()
        h.setLevel(levels[verbosity_print])
        h.setFormatter(PeriodFormatter(fmt))
        out.append(h)
    if out:
        _log.setLevel(logging.DEBUG)
    for h in out:
        _log.addHandler(h)
    return out


def _analyze(modelname, classes_out, precision, framehop_prop, chunklength, dir_audio, dir_out, embeddername, engine, rank,
             world_size, dist, analyzers_gpu, n_streamers, engines, gather_logits, analyzers_cpu, stream_buffer_depth,
             event_stopanalysis) -> AnalyzeReport:
    from .engine import HipEngine, hop_samples, patch_step
    if analyzers_cpu:
        _log.debug(f"analyzers_cpu={analyzers_cpu} ignored: the MI355X engine has no CPU path")

    framelength_s, digits_time, digits_results = 0.96, 2, 2
    framehop_s = framelength_s * framehop_prop
    chunklength = framing.round_chunklength(chunklength, framelength_s, digits_time)
    if engines:
        engine = None
    probe = engine or (engines[0] if engines else HipEngine(embeddername=embeddername, modelname=modelname))
    classes = probe.classes
    device_index = probe.device_index
    if classes_out == "all":
        classes_out = list(classes)
    threshold = None if precision is None else results.threshold_for_precision(modelname, precision)

    # The manifest locks a results folder to one set of settings.  Rank 0 writes (or checks) it; the others wait for
    # that, then every rank validates the folder for itself, so a conflict stops ALL ranks before any row is written.
    manifest = results.build_manifest(modelname, framehop_prop, precision, classes_out)
    ok, msg = (True, None)
    if rank == 0:
        ok, msg = results.check_or_write_manifest(dir_out, manifest)
    if dist is not None and world_size > 1:
        dist.barrier()
    if rank != 0:
        ok, msg = results.check_or_write_manifest(dir_out, manifest)
    if not ok:
        raise RuntimeError(msg)

    paths = search_audio(dir_audio)
    idents = [build_ident(p, dir_audio) for p in paths]
    conflicting = {i for i in idents if idents.count(i) > 1}
    todo = [(p, i) for p, i in zip(paths, idents) if i not in conflicting]
    mine = [todo[k] for k in sharding.shard_indices(len(todo), rank, world_size)]
    if engine is not None:
        analyzers, make_engine = 1, (lambda: engine)
    elif engines:
        analyzers, pool = len(engines), list(engines)
        pool_lock = __import__("threading").Lock()

        def make_engine():
            with pool_lock:
                return pool.pop()
    else:
        analyzers = max(1, int(analyzers_gpu))
        first = [probe]

        def make_engine():        # the probe engine serves the first analyzer thread; the others build their own
            if first:
                return first.pop()
            return HipEngine(embeddername=embeddername, modelname=modelname, device=device_index)

    readers = n_streamers if n_streamers else STREAMERS_PER_ANALYZER * analyzers
    common = dict(make_engine=make_engine, classes=classes, framehop_s=framehop_s, hop=hop_samples(framehop_s),
                  step=patch_step(framehop_s), chunklength=chunklength, framelength_s=framelength_s,
                  digits_time=digits_time, digits_results=digits_results, classes_out=classes_out, threshold=threshold,
                  readers=readers, analyzers=analyzers, stop_event=event_stopanalysis,
                  stream_buffer_depth=stream_buffer_depth, device=probe.device)

    if gather_logits and dist is not None and world_size > 1:
        report = _analyze_gathered(todo, dir_out, rank, world_size, probe, common, classes, framehop_s, chunklength,
                                   digits_time, digits_results, classes_out, threshold)
    else:
        jobs = [FileJob(path=p, ident=i, shortpath=i + os.path.splitext(p)[1], rf=results.ResultFile(os.path.join(dir_out, i)))
                for p, i in mine]
        report = Pipeline(**common).run(jobs)
    report.files_total = len(todo)
    for ident in sorted(conflicting):
        report.messages.append(f"conflicting names, skipped: {ident}")
    return report


def gather_plan(todo, dir_out: str, hop: int, step: int, chunklength: float) -> list:
    """What config 4's gathers move: for every recording still to analyse (path, ident, chunks, windows per chunk), in
    the order of ``todo``.  It depends on the planner's view of ``dir_out`` (finished recordings are left out), so ONE
    rank computes it and the others receive it (``broadcast_plan``): block sizes, round count and ownership must agree
    everywhere or the collectives mismatch."""
    from . import _lib
    from .pipeline import log
    from .wavio import WavFormatError
    lib = _lib.load()
    plan = []
    for path, ident in todo:
        rf = results.ResultFile(os.path.join(dir_out, ident))
        if rf.complete or os.path.getsize(path) < FILE_SIZE_MINIMUM:
            continue
        try:
            track = WavTrack(path)
        except (WavFormatError, OSError) as exc:
            log.warning(f"planner: {exc}; skipping")
            continue
        chunks, counts = [], []
        for chunk in framing.gaps_to_chunklist([(0, track.duration)], chunklength):
            a, b = framing.chunk_sample_range(chunk, track.samplerate)
            frames = min(b, track.frames) - a
            if frames <= 0:
                continue
            n16 = _lib.check(lib.bd_resample_length(frames, track.samplerate, 16000))
            chunks.append((float(chunk[0]), float(chunk[1])))
            counts.append(_lib.check(lib.bd_num_windows(n16, hop, step)))
        track.close()
        if chunks:
            plan.append((path, ident, chunks, counts))
    return plan


def broadcast_plan(plan, rank: int, src: int = 0, group=None):
    """The plan of rank ``src`` on every rank (``broadcast_object_list``; a few hundred bytes per recording)."""
    import torch.distributed as dist
    box = [plan if rank == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def _analyze_gathered(todo, dir_out, rank, world_size, probe, common, classes, framehop_s, chunklength, digits_time,
                      digits_results, classes_out, threshold) -> Report:
    """Config 4: round-robin recordings, one RCCL gather of the logit blocks per round, rank 0 writes every file.
    Rank 0 plans (which recordings, how many rows each: it is the rank that sees the output folder) and broadcasts."""
    import numpy as np
    from .pipeline import log
    hop, step = common["hop"], common["step"]
    plan = broadcast_plan(gather_plan(todo, dir_out, hop, step, chunklength) if rank == 0 else None, rank)
    rows_per_file = [sum(c) for _, _, _, c in plan]

    def write_file(index: int, rows: "np.ndarray") -> None:          # rank 0, from its writer thread
        path, ident, chunks, counts = plan[index]
        rf = results.ResultFile(os.path.join(dir_out, ident))
        os.makedirs(os.path.dirname(rf.path_complete) or ".", exist_ok=True)
        at, parts, head = 0, [], b""
        for chunk, n in zip(chunks, counts):
            block = rows[at:at + n]
            at += n
            if threshold is None:
                head, body = results.activation_csv(block, classes, framehop_s, digits_time, chunk[0], classes_out, digits_results)
            else:
                head, body = results.detection_csv(block, threshold, classes, framehop_s, digits_time, chunk[0])
            parts.append(body)
        with open(rf.path_complete + ".tmp", "wb") as f:
            f.write(head)
            f.writelines(parts)
        os.replace(rf.path_complete + ".tmp", rf.path_complete)
        if os.path.exists(rf.path_partial):
            os.remove(rf.path_partial)

    import torch.distributed as dist
    # RCCL moves device tensors; a gloo group (rehearsals on one GPU, CPU tests) moves host tensors
    gather_device = "cpu" if dist.get_backend() == "gloo" else probe.device
    gatherer = sharding.RoundGatherer(rows_per_file, len(classes), write_file, device=gather_device)

    def sink(job: FileJob, rows) -> None:
        block = np.concatenate([r for _, r in rows]) if rows else np.zeros((0, len(classes)), np.float32)
        gatherer.submit(job.index, block)

    jobs = []
    for k in sharding.shard_indices(len(plan), rank, world_size):
        path, ident, _, _ = plan[k]
        jobs.append(FileJob(path=path, ident=ident, shortpath=ident + os.path.splitext(path)[1],
                            rf=results.ResultFile(os.path.join(dir_out, ident)), index=k))
    error = None
    try:
        report = Pipeline(**common, file_sink=sink, ignore_partial=True).run(jobs)
    except BaseException as exc:              # noqa: BLE001 - the collectives below must still be issued
        error, report = exc, Report(files_total=len(jobs))
    # a recording this rank owns and did not deliver (skipped as unreadable, or the pipeline failed): every round's
    # collective is issued all the same, with the status row set, so that no other rank is left waiting in it
    for job in jobs:
        if job.index not in gatherer.submitted:
            gatherer.submit_failed(job.index, "pipeline failed" if error is not None else "skipped")
    failed = gatherer.finish()
    for f in failed:
        msg = f"not delivered by rank {sharding.owner_of(f, world_size)}, no result file written: {plan[f][1]}"
        log.warning(f"writer: {msg}")
        report.messages.append(msg)
    if error is not None:
        raise error
    return report
