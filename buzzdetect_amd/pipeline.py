"""The feeder around the hot path: reader threads -> bounded queue -> analyzer threads -> writer thread.

Same shape and semantics as the reference's worker pipeline (src/pipeline/coordination.py:26-194,
src/stream/worker.py:109-165, src/inference/worker.py:9-92, src/write/worker.py:67-87), built so that an engine
that consumes ~45 MB/s of 16-bit PCM per 10 k windows/s is not left waiting:

    planner     walks the recordings: skip rules, header, chunk list (resume aware) -> read units.
    readers     `readers` threads take read units of ANY recording (a single 24 h file is read by all of them):
                positioned reads in pieces of 8 MB into the reader's OWN pair of small pinned buffers (16-bit PCM as
                it lies in the file, anything else converted to float32), each piece sent on by an async H2D copy on
                the reader's own stream while the next piece is read: a chunk is DEVICE-resident (a slot of a pool of
                device buffers) when it enters the bounded queue (depth 2 x readers, coordination.py:84-102).  Only
                readers x 16 MB of host memory is ever page-locked, whatever the chunk length (round 6: the ring of
                chunk-sized pinned slots of round 5 page-locked ~1 GB inside the first call).  Reads release the GIL.
    analyzers   `analyzers` threads per GPU (the reference's analyzers_gpu; docs/source/tuning.rst:111), each
                constructing and initialising ITS OWN engine in-thread (src/inference/worker.py:21,78) on its own
                HIP stream: wait for the chunk's copy event, device-side downmix / resample / s16 -> f32, up to 64
                chunks (~4096 windows) per launch set, async D2H of the logits into pinned memory.
    writer      waits for the batch's event, formats rows (fastcsv: the bytes pandas would write), appends to the
                partial file (resume safety), finalises a recording when its last chunk has been written: a
                recording started from nothing is written out sorted from the rows kept in memory - what the
                reference's read-sort-rewrite produces, without pandas.
    logging     logger "buzzdetect", level PROGRESS = INFO - 5 (src/pipeline/loglevels.py): the reference's two
                analyzer lines, "analyzed <file>, chunk (a, b) in <t>s (rate: <r>)" and
                "BUFFER BOTTLENECK: analyzer <id> received assignment after <t>s".

An exception in ANY stage poisons the pipeline: the first one is kept, every queue is released, `run` re-raises it
after all threads have ended (the reference's workers die silently and leave the others blocked).
The caller's `stop_event` (the reference's early exit, src/pipeline/coordination.py:182-188: poison every queue) ends a
run the same way but without an error: `run` returns its report with end_reason "interrupted"; what has been written
stays in the partial result files, whole chunks only, and the next run resumes from their coverage.
"""
from __future__ import annotations

import logging
import os
import queue
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import framing, results
from .wavio import WavFormatError, WavTrack

PROGRESS = logging.INFO - 5
logging.addLevelName(PROGRESS, "PROGRESS")
log = logging.getLogger("buzzdetect")

EXIT = "exit"
BATCH_WINDOWS = 4096              # windows gathered per launch set (when that many are waiting): the engine walks a set in
                                  # passes of 1024 windows, so a set of one 600 s chunk (625 / 1249 windows) ends in a pass that
                                  # is a fifth full; four passes' worth leaves at most one partial pass in five
BATCH_CHUNKS = 64                 # bd_predict_batch's limit
FILE_SIZE_MINIMUM = 5000          # src/config.py:20
BOTTLENECK_SECONDS = 0.01         # src/inference/worker.py:86
BAD_READ_ALLOWANCE = 0.01         # src/config.py:18: share of a file's tail that may be unreadable before it is a WARNING
RESULT_BLOCKS = 6                 # pinned result blocks per analyzer: batches in flight between the GPU and the writer
STAGE_BYTES = 8 << 20             # a reader's pinned staging buffer: a chunk travels to the device in pieces of this size
STAGE_BUFFERS = 2                 # ... and the reader fills one while the copy out of the other is in flight (bd_stager_*)


class PipelineAborted(Exception):
    """Raised inside a worker that finds the pipeline poisoned by another stage."""


@dataclass
class FileJob:
    path: str
    ident: str
    shortpath: str
    rf: results.ResultFile
    fresh: bool = True            # no partial results existed when the planner opened it
    outstanding: int = 0          # chunks planned and not yet written (or dropped)
    track: Optional[WavTrack] = None
    header: bytes = b""
    bodies: List[Tuple[float, bytes]] = field(default_factory=list)      # (chunk start, rows) of a fresh recording
    bad_read: bool = False        # the file ended before its header said it would: said once (handle_bad_read)
    index: int = -1               # position in the run's recording list (gather mode)
    rows: List[Tuple[float, "np.ndarray"]] = field(default_factory=list)  # (chunk start, logits) when a file sink takes the rows


@dataclass
class ReadUnit:
    job: FileJob
    chunk: Tuple[float, float]


@dataclass
class ChunkTask:
    job: FileJob
    chunk: Tuple[float, float]
    slot: int
    nbytes: int
    frames: int
    channels: int
    rate: int
    s16: bool
    ready: object = None          # torch.cuda.Event behind the last host-to-device copy of the chunk (None: host-only stage)


@dataclass
class WriteItem:
    tasks: List[ChunkTask]
    host: "np.ndarray"            # [total windows, classes] pinned, filled when `done` has happened
    counts: List[int]
    done: object                  # torch.cuda.Event
    analyzer: int
    t_start: float
    block: object = None          # the pinned result block `host` is a view of; goes back to its analyzer's pool when written


@dataclass
class Report:
    files_total: int = 0
    files_done: int = 0
    files_skipped: int = 0
    chunks: int = 0
    windows: int = 0
    audio_seconds: float = 0.0
    messages: List[str] = field(default_factory=list)
    # wall seconds each stage spent doing its work (summed over the stage's threads; waiting on a queue is not work):
    # read = file -> pinned staging buffer (+ waiting for the copy that still reads it), pin = page-locking the readers'
    # staging buffers, analyze = enqueueing a batch's kernels, settle = waiting for a batch's range verdict, write_wait = the writer waiting for a batch's event,
    # format = rows -> CSV text, write = appending it to the result files
    busy: Dict[str, float] = field(default_factory=dict)
    end_reason: str = "completed"     # or "interrupted": the caller's stop event ended the run (coordination.py:147-154)


class ChunkPool:
    """A fixed number of DEVICE buffers, one per chunk in flight: a reader fills a slot piece by piece through its pinned
    staging pair, the kernels read the chunk where it lies, and the writer gives the slot back once the batch's rows have
    been read (so an exact-f32 repeat of a batch still finds its audio).  Buffers grow to the largest chunk seen.  The
    free list is a STACK: the slot given back last goes out first, so only as many buffers exist as are in flight at
    once.  `device` None: plain host buffers (a reader stage without a device, tests/test_analyze.py)."""

    def __init__(self, slots: int, device=None):
        import torch
        self._torch, self.device = torch, device
        self._free: "queue.LifoQueue[int]" = queue.LifoQueue()
        self._buf: List[Optional["torch.Tensor"]] = [None] * slots
        for i in reversed(range(slots)):
            self._free.put(i)

    def acquire(self, nbytes: int, aborted: threading.Event):
        while True:
            if aborted.is_set():
                raise PipelineAborted()
            try:
                slot = self._free.get(timeout=0.2)
                break
            except queue.Empty:
                continue
        buf = self._buf[slot]
        if buf is None or buf.numel() < nbytes:
            # (every kernel that read the old buffer finished before its slot came back: nothing is in flight on it)
            self._buf[slot] = buf = self._torch.empty(max(nbytes, 1 << 20), dtype=self._torch.uint8, device=self.device)
        return slot, buf

    def buffer(self, slot: int):
        return self._buf[slot]

    def release(self, slot: int) -> None:
        self._free.put(slot)


class ReaderStage:
    """One reader thread's conduit to the device: a native stager (bd_stager_*: STAGE_BUFFERS page-locked buffers of
    STAGE_BYTES, pread -> hipMemcpyAsync piece by piece with the interpreter lock released for the whole chunk) and a HIP
    stream of its own.  Page-locking costs 0.07-0.25 s per GB and is serialised in the driver (tools/pin_probe.py): these
    16 MB cost ~1 ms."""

    def __init__(self, torch, device):
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        t0 = time.perf_counter()
        self.handle = ctypes.c_void_p()
        self.device_index = device.index or 0
        _lib.check(self._lib.bd_stager_create(ctypes.byref(self.handle), self.device_index, STAGE_BYTES, STAGE_BUFFERS))
        self.pin_seconds = time.perf_counter() - t0
        self.stream = torch.cuda.Stream(device)

    def read(self, fd: int, offset: int, nbytes: int, dev) -> int:
        """File bytes -> device tensor `dev` (uint8) on this stage's stream; returns the bytes read and enqueued."""
        from . import _lib
        return _lib.check(self._lib.bd_stager_read(self.handle, fd, offset, nbytes, dev.data_ptr(), self.stream.cuda_stream))

    def send(self, fill: Callable[["np.ndarray"], int], dev_ptr: int) -> int:
        """One piece through the next free buffer: `fill(buffer)` writes it and returns its byte count."""
        import ctypes
        from . import _lib
        index, host = ctypes.c_int32(), ctypes.c_void_p()
        _lib.check(self._lib.bd_stager_acquire(self.handle, ctypes.byref(index), ctypes.byref(host)))
        buf = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(STAGE_BYTES,))
        n = fill(buf)
        _lib.check(self._lib.bd_stager_submit(self.handle, index.value, n, dev_ptr, self.stream.cuda_stream))
        return n

    def close(self) -> None:
        if self.handle:
            self._lib.bd_stager_destroy(self.handle)
            self.handle = None

    # Stages outlive a run: page-locking and un-locking their buffers costs a few milliseconds each, serialised in the driver,
    # which a process that analyses folder after folder would pay at both ends of every call.  A finished run hands its stages
    # back (every copy out of them has completed by then); the next run's readers take them from here.
    _idle: "Dict[int, List[ReaderStage]]" = {}
    _idle_lock = threading.Lock()

    @classmethod
    def take(cls, torch, device) -> "ReaderStage":
        with cls._idle_lock:
            idle = cls._idle.get(device.index or 0)
            if idle:
                st = idle.pop()
                st.pin_seconds = 0.0
                return st
        return cls(torch, device)

    def give_back(self) -> None:
        with ReaderStage._idle_lock:
            ReaderStage._idle.setdefault(self.device_index, []).append(self)


class EventPool:
    """HIP events, recycled.  Every batch needs a "results are on the host" event; a stream of freshly created events
    makes the HIP runtime grow its signal pool while the GPU is busy - a one-off stall of tens of milliseconds in the
    middle of a run (tools/stall_probe.py) - so the writer hands each event back once it has waited for it."""

    def __init__(self, torch):
        self._torch = torch
        self._free: "queue.SimpleQueue" = queue.SimpleQueue()

    def take(self):
        try:
            return self._free.get_nowait()
        except queue.Empty:
            return self._torch.cuda.Event()

    def give(self, event) -> None:
        self._free.put(event)


class ResultPool:
    """The pinned [rows, classes] float32 blocks one analyzer's logits land in, allocated ONCE when the analyzer starts
    (page-locking memory per batch cost the first call of analyze() a third of its time).  A block
    travels analyzer -> writer inside its WriteItem and comes back when its rows have been formatted; with all blocks out
    the analyzer waits for the writer - the bound on results in flight."""

    def __init__(self, torch, blocks: int, rows: int, classes: int):
        self._torch, self._classes, self.rows = torch, classes, rows
        self._free: "queue.SimpleQueue" = queue.SimpleQueue()
        for _ in range(blocks):
            self._free.put(torch.empty((rows, classes), dtype=torch.float32, pin_memory=True))

    def take(self, rows: int, aborted: threading.Event):
        while True:
            if aborted.is_set():
                raise PipelineAborted()
            try:
                block = self._free.get(timeout=0.2)
                break
            except queue.Empty:
                continue
        if block.shape[0] < rows:                      # a batch larger than planned (never with the sizes Pipeline derives)
            log.debug(f"result block grows to {rows} rows")
            block = self._torch.empty((rows, self._classes), dtype=self._torch.float32, pin_memory=True)
        return block

    def give(self, block) -> None:
        self._free.put(block)


class DeviceArena:
    """Two grow-only device buffers of one analyzer, used alternately (the 16 kHz PCM of batch n and of batch n + 1): written
    and read on the analyzer's one stream only, so a buffer is reused two batches later in stream order - no allocation in
    the steady state."""

    def __init__(self, torch, device, dtype):
        self._torch, self._device, self._dtype = torch, device, dtype
        self._buf = [None, None]

    def get(self, which: int, count: int, stream):
        buf = self._buf[which]
        if buf is None or buf.numel() < count:
            if buf is not None:
                buf.record_stream(stream)
            self._buf[which] = buf = self._torch.empty(max(int(count * 1.25), 1 << 20), dtype=self._dtype, device=self._device)
        return buf


class Pipeline:
    def __init__(self, *, make_engine: Callable[[], object], classes: Sequence[str], framehop_s: float, hop: int, step: int,
                 chunklength: float, framelength_s: float, digits_time: int, digits_results: int, classes_out,
                 threshold: Optional[float], readers: int = 4, analyzers: int = 2, device=None,
                 file_sink: Optional[Callable[[FileJob, List[Tuple[float, "np.ndarray"]]], None]] = None,
                 ignore_partial: bool = False, stop_event=None, stream_buffer_depth: Optional[int] = None,
                 pin_memory: bool = True, resample_quality: int = 1):
        import torch
        self.torch = torch
        self.make_engine = make_engine
        self.classes, self.classes_out, self.threshold = list(classes), classes_out, threshold
        self.framehop_s, self.hop, self.step = framehop_s, hop, step
        self.chunklength, self.framelength_s = chunklength, framelength_s
        self.digits_time, self.digits_results = digits_time, digits_results
        self.n_readers, self.n_analyzers = max(1, readers), max(1, analyzers)
        # file_sink: instead of writing result files, hand every finished recording's rows (sorted by chunk start) to this
        # callable from the writer thread (the multi-GPU gather: rank 0 writes what the other ranks computed)
        self.file_sink, self.ignore_partial = file_sink, ignore_partial
        self.q_files: "queue.Queue" = queue.Queue()
        self.q_units: "queue.Queue" = queue.Queue(maxsize=8 * self.n_readers)
        # chunks buffered between streamers and analyzers: the reference's stream_buffer_depth, by default 2 x streamers
        # (coordination.py:129-138)
        depth = int(stream_buffer_depth) if stream_buffer_depth else 2 * self.n_readers
        self.q_analyze: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.stop_event = stop_event          # anything with is_set(): threading.Event, multiprocessing.Event
        self.q_write: "queue.Queue" = queue.Queue()
        # chunks in flight: queue + readers' hands + batches between the analyzers and the writer.  They live in DEVICE
        # memory; `pin_memory` False (no device: reader-stage tests) keeps them in host memory
        if pin_memory and device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device if pin_memory else None
        self.pool = ChunkPool(max(1, depth) + 2 * self.n_readers + 16 * self.n_analyzers, self.device)
        self._stage = threading.local()
        self._stages: List[ReaderStage] = []             # every reader's stage, closed when the run has ended
        self.resample_quality, self._rates = resample_quality, {}     # 1: "hq", HipEngine's default (RESAMPLE_QUALITIES)
        self.events = EventPool(torch)
        self.aborted = threading.Event()
        self.error: Optional[BaseException] = None
        self.lock = threading.Lock()
        self.report = Report()

    def _busy(self, stage: str, seconds: float) -> None:
        with self.lock:
            self.report.busy[stage] = self.report.busy.get(stage, 0.0) + seconds

    # ------------------------------------------------------------------ failure handling
    def fail(self, exc: BaseException, who: str) -> None:
        with self.lock:
            if self.error is None and not isinstance(exc, PipelineAborted):
                self.error = exc
                log.error(f"{who}: {type(exc).__name__}: {exc}; stopping the analysis")
        self.aborted.set()

    def _put(self, q: "queue.Queue", item) -> None:
        while True:
            if self.aborted.is_set():
                raise PipelineAborted()
            try:
                q.put(item, timeout=0.2)
                return
            except queue.Full:
                continue

    def _get(self, q: "queue.Queue"):
        while True:
            if self.aborted.is_set():
                raise PipelineAborted()
            try:
                return q.get(timeout=0.2)
            except queue.Empty:
                continue

    # ------------------------------------------------------------------ planner + readers
    def _skip(self, who: str, job: FileJob, why: str, level=logging.DEBUG, message: Optional[str] = None) -> None:
        log.log(level, f"{who}: {why}")
        with self.lock:
            self.report.files_skipped += 1
            if message:
                self.report.messages.append(message)

    def _plan_file(self, job: FileJob) -> None:
        if job.rf.complete and not self.ignore_partial:     # (gather mode: the plan is rank 0's, this rank just delivers)
            return self._skip("planner", job, f"Skipping {job.shortpath}; already analyzed")
        if os.path.getsize(job.path) < FILE_SIZE_MINIMUM:
            return self._skip("planner", job, f"Skipping {job.shortpath}; below minimum analyzeable size")
        try:
            track = WavTrack(job.path)
        except (WavFormatError, OSError) as exc:           # one unreadable recording does not stop the others
            return self._skip("planner", job, f"{exc}; skipping", logging.WARNING,
                              f"unreadable, skipped: {job.shortpath} ({exc})")
        if track.samplerate != 16000 and not self._rate_supported(track.samplerate):
            track.close()                                   # (the reference resamples any rate; bd_resample refuses these)
            return self._skip("planner", job, f"{job.shortpath}: cannot resample {track.samplerate} Hz to 16000 Hz on the device; skipping",
                              logging.WARNING, f"sample rate {track.samplerate} Hz not supported, skipped: {job.shortpath}")
        job.fresh = not os.path.exists(job.rf.path_partial)
        if self.ignore_partial:
            chunks = framing.gaps_to_chunklist([(0, track.duration)], self.chunklength)
        else:
            chunks = job.rf.pending_chunks(track.duration, self.chunklength, self.framelength_s)
        if not chunks:
            track.close()
            return self._skip("planner", job, f"Skipping {job.shortpath}; nothing left to analyze")
        log.info(f"planner: buffering {job.shortpath}")
        job.track = track
        job.outstanding = len(chunks)                      # known before the first unit is out: no finalisation race
        for chunk in chunks:
            self._put(self.q_units, ReadUnit(job, (float(chunk[0]), float(chunk[1]))))

    def _rate_supported(self, rate: int) -> bool:
        """bd_resample_supported for the quality the engines run (host only: no handle, no device)."""
        if rate not in self._rates:
            from . import _lib
            self._rates[rate] = _lib.load().bd_resample_supported(int(rate), 16000, int(self.resample_quality)) == 1
        return self._rates[rate]

    def _planner(self) -> None:
        try:
            while True:
                job = self._get(self.q_files)
                if job == EXIT:
                    return
                self._plan_file(job)
        except PipelineAborted:
            pass
        except BaseException as exc:                       # noqa: BLE001 - every failure must poison the pipeline
            self.fail(exc, "planner")

    def _drop(self, job: FileJob) -> None:
        """A planned chunk that yields no rows (the file ends before it): count it as done."""
        with self.lock:
            job.outstanding -= 1
            finish = job.outstanding == 0
        if finish:
            self._put(self.q_write, job)

    def _bad_read(self, job: FileJob, track: WavTrack, frames_reached: int) -> None:
        """The reference's handle_bad_read (src/stream/worker.py:41-59): the file holds fewer frames than its header
        declares - a recorder whose battery died.  Said ONCE per file: WARNING when more than BAD_READ_ALLOWANCE of the
        declared length is missing, DEBUG when the loss is at the very end; the file stops there (chunks behind the end
        read nothing and are dropped)."""
        with self.lock:
            if job.bad_read:
                return
            job.bad_read = True
        final_second = frames_reached / track.samplerate
        msg = f"Unreadable audio at {round(final_second, 1)}s out of {round(track.duration, 1)}s for {job.shortpath}."
        if 1 - (final_second / track.duration) > BAD_READ_ALLOWANCE:
            log.warning(f"streamer: {msg}\nAborting early due to corrupt audio data.")
            with self.lock:
                self.report.messages.append(f"unreadable audio, stopped at {round(final_second, 1)}s: {job.shortpath}")
        else:
            log.debug(f"streamer: {msg}\nBad audio is near file end, results should be mostly unaffected.")

    def _read_unit(self, unit: ReadUnit) -> None:
        job, chunk, track = unit.job, unit.chunk, unit.job.track
        a, b = framing.chunk_sample_range(chunk, track.samplerate)
        want = min(b, track.frames_declared) - a           # what the header promises for this chunk
        if want <= 0:
            return self._drop(job)
        have = min(want, max(track.frames - a, 0))          # what the file holds
        if have <= 0:                                       # the whole chunk lies behind the end of a file cut short
            self._bad_read(job, track, track.frames)
            return self._drop(job)
        bpf = track.bytes_per_frame
        out_bpf = bpf if track.is_s16 else track.channels * 4          # any other sample format: float32 on the host
        slot, dev = self.pool.acquire(have * out_bpf, self.aborted)
        try:
            t0 = time.perf_counter()
            if self.device is None:                        # host-only stage (tests): straight into the pool's host buffer
                host = dev.numpy()
                got = track.read_raw_into(a, have, host)
                at = got * bpf
                if got and not track.is_s16:
                    f32 = track.convert(host[:at].copy())
                    at = f32.size * 4
                    host[:at] = f32.reshape(-1).view(np.uint8)
                st = None
            else:
                st = getattr(self._stage, "st", None)
                if st is None:
                    st = self._stage.st = ReaderStage.take(self.torch, self.device)
                    with self.lock:
                        self._stages.append(st)
                    self._busy("pin", st.pin_seconds)
                    t0 = time.perf_counter()
                if track.is_s16:                           # the file's bytes as they lie: one native call, no interpreter lock
                    fd, off, nb = track.file_range(a, have)
                    at = st.read(fd, off, nb, dev)
                    got = at // bpf
                else:                                      # converted to float32 on the host, piece by piece
                    piece = max(1, STAGE_BYTES // max(bpf, out_bpf))
                    got = at = 0
                    while got < have:
                        n = min(piece, have - got)
                        box = {}

                        def fill(buf, n=n, first=a + got):
                            r = box["frames"] = track.read_raw_into(first, n, buf)
                            if r == 0:
                                return 0
                            f32 = track.convert(buf[: r * bpf].copy())
                            buf[: f32.size * 4] = f32.reshape(-1).view(np.uint8)
                            return f32.size * 4
                        at += st.send(fill, dev.data_ptr() + at)
                        got += box["frames"]
                        if box["frames"] < n:
                            break
            self._busy("read", time.perf_counter() - t0)
            if got < want:                                 # short read (src/stream/worker.py:119-127): say it, truncate the
                self._bad_read(job, track, a + got)        # chunk, and the file ends here
                chunk = (chunk[0], round(chunk[0] + got / track.samplerate, 1))
            if got == 0:
                self.pool.release(slot)
                return self._drop(job)
            ready = None
            if st is not None:
                ready = self.events.take()
                ready.record(st.stream)
            with self.lock:
                self.report.chunks += 1
                self.report.audio_seconds += float(chunk[1] - chunk[0])
            self._put(self.q_analyze, ChunkTask(job, chunk, slot, at, got, track.channels, track.samplerate, track.is_s16, ready))
        except BaseException:
            self.pool.release(slot)
            raise

    def _reader(self, rid: int) -> None:
        try:
            while True:
                unit = self._get(self.q_units)
                if unit == EXIT:
                    return
                self._read_unit(unit)
        except PipelineAborted:
            pass
        except BaseException as exc:                       # noqa: BLE001
            self.fail(exc, f"streamer {rid}")

    # ------------------------------------------------------------------ analyzers
    def _analyzer(self, aid: int) -> None:
        torch = self.torch
        held: List[int] = []
        try:
            log.info(f"analyzer {aid}: launching")
            from .engine import LaunchVerdict
            engine = self.make_engine()                    # per-thread engine, initialised in-thread
            device = engine.device
            stream = torch.cuda.Stream(device)
            n_classes = engine.n_classes
            # everything a batch needs is allocated here, once: pinned result blocks sized for the largest batch the
            # batching rule below can form, and (grow-only) device buffers for the 16 kHz PCM; the chunks themselves are
            # already on the device (ChunkPool), brought there by the readers
            chunk_windows = int(self.chunklength / self.framehop_s) + 2
            results_pool = ResultPool(torch, RESULT_BLOCKS, BATCH_WINDOWS + chunk_windows + 8, n_classes)
            pcm_arena = DeviceArena(torch, device, torch.float32)
            logit_dev = [torch.empty((results_pool.rows, n_classes), dtype=torch.float32, device=device) for _ in range(2)]
            log.info(f"analyzer {aid}: processing on GPU")
            t_wait = time.perf_counter()
            finished = False
            pending = None                                  # the batch before the current one: (item, range word, its PCM)
            n_batch = 0

            def settle(p) -> None:
                """Forward a batch to the writer once it is known to be sound: the f16 matrix path cannot represent an
                activation beyond its calibrated headroom (the batch's own range word, LaunchVerdict); such a batch is
                computed again with exact f32 products.  Runs while the NEXT batch occupies the GPU, so the wait costs
                nothing."""
                item, verdict, pcms = p
                t0 = time.perf_counter()
                raised = verdict.wait()
                self._busy("settle", time.perf_counter() - t0)
                if raised:
                    log.warning(f"analyzer {aid}: an activation left the f16 range; recomputing {len(item.tasks)} chunk(s) in exact f32")
                    with torch.cuda.stream(stream):
                        _, whole, _ = engine.launch(pcms, self.hop, self.step, False, True, mode="f32")
                        torch.from_numpy(item.host).copy_(whole, non_blocking=True)
                        # (chunks that need no resampling are views of their ChunkPool slots, which the WRITER gives back
                        #  behind this event: the repeat reads the audio of its own batch whatever has been uploaded since)
                        item.done.record(stream)          # (the writer has not seen this item yet: the event is ours)
                    engine.overflow_reruns += 1
                self._put(self.q_write, item)

            while not finished:
                try:
                    task = self.q_analyze.get_nowait()
                except queue.Empty:                        # nothing to overlap with: do not sit on a finished batch
                    if pending is not None:
                        settle(pending)
                        pending = None
                    task = self._get(self.q_analyze)
                if task == EXIT:
                    break
                waited = time.perf_counter() - t_wait
                if waited > BOTTLENECK_SECONDS:
                    log.debug(f"analyzer {aid}: BUFFER BOTTLENECK: analyzer {aid} received assignment after {round(waited, 1)}s")
                t_start = time.perf_counter()
                batch = [task]
                windows = engine.num_windows(self._out_samples(task), self.hop, self.step)
                while windows < BATCH_WINDOWS and len(batch) < BATCH_CHUNKS:
                    try:
                        nxt = self.q_analyze.get_nowait()
                    except queue.Empty:
                        break
                    if nxt == EXIT:
                        finished = True
                        break
                    batch.append(nxt)
                    windows += engine.num_windows(self._out_samples(nxt), self.hop, self.step)
                held = [t.slot for t in batch]
                which = n_batch & 1
                n_batch += 1
                outs = [self._out_samples(t) if (t.s16 or t.rate != 16000 or t.channels > 1) else 0 for t in batch]
                with torch.cuda.stream(stream):
                    pcm = pcm_arena.get(which, sum((n + 63) & ~63 for n in outs), stream)
                    pcms, pat = [], 0
                    for t, n_out in zip(batch, outs):
                        if t.ready is not None:
                            stream.wait_event(t.ready)       # the reader's last copy of this chunk (the wait captures the
                            self.events.give(t.ready)        # event's state now: it may be recorded again at once)
                            t.ready = None
                        raw = self.pool.buffer(t.slot)
                        view = raw[: t.nbytes].view(torch.int16 if t.s16 else torch.float32).view(t.frames, t.channels)
                        if n_out:
                            pcms.append(engine.resample(view, t.rate, 16000, out=pcm[pat:pat + n_out]))   # also s16 -> f32, channel mean
                            pat += (n_out + 63) & ~63
                        else:
                            pcms.append(view[:, 0])
                    verdict = LaunchVerdict(stream)
                    total = sum(engine.num_windows(int(p.numel()), self.hop, self.step) for p in pcms)
                    block = results_pool.take(total, self.aborted)
                    dev_rows = logit_dev[which] if total <= logit_dev[which].shape[0] else None
                    _, whole, counts = engine.launch(pcms, self.hop, self.step, False, True, verdict=verdict,
                                                     out=dev_rows[:total] if dev_rows is not None and total else None)
                    host = block[:total]
                    if total:
                        host.copy_(whole, non_blocking=True)
                    done = self.events.take()
                    done.record(stream)
                self._busy("analyze", time.perf_counter() - t_start)
                # the chunks' device slots go back to the pool when the batch's rows have been read (the writer waits for its
                # event anyway); the batch itself goes to the writer one batch later, after its range word has been looked at
                item = WriteItem(batch, host.numpy(), counts, done, aid, t_start, (results_pool, block))
                held = []
                if pending is not None:
                    settle(pending)
                pending = (item, verdict, pcms)
                t_wait = time.perf_counter()
            if pending is not None:
                settle(pending)
            log.debug(f"analyzer {aid}: terminating")
        except PipelineAborted:
            pass
        except BaseException as exc:                       # noqa: BLE001
            self.fail(exc, f"analyzer {aid}")
        finally:
            for s in held:
                self.pool.release(s)

    def _out_samples(self, t: ChunkTask) -> int:
        """16 kHz samples the chunk becomes (resample_poly's length)."""
        if t.rate == 16000:
            return t.frames
        from math import gcd
        g = gcd(16000, t.rate)
        up, down = 16000 // g, t.rate // g
        return (t.frames * up + down - 1) // down

    # ------------------------------------------------------------------ writer
    def _finalize(self, job: FileJob) -> None:
        if job.track is not None:
            job.track.close()
            job.track = None
        if self.file_sink is not None:
            rows, job.rows = sorted(job.rows, key=lambda sr: sr[0]), []
            self.file_sink(job, rows)
            with self.lock:
                self.report.files_done += 1
            return
        if not os.path.exists(job.rf.path_partial):
            return
        if job.fresh and job.bodies:
            # the reference reads the partial file, sorts by start and writes the complete one (src/write/worker.py:82-86);
            # for a recording this run started from nothing, the same bytes come from the rows kept in memory
            job.bodies.sort(key=lambda sb: sb[0])
            tmp = job.rf.path_complete + ".tmp"
            with open(tmp, "wb") as f:
                f.write(job.header)
                f.writelines(b for _, b in job.bodies)
            os.replace(tmp, job.rf.path_complete)
            os.remove(job.rf.path_partial)
        else:
            job.rf.finalize()
        job.bodies = []
        with self.lock:
            self.report.files_done += 1

    def _writer(self) -> None:
        try:
            log.info("writer: launching")
            while True:
                item = self._get(self.q_write)
                if item == EXIT:
                    break
                if isinstance(item, FileJob):              # a recording whose last planned chunk turned out to be empty
                    self._finalize(item)
                    continue
                t0 = time.perf_counter()
                item.done.synchronize()
                self._busy("write_wait", time.perf_counter() - t0)
                self.events.give(item.done)
                for t in item.tasks:                       # every kernel that read the chunks has finished
                    self.pool.release(t.slot)
                seconds = time.perf_counter() - item.t_start
                audio = sum(t.chunk[1] - t.chunk[0] for t in item.tasks)
                rate = audio / seconds if seconds > 0 else float("inf")
                at = 0
                t_fmt0, t_io = time.perf_counter(), 0.0
                for t, n in zip(item.tasks, item.counts):
                    rows = item.host[at:at + n]
                    at += n
                    if self.file_sink is not None:
                        t.job.rows.append((t.chunk[0], rows.copy()))
                        head = body = None
                    elif self.threshold is None:
                        head, body = results.activation_csv(rows, self.classes, self.framehop_s, self.digits_time, t.chunk[0],
                                                            self.classes_out, self.digits_results)
                    else:
                        head, body = results.detection_csv(rows, self.threshold, self.classes, self.framehop_s,
                                                           self.digits_time, t.chunk[0])
                    if head is not None:
                        t1 = time.perf_counter()
                        t.job.rf.append_text(head, body)
                        t_io += time.perf_counter() - t1
                    if head is not None and t.job.fresh:
                        t.job.header = head
                        t.job.bodies.append((t.chunk[0], body))
                    log.log(PROGRESS, f"analyzer {item.analyzer}: analyzed {t.job.shortpath}, chunk "
                                      f"({t.chunk[0]:.{self.digits_time}f}, {t.chunk[1]:.{self.digits_time}f}) "
                                      f"in {seconds:.2f}s (rate: {rate:.1f})")
                    finish = False
                    with self.lock:
                        self.report.windows += n
                        t.job.outstanding -= 1
                        finish = t.job.outstanding == 0
                    if finish:
                        t1 = time.perf_counter()
                        self._finalize(t.job)
                        t_io += time.perf_counter() - t1
                self._busy("format", time.perf_counter() - t_fmt0 - t_io)
                self._busy("write", t_io)
                if item.block is not None:                 # every row of the block has been turned into text (or copied)
                    item.block[0].give(item.block[1])
            log.debug("writer: terminating")
        except PipelineAborted:
            pass
        except BaseException as exc:                       # noqa: BLE001
            self.fail(exc, "writer")

    # ------------------------------------------------------------------ driver
    def run(self, jobs: Sequence[FileJob]) -> Report:
        self.report.files_total += len(jobs)
        for j in jobs:
            self.q_files.put(j)
        self.q_files.put(EXIT)
        planner = threading.Thread(target=self._planner, name="planner", daemon=True)
        readers = [threading.Thread(target=self._reader, args=(i,), name=f"streamer-{i}", daemon=True) for i in range(self.n_readers)]
        analyzers = [threading.Thread(target=self._analyzer, args=(i,), name=f"analyzer-{i}", daemon=True) for i in range(self.n_analyzers)]
        writer = threading.Thread(target=self._writer, name="writer", daemon=True)
        finished = threading.Event()

        def stop_requested() -> bool:
            if not self.stop_event.is_set():
                return False
            with self.lock:
                if self.error is None and not self.aborted.is_set():
                    self.report.end_reason = "interrupted"
                    log.warning("coordinator: stop requested; ending the analysis (partial results stay resumable)")
            self.aborted.set()
            return True

        def watch_stop() -> None:
            while not stop_requested() and not finished.wait(0.02):
                pass

        watcher = None
        if self.stop_event is not None:
            stop_requested()                  # already set: no worker gets to do anything
            watcher = threading.Thread(target=watch_stop, name="stop-watch", daemon=True)
            watcher.start()
        for t in [planner] + readers + analyzers + [writer]:
            t.start()
        planner.join()
        for _ in readers:
            self._force_put(self.q_units, EXIT)
        for t in readers:
            t.join()
        log.debug("coordinator: streamers done")
        for _ in analyzers:
            self._force_put(self.q_analyze, EXIT)
        for t in analyzers:
            t.join()
        log.debug("coordinator: analyzers done")
        self._force_put(self.q_write, EXIT)
        writer.join()
        log.debug("coordinator: writer done")
        finished.set()
        if watcher is not None:
            watcher.join()
        for st in self._stages:                           # (an aborted run may have left copies in flight: they end before the
            st.stream.synchronize()                        #  stage is used again)
            st.give_back()
        self._stages = []
        if self.error is not None:
            raise self.error
        return self.report

    def _force_put(self, q: "queue.Queue", item) -> None:
        """Sentinels must get through even when the pipeline is poisoned and the queue is full of abandoned work."""
        while True:
            try:
                q.put(item, timeout=0.2)
                return
            except queue.Full:
                if self.aborted.is_set():
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        pass
