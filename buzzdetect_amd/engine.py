"""Host side of the MI355X analyze hot path: owns the C-ABI handle, the device workspace and
the hop arithmetic; PyTorch is used only for device memory and streams.

Replaces, behind the reference's plugin surface (see ``buzzdetect_amd/dropin``):
``YamnetK2.embed`` (embedders/yamnet_k2/embedder.py:27-37), ``EmbedderYamnet.embed``
(embedders/yamnet/embedder.py:33-44) and ``ModelGeneralV3.predict``
(models/model_general_v3/model.py:18-31).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, weights

SAMPLE_RATE = 16000
STFT_HOP = 160


def hop_samples(framehop_s: float) -> int:
    """``tf.cast(patch_hop_seconds * sample_rate, tf.int32)`` (features.py:99): the Python-float product becomes
    a float32 tensor first, then truncates (0.96 * 0.35 * 16000 = 5375.999... -> 5376.0f -> 5376, not 5375)."""
    return int(np.float32(framehop_s * 16000.0))


def patch_step(framehop_s: float) -> int:
    """``int(round(spectrogram_sample_rate * patch_hop_seconds))`` (features.py:66-71)."""
    spectrogram_sample_rate = 16000.0 / STFT_HOP
    return int(round(spectrogram_sample_rate * framehop_s))


class _VerdictPool:
    """(pinned int32 word, HIP event) pairs for the range verdicts, recycled.

    The kernels of a launch set write their word straight across PCIe (only when a chunk does leave the range), which
    PyTorch knows nothing about, so the caching host allocator must not get the memory back while those kernels may still
    run: a pair is reused only once the event recorded behind its launch set has completed.  Events are recycled with
    their words - a fresh event per launch makes the HIP runtime grow its signal pool while the host runs ahead of the
    GPU, a one-off stall of tens of milliseconds in the middle of a run."""

    BLOCK = 256

    def __init__(self):
        from collections import deque
        self._lock = threading.Lock()
        self._idle: "deque" = deque()         # (word, event) given back, oldest first; the event may still be pending
        self._blocks: list = []
        self._next = 0

    def take(self):
        # ``give_back`` runs from ``LaunchVerdict.__del__``, i.e. whenever the garbage collector pleases - also in the
        # middle of this function, on this thread, when an allocation below triggers a collection that finalises a verdict
        # caught in a reference cycle.  So ``give_back`` takes no lock (``deque.append`` is atomic), and this lock only
        # keeps two takers from claiming the same idle pair.
        word = event = None
        with self._lock:
            if self._idle and self._idle[0][1].query():
                word, event = self._idle.popleft()
            elif self._blocks and self._next < self.BLOCK:
                word = self._blocks[-1][self._next:self._next + 1]
                self._next += 1
        if word is None:                       # a new block of pinned words: allocated outside the lock
            block = torch.zeros(self.BLOCK, dtype=torch.int32, pin_memory=True)
            with self._lock:
                self._blocks.append(block)
                self._next = 1
            word = block[0:1]
        if event is None:
            event = torch.cuda.Event()
        word[0] = 0
        return word, event

    def give_back(self, word: torch.Tensor, event: "torch.cuda.Event") -> None:
        self._idle.append((word, event))       # lock-free on purpose (see take)


class _EventPool:
    """Recycled HIP events for results that carry no range verdict (exact-f32 mode, embeddings of the f32 path ...): the
    same reason as above - a fresh event per result makes the runtime grow its signal pool mid-run."""

    def __init__(self):
        from collections import deque
        self._idle: "deque" = deque()

    def take(self) -> "torch.cuda.Event":
        try:
            event = self._idle.popleft()       # atomic; an event is only given back by the one result that owned it
        except IndexError:
            return torch.cuda.Event()
        if event.query():
            return event
        self._idle.append(event)               # still pending behind somebody's stream: leave it, make a new one
        return torch.cuda.Event()

    def give_back(self, event: "torch.cuda.Event") -> None:
        self._idle.append(event)


_event_pools: dict = {}


def _events_of(device_index: int) -> _EventPool:
    with _verdict_pools_lock:
        pool = _event_pools.get(device_index)
        if pool is None:
            pool = _event_pools[device_index] = _EventPool()
        return pool


_verdict_pools: dict = {}                  # device index -> pool (an event belongs to the device it was first recorded on)
_verdict_pools_lock = threading.Lock()


def _verdicts_of(device_index: int) -> _VerdictPool:
    with _verdict_pools_lock:
        pool = _verdict_pools.get(device_index)
        if pool is None:
            pool = _verdict_pools[device_index] = _VerdictPool()
        return pool


class LaunchVerdict:
    """The range word of ONE launch set (``bd_predict_chunks``): a pinned int32 that only this set's kernels can raise,
    and the event that says they have all finished.  Every DeviceResult of the set shares it, nobody
    resets anything: whichever thread reads a result later - the reference's writer does, src/write/worker.py:69, while the
    analyzer has long enqueued the next chunk, src/inference/worker.py:71-74 - learns about exactly its own launches."""

    def __init__(self, stream: torch.cuda.Stream):
        self._pool = _verdicts_of(stream.device_index)
        self.word, self.event = self._pool.take()
        self.stream = stream

    def wait(self) -> bool:
        """Blocks until the launch set has finished; True when one of its activations left the f16 range."""
        self.event.synchronize()
        return int(self.word[0]) != 0

    def __del__(self):
        try:
            self._pool.give_back(self.word, self.event)
        except Exception:                      # interpreter shutdown
            pass


class DeviceResult:
    """What ``predict``/``embed`` hand back: a device tensor that also answers ``.numpy()``,
    the one method the reference's writer calls on results (src/write/worker.py:69).

    ``verdict``: the range word of the launch set that computed the rows.  ``redo``: how to recompute them with exact-f32
    products (on the stream they were computed on, with a per-call mode: the engine's own state is not touched, so it is
    safe from a thread other than the one that keeps predicting); ``.numpy()`` uses it when the verdict says so."""

    def __init__(self, tensor: torch.Tensor, stream: torch.cuda.Stream, engine: "Optional[HipEngine]" = None,
                 redo=None, verdict: Optional[LaunchVerdict] = None):
        self.tensor = tensor
        self._stream = stream
        self._host: Optional[np.ndarray] = None
        self._engine = engine
        self._redo = redo
        self._verdict = verdict
        self._lock = threading.Lock()
        self._done = None
        if verdict is None:                   # results that never pass through the f16 path (mode f32, resample ...)
            self._events = _events_of(stream.device_index)
            self._done = self._events.take()
            self._done.record(stream)

    def __del__(self):
        try:
            if self._done is not None:
                self._events.give_back(self._done)
        except Exception:                      # interpreter shutdown
            pass

    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(self.tensor.shape)

    def numpy(self) -> np.ndarray:
        with self._lock:
            if self._host is None:
                if self._verdict is not None and self._verdict.wait() and self._redo is not None:
                    # an activation beyond the f16 range went through the matrix path: these rows may be garbage (and a
                    # ReLU can hide it).  Compute them again with exact f32 products.
                    self.tensor = self._redo()
                    if self._engine is not None:
                        self._engine.overflow_reruns += 1
                elif self._verdict is None:
                    self._done.synchronize()
                self._redo = None
                self._host = self.tensor.cpu().numpy()
            return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self) -> int:
        return self.tensor.shape[0]


class HipEngine:
    """One engine per GPU / per analyzer thread (src/inference/worker.py:21,78)."""

    def __init__(self, embeddername: str = "yamnet_k2", modelname: Optional[str] = "model_general_v3",
                 device: Optional[int] = None, embedder_variables: Optional[str] = None,
                 embedder_blob: Optional[np.ndarray] = None, variables_candidates=None,
                 synthetic_weights: Optional[bool] = None):
        """``embedder_blob`` / ``embedder_variables``: the weights / an explicit ``variables.data-00000-of-00001``;
        else ``variables_candidates`` (the embedder plugin passes the places beside itself where the reference keeps its
        SavedModel; default: ``embedders/<name>`` under the working directory and under the packaged overlay), then
        ``$BUZZDETECT_YAMNET_VARIABLES``; ``synthetic_weights`` (or ``BUZZDETECT_SYNTHETIC_WEIGHTS=1``) opts in to seeded
        stand-ins with a warning.  No source: ``FileNotFoundError`` (``weights.load_embedder_blob``)."""
        self._handle = C.c_void_p()
        self._lib = _lib.load()
        # the weights first (host only): a missing model fails the same way with or without a GPU in the box
        if embedder_blob is None:
            if variables_candidates is None:
                variables_candidates = weights.default_candidates(embeddername)
            blob = weights.load_embedder_blob(embedder_variables, variables_candidates, synthetic_weights)
        else:
            blob = embedder_blob
        if not torch.cuda.is_available():
            raise RuntimeError("buzzdetect_amd: no HIP device visible to PyTorch; this engine has no CPU path")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)

        blob = np.ascontiguousarray(blob, dtype=np.float32)
        mel = np.ascontiguousarray(weights.load_mel(embeddername), dtype=np.float32)
        w = _lib.bd_weights()
        w.embedder_blob = blob.ctypes.data_as(C.POINTER(C.c_float))
        w.embedder_floats = blob.size
        w.mel = mel.ctypes.data_as(C.POINTER(C.c_float))
        self.classes = None
        self.n_classes = 0
        if modelname is not None:
            head = weights.load_head(modelname)
            hk = np.ascontiguousarray(head.kernel, dtype=np.float32)
            hb = np.ascontiguousarray(head.bias, dtype=np.float32)
            w.head_kernel = hk.ctypes.data_as(C.POINTER(C.c_float))
            w.head_bias = hb.ctypes.data_as(C.POINTER(C.c_float))
            w.n_classes = hb.size
            self.classes = head.classes
            self.n_classes = int(hb.size)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.bd_create(C.byref(self._handle), self.device_index, C.byref(w)))
        # the handle is not thread-safe (include/buzzdetect_hip.h): every call that takes it goes through this lock, so a
        # writer thread that repeats a flagged chunk cannot interleave with the analyzer thread's next predict
        self._lock = threading.RLock()
        self._workspace: Optional[torch.Tensor] = None
        self._mode = "f16x3"
        self.overflow_reruns = 0          # results recomputed in exact f32 because an activation left the f16 range
        # host staging for NumPy inputs: a ring of pinned buffers, each guarded by the event of the async
        # H2D copy that last read it (a single shared buffer would be overwritten while a copy is in flight)
        self._pinned: list = [None] * 4
        self._pinned_events: list = [None] * 4
        self._pinned_next = 0

    # ------------------------------------------------------------------ lifecycle
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.bd_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def set_group_windows(self, windows: int) -> None:
        with self._lock:
            _lib.check(self._lib.bd_set_group_windows(self._handle, int(windows)))

    POINTWISE_MODES = ("f32", "f16x3", "f16")

    MODE_CODES = {"f32": 0, "f16x3": 1, "f16": 2}

    def set_pointwise_mode(self, mode: str) -> None:
        """'f32' = exact f32 MFMA products; 'f16x3' = split-f16 MFMA (default, same accuracy class); 'f16' = plain f16
        operands, one MFMA per product (config 5; ~1e-3 on the logits)."""
        with self._lock:
            _lib.check(self._lib.bd_set_pointwise_mode(self._handle, self.MODE_CODES[mode]))
            self._mode = mode

    # ------------------------------------------------------------------ operand scales of the f16 modes
    def scales(self) -> Tuple[np.ndarray, np.ndarray]:
        """(act_exp[13], act_max[13]) for layers 2..14: the power-of-two exponent each layer's GEMM input is scaled by and
        the largest activation the calibration passes saw there (``bd_get_scales``)."""
        ex = (C.c_int32 * 13)()
        mx = (C.c_float * 13)()
        with self._lock:
            _lib.check(self._lib.bd_get_scales(self._handle, ex, mx))
        return np.array(ex[:], dtype=np.int32), np.array(mx[:], dtype=np.float32)

    def set_activation_exponents(self, exps: Sequence[int]) -> None:
        """Test hook (``bd_set_activation_exponents``): override the calibrated exponents; synchronises the device."""
        arr = (C.c_int32 * 13)(*[int(v) for v in exps])
        with self._lock, torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(self._lib.bd_set_activation_exponents(self._handle, arr))

    def calibrate(self, samples, framehop_s: float = 0.96) -> None:
        """Widen the activation scales with a pass over ``samples`` in exact-f32 arithmetic (``bd_calibrate``).  Call it
        while nothing else is in flight on this engine."""
        hop, step = hop_samples(framehop_s), patch_step(framehop_s)
        x = self.to_device(samples)
        with self._lock, torch.cuda.device(self.device):
            ws_bytes = _lib.check(self._lib.bd_workspace_bytes(self._handle, x.numel(), hop, step))
            ws = self._ws(ws_bytes)
            torch.cuda.synchronize(self.device)
            _lib.check(self._lib.bd_calibrate(self._handle, x.data_ptr(), x.numel(), hop, step, ws.data_ptr(), ws.numel(),
                                              self._stream().cuda_stream))

    def range_exceeded(self, stream: Optional[torch.cuda.Stream] = None, reset: bool = True) -> bool:
        """Did any launch since the last reset convert an activation beyond the f16 range (modes 'f16x3' / 'f16')?
        Waits for ``stream``."""
        flag = C.c_int32(0)
        s = stream or self._stream()
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.bd_range_flag(self._handle, C.byref(flag), 1 if reset else 0, s.cuda_stream))
        return bool(flag.value)

    def range_flag_to(self, dst: torch.Tensor, reset: bool = False) -> None:
        """Enqueue (no wait) a copy of the range word into ``dst`` (int32, device or pinned host) on the current stream."""
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.bd_range_flag_copy(self._handle, dst.data_ptr(), 1 if reset else 0,
                                                    self._stream().cuda_stream))

    def set_fusion(self, stem=True, separable=True) -> None:
        """``bd_set_fusion``: stem True / 3 = layers 1-3 as one kernel (the default: the layer-2 tile handed over in registers),
        5 = the same on the kernel of rounds 2-4 (its tile through LDS), False = one kernel per op; separable True / 1 = the
        default launch set behind the stem (layer 4 + depthwise 5, pointwise 5 - layer 7 on chip, layers 8-12 + depthwise 13
        on chip, pointwise 13 + depthwise 14, pointwise 14 + pool), 10 = layers 5-7 on the four kernels of round 4, False = one
        kernel per op.  Every other code is refused (removed in round 6)."""
        stem_code = 3 if stem is True else int(stem)
        with self._lock:
            _lib.check(self._lib.bd_set_fusion(self._handle, stem_code, int(separable)))

    def set_pointwise_variant(self, layer: int, variant: int) -> None:
        with self._lock:
            _lib.check(self._lib.bd_set_pointwise_variant(self._handle, int(layer), int(variant)))

    def num_windows(self, n_samples: int, hop: int, step: int) -> int:
        return _lib.check(self._lib.bd_num_windows(int(n_samples), int(hop), int(step)))

    def num_frames(self, n_samples: int, hop: int) -> int:
        return _lib.check(self._lib.bd_num_frames(int(n_samples), int(hop)))

    def _stream(self) -> torch.cuda.Stream:
        return torch.cuda.current_stream(self.device)

    def _ws(self, nbytes: int) -> torch.Tensor:
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = None
            self._workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._workspace

    def to_device(self, samples) -> torch.Tensor:
        """1-D float32 chunk (numpy / torch CPU / torch device) -> contiguous device tensor."""
        if isinstance(samples, DeviceResult):
            samples = samples.tensor
        if isinstance(samples, torch.Tensor):
            t = samples
            if t.dim() != 1:
                raise ValueError("audio samples must be one-dimensional")
            if t.dtype != torch.float32:
                t = t.to(torch.float32)
            if t.device != self.device:
                t = t.to(self.device, non_blocking=True)
            t = t.contiguous()
        else:
            a = np.asarray(samples)
            if a.ndim != 1:
                raise ValueError("audio samples must be one-dimensional")
            a = np.ascontiguousarray(a, dtype=np.float32)
            n = a.size
            slot = self._pinned_next
            self._pinned_next = (slot + 1) % len(self._pinned)
            if self._pinned_events[slot] is not None:
                self._pinned_events[slot].synchronize()          # the copy that last used this buffer is done
            if self._pinned[slot] is None or self._pinned[slot].numel() < n:
                self._pinned[slot] = torch.empty(max(n, 1), dtype=torch.float32).pin_memory()
            staging = self._pinned[slot]
            staging[:n].copy_(torch.from_numpy(a))
            t = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)[:n]
            t.copy_(staging[:n], non_blocking=True)
            if self._pinned_events[slot] is None:          # one event per slot, recorded again for every copy
                self._pinned_events[slot] = torch.cuda.Event()
            self._pinned_events[slot].record(self._stream())
        if t.data_ptr() % 16:
            t = t.clone()
        return t

    # ------------------------------------------------------------------ streamer stage
    RESAMPLE_QUALITIES = {"scipy": 0, "hq": 1}

    def set_resample_quality(self, quality: str) -> None:
        """"hq" (default): the soxr_hq filter class librosa.resample runs in the reference (src/stream/worker.py:128);
        "scipy": rounds 1-3's scipy.signal.resample_poly default (61 taps for 48 -> 16 kHz)."""
        if quality not in self.RESAMPLE_QUALITIES:
            raise ValueError(f"quality must be one of {sorted(self.RESAMPLE_QUALITIES)}")
        with self._lock:
            _lib.check(self._lib.bd_set_resample_quality(self._handle, self.RESAMPLE_QUALITIES[quality]))

    def resample(self, samples, rate_in: int, rate_out: int = SAMPLE_RATE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n] or [n, channels] float32 — or int16 PCM, scaled by 1/32768 — at ``rate_in`` -> mono float32 [m] at
        ``rate_out`` on the device (np.mean(axis=1) + librosa.resample of src/stream/worker.py:116-128 as one kernel).
        ``out``: a caller-owned float32 device tensor of exactly the output's length (the feeder's staging arena)."""
        is_s16 = (samples.dtype == torch.int16) if isinstance(samples, torch.Tensor) else (np.asarray(samples).dtype == np.int16)
        dt_t, dt_n = (torch.int16, np.int16) if is_s16 else (torch.float32, np.float32)
        if isinstance(samples, torch.Tensor):
            t = samples.to(self.device, dtype=dt_t, non_blocking=True)
        else:
            a = np.ascontiguousarray(np.asarray(samples), dtype=dt_n)
            if not a.flags.writeable:                 # e.g. np.frombuffer views: torch wants a writable array
                a = a.copy()
            t = torch.from_numpy(a).to(self.device)
        if t.dim() == 1:
            t = t[:, None]
        if t.dim() != 2:
            raise ValueError("samples must be [n] or [n, channels]")
        t = t.contiguous()
        n_in, channels = t.shape
        n_out = _lib.check(self._lib.bd_resample_length(n_in, int(rate_in), int(rate_out)))
        if out is None:
            out = torch.empty(max(n_out, 1), dtype=torch.float32, device=self.device)[:n_out]
        elif (out.dim() != 1 or out.numel() != n_out or out.dtype != torch.float32 or out.device != self.device
              or not out.is_contiguous() or out.data_ptr() % 16):
            raise ValueError(f"out must be a contiguous, 16-byte aligned float32 [{n_out}] tensor on {self.device}")
        fn = self._lib.bd_resample_s16 if is_s16 else self._lib.bd_resample
        with self._lock, torch.cuda.device(self.device):      # (the first use of a rate ratio adds its filter to the handle)
            _lib.check(fn(self._handle, t.data_ptr(), n_in, channels, int(rate_in), int(rate_out),
                          out.data_ptr(), self._stream().cuda_stream))
        t.record_stream(self._stream())
        return out

    # ------------------------------------------------------------------ hot path
    def frontend(self, samples, hop: int) -> torch.Tensor:
        """[N] PCM -> [T,64] log-mel (features.py:22-58 + pad_waveform :82-108)."""
        x = self.to_device(samples)
        t = self.num_frames(x.numel(), hop)
        out = torch.empty((t, _lib.MEL_BANDS), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            _lib.check(self._lib.bd_frontend(self._handle, x.data_ptr(), x.numel(), hop, out.data_ptr(),
                                             self._stream().cuda_stream))
        return out

    def patches(self, logmel: torch.Tensor, step: int) -> torch.Tensor:
        """[T,64] -> [W,96,64] (features.py:65-79)."""
        t = logmel.shape[0]
        w = 1 + (t - _lib.PATCH_FRAMES) // step if t >= _lib.PATCH_FRAMES else 0
        out = torch.empty((w, _lib.PATCH_FRAMES, _lib.MEL_BANDS), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.bd_patches(self._handle, logmel.data_ptr(), t, step, out.data_ptr(),
                                            self._stream().cuda_stream))
        return out

    def launch(self, parts: List[torch.Tensor], hop: int, step: int, want_embeddings: bool, want_logits: bool,
               out: Optional[torch.Tensor] = None, mode: Optional[str] = None, verdict: Optional[LaunchVerdict] = None):
        """One launch set over ``parts`` (device tensors, one per chunk; nothing is concatenated) on the current stream
        (``bd_predict_chunks``).  ``mode``: arithmetic for this call only.  ``verdict``: takes this call's range word.
        Returns (embeddings or None, logits or None, windows per chunk)."""
        if not 1 <= len(parts) <= 64:
            raise ValueError("a launch set takes 1..64 chunks")
        if want_logits and self.n_classes == 0:
            raise RuntimeError("engine was created without a classifier head")
        for p in parts:
            if (not isinstance(p, torch.Tensor) or p.dim() != 1 or p.dtype != torch.float32 or p.device != self.device
                    or not p.is_contiguous() or p.data_ptr() % 4):
                raise ValueError("launch() takes one-dimensional contiguous float32 tensors on the engine's device "
                                 "(to_device() makes one out of anything predict() accepts)")
        nc = len(parts)
        lengths = (C.c_int64 * nc)(*[int(p.numel()) for p in parts])
        ptrs = (C.c_void_p * nc)(*[p.data_ptr() if p.numel() else None for p in parts])
        per = (C.c_int64 * nc)()
        total = _lib.check(self._lib.bd_batch_num_windows(lengths, nc, hop, step, per))
        stream = self._stream()
        with self._lock, torch.cuda.device(self.device):
            ws_bytes = _lib.check(self._lib.bd_batch_workspace_bytes(self._handle, lengths, nc, hop, step))
            ws = self._ws(ws_bytes)
            emb = torch.empty((total, _lib.EMBEDDING_SIZE), dtype=torch.float32, device=self.device) if want_embeddings else None
            logits = None
            if want_logits:
                if out is None:
                    logits = torch.empty((total, self.n_classes), dtype=torch.float32, device=self.device)
                else:       # caller-owned rows (e.g. a slice of a per-recording buffer that is gathered later)
                    if (tuple(out.shape) != (total, self.n_classes) or out.dtype != torch.float32 or
                            out.device != self.device or not out.is_contiguous() or out.data_ptr() % 16):
                        raise ValueError(f"out must be a contiguous, 16-byte aligned float32 [{total}, {self.n_classes}] "
                                         f"tensor on {self.device}")
                    logits = out
            _lib.check(self._lib.bd_predict_chunks(
                self._handle, ptrs, lengths, nc, hop, step, ws.data_ptr(), ws.numel(),
                emb.data_ptr() if emb is not None else None, logits.data_ptr() if logits is not None else None,
                -1 if mode is None else self.MODE_CODES[mode],
                verdict.word.data_ptr() if verdict is not None else None, stream.cuda_stream))
            if verdict is not None:
                verdict.event.record(stream)
        for p in parts:
            p.record_stream(stream)
        return emb, logits, [int(v) for v in per]

    def run(self, samples, hop: int, step: int, want_embeddings: bool, want_logits: bool,
            out: Optional[torch.Tensor] = None, mode: Optional[str] = None, verdict: Optional[LaunchVerdict] = None):
        emb, logits, _ = self.launch([self.to_device(samples)], hop, step, want_embeddings, want_logits, out, mode, verdict)
        return emb, logits

    def _redo_on(self, stream: torch.cuda.Stream, part: torch.Tensor, hop: int, step: int, embeddings: bool,
                 out: Optional[torch.Tensor] = None):
        """The exact-f32 repeat of one chunk: enqueued on the stream the chunk was computed on (behind whatever the
        owning thread has queued there since - same workspace, stream order keeps it safe), waited for, returned."""
        def run():
            with torch.cuda.stream(stream):
                e, l = self.run(part, hop, step, embeddings, not embeddings, out=out, mode="f32")
                done = torch.cuda.Event()
                done.record(stream)
            done.synchronize()
            return e if embeddings else l
        return run

    def predict_batch(self, chunks, framehop_s: float, want_embeddings: bool = False, mode: Optional[str] = None):
        """Several chunks through one launch set (``bd_predict_chunks``): each chunk keeps its own end-of-chunk
        zero padding, so the rows are exactly those of one ``predict`` per chunk.  Returns one DeviceResult per
        chunk (views of one device tensor); with ``want_embeddings`` a second list with the embeddings.  The results
        share the launch set's range word; a flagged set repeats chunk by chunk, each when it is read."""
        if not 1 <= len(chunks) <= 64:
            raise ValueError("predict_batch takes 1..64 chunks")
        hop, step = hop_samples(framehop_s), patch_step(framehop_s)
        parts = [self.to_device(c) for c in chunks]
        stream = self._stream()
        f16 = (mode or self._mode) != "f32"
        verdict = LaunchVerdict(stream) if f16 else None
        emb, logits, counts = self.launch(parts, hop, step, want_embeddings, True, mode=mode, verdict=verdict)

        def redo_chunk(i: int, embeddings: bool):
            return self._redo_on(stream, parts[i], hop, step, embeddings) if f16 else None

        out = [DeviceResult(t, stream, self, redo_chunk(i, False), verdict) for i, t in enumerate(torch.split(logits, counts))]
        if want_embeddings:
            return out, [DeviceResult(t, stream, self, redo_chunk(i, True), verdict)
                         for i, t in enumerate(torch.split(emb, counts))]
        return out

    def embed(self, samples, framehop_s: float) -> DeviceResult:
        hop, step = hop_samples(framehop_s), patch_step(framehop_s)
        x = self.to_device(samples)
        stream = self._stream()
        f16 = self._mode != "f32"
        verdict = LaunchVerdict(stream) if f16 else None
        emb, _ = self.run(x, hop, step, True, False, verdict=verdict)
        return DeviceResult(emb, stream, self, self._redo_on(stream, x, hop, step, True) if f16 else None, verdict)

    def predict(self, samples, framehop_s: float, out: Optional[torch.Tensor] = None) -> DeviceResult:
        """``out``: optional caller-owned ``[W, n_classes]`` device rows to write the logits into."""
        hop, step = hop_samples(framehop_s), patch_step(framehop_s)
        x = self.to_device(samples)
        stream = self._stream()
        f16 = self._mode != "f32"
        verdict = LaunchVerdict(stream) if f16 else None
        _, logits = self.run(x, hop, step, False, True, out=out, verdict=verdict)
        return DeviceResult(logits, stream, self, self._redo_on(stream, x, hop, step, False, out=out) if f16 else None, verdict)

    def stage_tap(self, samples, hop: int, step: int, stage: int, windows: int) -> torch.Tensor:
        """Test hook: NHWC activation after CNN stage ``stage`` for the first ``windows`` windows."""
        x = self.to_device(samples)
        h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.bd_stage_shape(stage, C.byref(h), C.byref(w), C.byref(c)))
        out = torch.empty((windows, h.value, w.value, c.value), dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            ws_bytes = _lib.check(self._lib.bd_workspace_bytes(self._handle, x.numel(), hop, step))
            ws = self._ws(ws_bytes)
            _lib.check(self._lib.bd_stage_tap(self._handle, x.data_ptr(), x.numel(), hop, step, ws.data_ptr(),
                                              ws.numel(), stage, windows, out.data_ptr(),
                                              self._stream().cuda_stream))
        return out

    # ------------------------------------------------------------------ timing
    def profile_enable(self, on: bool) -> None:
        with self._lock:
            _lib.check(self._lib.bd_profile_enable(self._handle, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * _lib.PROFILE_SLOTS)()
        cnt = (C.c_int64 * _lib.PROFILE_SLOTS)()
        with self._lock:
            _lib.check(self._lib.bd_profile_read(self._handle, ms, cnt, _lib.PROFILE_SLOTS))
        return np.array(ms[:], dtype=np.float64), np.array(cnt[:], dtype=np.int64)
