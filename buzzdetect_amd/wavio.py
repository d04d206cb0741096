"""PCM / float WAV files by positioned reads: the streamer's view of a recording.

The reference reads audio through soundfile / PyAV (src/stream/audio.py:24-44: ``seek(frame)``, ``read(n, float32)``
-> ``[n, channels]`` in [-1, 1)); codecs are out of scope here, uncompressed WAV is not.  The file is parsed once
(RIFF chunks; PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE with either sub-format) and then read with
positioned reads (``os.preadv`` straight into the caller's buffer: no intermediate copy, no page faults of a mapping, the
GIL is released): 16-bit data goes to the device as it lies in the file (the device stage scales by 1/32768 like
libsndfile's float read), everything else is converted to float32 on the way into the pinned staging buffer.
"""
from __future__ import annotations

import os
import struct

import numpy as np

FORMAT_PCM, FORMAT_FLOAT, FORMAT_EXTENSIBLE = 1, 3, 0xFFFE


class WavFormatError(ValueError):
    pass


class WavTrack:
    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            head = f.read(12)
            if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
                raise WavFormatError(f"{path}: not a RIFF/WAVE file")
            fmt = None
            data_off = data_len = data_declared = None
            size = f.seek(0, 2)
            f.seek(12)
            while f.tell() + 8 <= size:
                cid, clen = struct.unpack("<4sI", f.read(8))
                at = f.tell()
                if cid == b"fmt ":
                    fmt = f.read(min(clen, 40))
                elif cid == b"data":
                    # a recorder that died mid-file leaves a count longer than the file: `frames` below is what can be
                    # read, `frames_declared` what the header promises (libsndfile's .frames, the reference's duration)
                    data_off, data_len, data_declared = at, min(clen, size - at), clen
                    break
                f.seek(at + clen + (clen & 1))
        if fmt is None or len(fmt) < 16 or data_off is None:
            raise WavFormatError(f"{path}: missing fmt or data chunk")
        tag, channels, rate, _, block, bits = struct.unpack("<HHIIHH", fmt[:16])
        if tag == FORMAT_EXTENSIBLE and len(fmt) >= 26:
            tag = struct.unpack("<H", fmt[24:26])[0]
        if tag not in (FORMAT_PCM, FORMAT_FLOAT) or channels < 1 or rate < 1:
            raise WavFormatError(f"{path}: unsupported WAVE format tag {tag:#x}")
        width = bits // 8
        if (tag == FORMAT_PCM and width not in (1, 2, 3, 4)) or (tag == FORMAT_FLOAT and width not in (4, 8)) or block != width * channels:
            raise WavFormatError(f"{path}: unsupported sample layout ({bits} bits, block align {block})")
        self.samplerate, self.channels = int(rate), int(channels)
        self._tag, self._width = tag, width
        self.frames = data_len // block
        # 0xFFFFFFFF is the "length unknown" convention of streaming writers (and RF64's placeholder), 0x7FFFF000 .. 0x7FFFFFFF
        # what sox / ffmpeg write into a pipe: no statement about the length at all - the file's own length is the only one
        # there is.  (Any other count larger than the file is taken as a recorder's promise that was cut short: the reference's
        # bad-read rule applies, pipeline._bad_read.  libsndfile may clamp such a count to the file for WAV - not verifiable
        # here, soundfile is absent; what a clamped count changes is the log line and the planned tail chunks, which are
        # dropped unread, never a row of the results.)
        placeholder = data_declared >= 0x7FFFF000
        self.frames_declared = self.frames if placeholder else max(self.frames, data_declared // block)
        self._data_off, self._block = data_off, block
        self._fd = os.open(path, os.O_RDONLY)
        self._pos = 0

    def __del__(self):
        self.close()

    @property
    def duration(self) -> float:
        """Seconds the HEADER declares (soundfile's frames / samplerate, what the reference plans its chunks over,
        src/stream/worker.py:84-107); a file cut short holds less: `duration_readable`."""
        return self.frames_declared / self.samplerate

    @property
    def duration_readable(self) -> float:
        return self.frames / self.samplerate

    @property
    def is_s16(self) -> bool:
        return self._tag == FORMAT_PCM and self._width == 2

    def seek(self, frame: int) -> None:
        self._pos = min(max(int(frame), 0), self.frames)

    def _raw(self, n: int) -> np.ndarray:
        n = max(0, min(int(n), self.frames - self._pos))
        data = os.pread(self._fd, n * self._block, self._data_off + self._pos * self._block) if n else b""
        n = len(data) // self._block                       # a file cut short after its header was written
        self._pos += n
        return np.frombuffer(data, np.uint8, n * self._block)

    def read_s16(self, n: int) -> np.ndarray:
        """[frames, channels] int16 as it lies in the file (16-bit files only): no conversion."""
        if not self.is_s16:
            raise WavFormatError("read_s16 on a file that is not 16-bit PCM")
        return self._raw(n).view("<i2").reshape(-1, self.channels)

    def read_raw_into(self, frame: int, n: int, out: np.ndarray) -> int:
        """``n`` frames from ``frame`` on, as they lie in the file, into the uint8 buffer ``out`` (positioned read; does
        not move the seek position, so several threads may share one track).  Returns the frames actually read."""
        frame = min(max(int(frame), 0), self.frames)
        n = max(0, min(int(n), self.frames - frame))
        want = n * self._block
        if want == 0:
            return 0
        view = memoryview(out)[:want]
        got, at = 0, self._data_off + frame * self._block
        while got < want:
            r = os.preadv(self._fd, [view[got:]], at + got)
            if r <= 0:
                break
            got += r
        return got // self._block

    @property
    def bytes_per_frame(self) -> int:
        return self._block

    def file_range(self, frame: int, n: int):
        """(file descriptor, byte offset, byte count) of ``n`` frames from ``frame`` on, clipped to what the file holds:
        for readers that do their own positioned reads (bd_stager_read)."""
        frame = min(max(int(frame), 0), self.frames)
        n = max(0, min(int(n), self.frames - frame))
        return self._fd, self._data_off + frame * self._block, n * self._block

    def read(self, n: int, keep_s16: bool = False) -> np.ndarray:
        """[frames, channels] float32 in [-1, 1) (libsndfile's scaling); with ``keep_s16`` 16-bit files come back as the
        int16 view instead (the device stage scales by 1/32768 itself, halving the host-to-device bytes)."""
        if keep_s16 and self.is_s16:
            return self.read_s16(n)
        return self.convert(self._raw(n))

    def convert(self, raw: np.ndarray) -> np.ndarray:
        """Bytes as read by ``read_raw_into`` -> [frames, channels] float32 in [-1, 1) (any supported format)."""
        if self._tag == FORMAT_FLOAT:
            a = raw.view("<f4" if self._width == 4 else "<f8").astype(np.float32)
        elif self._width == 2:
            a = raw.view("<i2").astype(np.float32) / 32768.0
        elif self._width == 4:
            a = (raw.view("<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif self._width == 1:
            a = (raw.astype(np.float32) - 128.0) / 128.0
        else:
            b = raw.reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            a = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        return a.reshape(-1, self.channels)

    def close(self) -> None:
        fd, self._fd = getattr(self, "_fd", None), None
        if fd is not None:
            os.close(fd)
