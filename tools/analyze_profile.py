#!/usr/bin/env python3
"""Where analyze() spends its wall clock: a generated 16-bit WAV on tmpfs, three calls with prebuilt engines, busy time per
stage (summed over the threads of that stage) from wrappers around the stage functions.

    python tools/analyze_profile.py [hours=24] [chunklength=600] [hop=1.0]
"""
import collections
import os
import shutil
import sys
import tempfile
import threading
import time
import wave

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import torch

import bench
from buzzdetect_amd import fastcsv, pipeline, results, wavio
from buzzdetect_amd.analyze import analyze
from buzzdetect_amd.engine import HipEngine

busy = collections.defaultdict(float)
calls = collections.defaultdict(int)
lock = threading.Lock()


def timed(owner, name, label):
    fn = getattr(owner, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = time.perf_counter() - t
            with lock:
                busy[label] += d
                calls[label] += 1
    setattr(owner, name, wrapper)


def main():
    hours = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    chunk = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
    hop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    dev = torch.device("cuda", 0)
    root = tempfile.mkdtemp(prefix="bd_prof_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        audio = os.path.join(root, "audio")
        os.makedirs(audio)
        hour = bench.synthetic_audio(dev, bench.FILE_SAMPLES, 4242)
        block = (hour * 32768.0).round().clamp_(-32768, 32767).to(torch.int16).cpu().numpy().astype("<i2").tobytes()
        with wave.open(os.path.join(audio, f"synthetic_{hours}h.wav"), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            for _ in range(hours):
                w.writeframes(block)
        engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=0) for _ in range(2)]
        timed(wavio.WavTrack, "read_raw_into", "reader: preadv into the pinned slot")
        timed(pipeline.PinnedRing, "acquire", "reader: waiting for / allocating a pinned slot")
        timed(HipEngine, "predict_batch", "analyzer: enqueue predict_batch")
        timed(HipEngine, "resample", "analyzer: enqueue resample / convert")
        timed(fastcsv, "rows", "writer: CSV text")
        timed(results.ResultFile, "append_text", "writer: append to the partial file")
        timed(results.ResultFile, "finalize_sorted", "writer: final file")
        timed(torch.cuda.Event, "synchronize", "event.synchronize (analyzer settle + writer)")
        for i in range(3):
            busy.clear()
            calls.clear()
            out = os.path.join(root, f"out{i}")
            t0 = time.perf_counter()
            rep = analyze("model_general_v3", classes_out="all", framehop_prop=hop, chunklength=chunk, dir_audio=audio,
                          dir_out=out, embeddername="yamnet_k2", engines=engines, rank=0, world_size=1)
            sec = time.perf_counter() - t0
            print(f"call {i}: {sec:.3f} s, {rep.audio_seconds / sec:,.0f} audio-s/s, {rep.windows / sec:,.0f} windows/s, "
                  f"{rep.chunks} chunks", flush=True)
            for k in sorted(busy, key=busy.get, reverse=True):
                print(f"    {busy[k] * 1e3:9.1f} ms in {calls[k]:5d} calls  {k}")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
