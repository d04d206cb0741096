#!/usr/bin/env python3
"""The host-resident s16 leg of bench.py with different ring depths / numbers of compute streams (where does the gap between
the PCIe rate, 56 GB/s = 1.78 M windows/s of 16-bit PCM, and the measured 1.2 M come from?).

    python tools/h2d_leg_sweep.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import torch

import bench
from buzzdetect_amd.engine import HipEngine, hop_samples


def leg(engines, streams, device, hop, ring_depth, what, reuse_events=True):
    n = bench.WINDOWS_PER_BATCH * hop
    host = []
    for i in range(2):
        x = bench.synthetic_audio(device, n, 777 + i)
        host.append((x * 32768.0).round().clamp_(-32768, 32767).to(torch.int16).cpu().pin_memory())
    copy_stream = torch.cuda.Stream(device)
    ring = [torch.empty(n, dtype=torch.int16, device=device) for _ in range(ring_depth)]
    copied = [torch.cuda.Event() for _ in ring]
    consumed = [None] * len(ring)
    spare = [torch.cuda.Event() for _ in ring]        # one event per slot, recorded again and again

    def run(count):
        for i in range(count):
            slot = i % len(ring)
            with torch.cuda.stream(copy_stream):
                if consumed[slot] is not None:
                    copy_stream.wait_event(consumed[slot])
                if what != "compute":
                    ring[slot].copy_(host[i % 2], non_blocking=True)
                copied[slot].record(copy_stream)
            s = streams[i % len(streams)]
            e = engines[i % len(engines)]
            with torch.cuda.stream(s):
                s.wait_event(copied[slot])
                if what != "copy":
                    pcm = e.resample(ring[slot], 16000, 16000)
                    e.predict(pcm, 0.96)
                ev = spare[slot] if reuse_events else torch.cuda.Event()
                ev.record(s)
                consumed[slot] = ev

    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(48)
    torch.cuda.synchronize()
    return 48 * bench.WINDOWS_PER_BATCH / (time.perf_counter() - t0)


def main():
    device = torch.device("cuda", 0)
    hop = hop_samples(0.96)
    for n_streams in (2,):
        engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=0) for _ in range(n_streams)]
        streams = [torch.cuda.Stream(device) for _ in engines]
        for depth in (3, 6):
            for reuse in (True, False):
                for what in ("both", "copy", "compute"):
                    r = leg(engines, streams, device, hop, depth, what, reuse)
                    print(f"{n_streams} compute streams, ring {depth}, {'one event per slot' if reuse else 'new event per batch'}, "
                          f"{what:8s}: {r:,.0f} windows/s", flush=True)


if __name__ == "__main__":
    main()
