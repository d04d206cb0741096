#!/usr/bin/env python3
"""GPU box: throughput on the reference's default 199.68 s chunks (208 windows), one predict() per chunk vs
bd_predict_batch over 5 chunks (1040 windows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import torch
from buzzdetect_amd.engine import HipEngine

def main():
    eng = HipEngine()
    dev = eng.device
    g = torch.Generator(device=dev).manual_seed(1)
    chunks = [0.1 * torch.randn(3_194_880, generator=g, device=dev) for _ in range(5)]
    for name, fn in (("per-chunk predict", lambda: [eng.predict(c, 0.96) for c in chunks]),
                     ("predict_batch x5", lambda: eng.predict_batch(chunks, 0.96))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 40
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(f"{name:20s} {1040 / dt:10.0f} windows/s  ({dt * 1e3:.3f} ms per 5 chunks)")

if __name__ == "__main__":
    main()
