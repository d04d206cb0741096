"""One launch set per 1 h recording (its four 1024-window chunks through bd_predict_chunks) with CNN passes of 1024 / 2048 /
4096 windows, two analyzer streams, against one call per chunk (bench.py's headline loop).  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(2)]
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
hop, step = hop_samples(0.96), patch_step(0.96)
N = 57_600_000
files = [torch.randn(N, device=dev) * 0.1 for _ in range(3)]
edges = [(i * 1024 * hop, min((i + 1) * 1024 * hop, N)) for i in range(4)]
out = [torch.empty((3750, 13), device=dev) for _ in range(2)]


def per_chunk(n):
    for r in range(n):
        at = 0
        for b, (a, e) in enumerate(edges):
            j = b % 2
            w = 1024 if b < 3 else 678
            with torch.cuda.stream(streams[j]):
                engs[j].launch([files[r % 3][a:e]], hop, step, False, True, out=out[r % 2][at:at + w])
            at += w


def per_recording(n):
    for r in range(n):
        j = r % 2
        with torch.cuda.stream(streams[j]):
            engs[j].launch([files[r % 3][a:e] for a, e in edges], hop, step, False, True, out=out[j])


def timed(fn, n=60):
    fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n * 3750 / (time.perf_counter() - t0)


print(f"one call per chunk            : {timed(per_chunk) / 1e6:.3f} M windows/s")
for g in (1024, 2048, 4096):
    for e in engs:
        e.set_group_windows(g)
    print(f"one call per recording, G={g:4d}: {timed(per_recording) / 1e6:.3f} M windows/s")
