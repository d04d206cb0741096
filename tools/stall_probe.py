"""Looks for one-off host stalls in a stream of predict() calls (GPU box): prints every call whose enqueue took > 1 ms."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine

dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(2)]
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
x = torch.randn(15360 * 1024, device=dev) * 0.1
outs = [torch.empty((1024, 13), device=dev) for _ in range(2)]
for i in range(8):
    with torch.cuda.stream(streams[i % 2]):
        engs[i % 2].predict(x, 0.96, out=outs[i % 2])
torch.cuda.synchronize()
t_all = time.perf_counter()
slow = []
for i in range(400):
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[i % 2]):
        engs[i % 2].predict(x, 0.96, out=outs[i % 2])
    dt = time.perf_counter() - t0
    if dt > 1e-3:
        slow.append((i, round(1e3 * dt, 2)))
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"400 calls: host {1e3 * host:.1f} ms, total {1e3 * tot:.1f} ms ({1e6 * tot / 400:.0f} us per call); calls over 1 ms: {slow[:20]}")
