"""Do host-to-device copies and the hot path's kernels get in each other's way?  Independent copies (pinned -> device,
31.5 MB = one 1024-window batch of 16-bit PCM) on a copy stream while two analyzer streams run predict() on resident PCM.
Run with HSA_ENABLE_SDMA=0 / 1 to see which engine moves the bytes.
    python tools/h2d_overlap_probe.py"""
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
copy_stream = torch.cuda.Stream(dev)
nbytes = 31_457_280
host = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
dst = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]
x = torch.randn(15360 * 1024, device=dev) * 0.1
outs = [torch.empty((1024, 13), device=dev) for _ in range(2)]
N = 60


def copies(n, pieces=1):
    with torch.cuda.stream(copy_stream):
        step = nbytes // pieces
        for i in range(n):
            for p in range(pieces):
                dst[i % 3][p * step:(p + 1) * step].copy_(host[i % 3][p * step:(p + 1) * step], non_blocking=True)


def compute(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]):
            engs[i % 2].predict(x, 0.96, out=outs[i % 2])


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


copies(3); compute(4)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA", "(default)"))
tc = timed(lambda: copies(N))
print(f"copies alone : {N * nbytes / tc / 1e9:6.1f} GB/s  ({1e3 * tc / N:.3f} ms per batch)")
tk = timed(lambda: compute(N))
print(f"compute alone: {N * 1024 / tk / 1e6:6.3f} M windows/s ({1e3 * tk / N:.3f} ms per batch)")
for pieces in (1, 4, 16):
    ev_c, ev_k = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    copy_stream.wait_event(start); streams[1].wait_event(start)
    t0 = time.perf_counter()
    copies(N, pieces)
    ev_c.record(copy_stream)
    compute(N)
    streams[0].wait_stream(streams[1])
    ev_k.record(streams[0])
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"together ({pieces:2d} pieces per copy): copies done after {start.elapsed_time(ev_c):7.1f} ms "
          f"({N * nbytes / start.elapsed_time(ev_c) / 1e6:5.1f} GB/s), compute done after {start.elapsed_time(ev_k):7.1f} ms "
          f"({N * 1024 / start.elapsed_time(ev_k) / 1e3:5.3f} M windows/s); wall {1e3 * tot:.1f} ms")

# ---- the bench leg's dependency pattern, piece by piece (ring of 3 device buffers) ----
copied = [torch.cuda.Event() for _ in range(8)]
consumed = [torch.cuda.Event() for _ in range(8)]
ring = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(8)]


def leg(n, depth, wait_consumed, wait_copied, convert):
    seen = [False] * depth
    for i in range(n):
        slot = i % depth
        with torch.cuda.stream(copy_stream):
            if wait_consumed and seen[slot]:
                copy_stream.wait_event(consumed[slot])
            ring[slot].copy_(host[i % 3], non_blocking=True)
            copied[slot].record(copy_stream)
        s = streams[i % 2]
        with torch.cuda.stream(s):
            if wait_copied:
                s.wait_event(copied[slot])
            e = engs[i % 2]
            pcm = e.resample(ring[slot].view(torch.int16), 16000, 16000) if convert else x
            e.predict(pcm, 0.96, out=outs[i % 2])
            consumed[slot].record(s)
            seen[slot] = True


for depth in (3, 6):
    for wc, wk, cv in ((False, False, False), (True, False, False), (False, True, False), (True, True, False), (True, True, True)):
        leg(6, depth, wc, wk, cv)
        t = timed(lambda: leg(200, depth, wc, wk, cv))
        print(f"ring {depth}: copy waits for consumed={wc!s:5} compute waits for copied={wk!s:5} convert={cv!s:5}: "
              f"{200 * 1024 / t / 1e6:5.3f} M windows/s, {200 * nbytes / t / 1e9:5.1f} GB/s")
