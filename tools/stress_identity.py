"""Stress check: the headline loop on N analyzer streams for a while, EVERY batch's logits compared on the device with the
logits the same batch gave on an idle GPU (bit for bit).  The fused kernels hand data from layer to layer through global
memory inside one launch (layers 8-11) and overlay LDS tiles; this looks for an ordering mistake that only shows under load.
GPU box.    python tools/stress_identity.py [streams=3] [seconds=30]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(n_streams)]
streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
hop, step = hop_samples(0.96), patch_step(0.96)
N = 57_600_000
g = torch.Generator(device="cpu").manual_seed(5)
files = [(torch.randn(N, generator=g) * 0.1).to(dev) for _ in range(3)]
edges = [(i * 1024 * hop, min((i + 1) * 1024 * hop, N)) for i in range(4)]
sizes = [1024, 1024, 1024, 678]

# references on an idle GPU, one stream
ref = {}
for f in range(3):
    for b, (a, e) in enumerate(edges):
        out = torch.empty((sizes[b], 13), device=dev)
        engs[0].launch([files[f][a:e]], hop, step, False, True, out=out)
        torch.cuda.synchronize()
        ref[(f, b)] = out
bad = torch.zeros(1, dtype=torch.int64, device=dev)
ring = [[torch.empty((1024, 13), device=dev) for _ in range(4)] for _ in range(n_streams)]
k = 0
batches = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds:
    for r in range(30):
        for b, (a, e) in enumerate(edges):
            j = k % n_streams
            out = ring[j][(k // n_streams) % 4][:sizes[b]]
            with torch.cuda.stream(streams[j]):
                engs[j].launch([files[r % 3][a:e]], hop, step, False, True, out=out)
                bad += (out != ref[(r % 3, b)]).any().to(torch.int64)
            k += 1
            batches += 1
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{batches} batches on {n_streams} streams in {dt:.1f} s ({batches * 937.5 / dt / 1e6:.2f} M windows/s incl. the comparisons): "
      f"{int(bad.item())} batches differ from their idle-GPU result")
sys.exit(1 if int(bad.item()) else 0)
