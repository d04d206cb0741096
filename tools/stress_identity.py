"""Stress check: the headline loop on N analyzer streams for a while, EVERY batch's logits compared on the device with the
logits the same batch gave on an idle GPU (bit for bit).  The fused kernels overlay LDS tiles and keep tiles resident across
layers inside one launch; this looks for an ordering mistake that only shows under load.  One mismatch counter per stream
(a shared one is a non-atomic read-modify-write from several streams and can lose a count), side streams ordered behind
the set-up, and a negative control first: one reference row corrupted must be reported for exactly the batches that use it.
GPU box.    python tools/stress_identity.py [streams=3] [seconds=30] [separable fusion code, default 1] [f16x3|f32|f16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
separable = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(n_streams)]
if separable != 1:
    from buzzdetect_amd import _lib
    for e in engs:
        _lib.check(e._lib.bd_set_fusion(e._handle, 3, separable))
for e in engs:
    e.set_pointwise_mode(mode)
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
ring = [[torch.empty((1024, 13), device=dev) for _ in range(4)] for _ in range(n_streams)]


def run(reference, seconds, rounds=30):
    bad = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n_streams)]     # one counter per stream
    torch.cuda.synchronize()
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    k = batches = 0
    t0 = time.perf_counter()
    while True:
        for r in range(rounds):
            for b, (a, e) in enumerate(edges):
                j = k % n_streams
                out = ring[j][(k // n_streams) % 4][:sizes[b]]
                with torch.cuda.stream(streams[j]):
                    engs[j].launch([files[r % 3][a:e]], hop, step, False, True, out=out)
                    bad[j] += (out != reference[(r % 3, b)]).any().to(torch.int64)
                k += 1
                batches += 1
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            break
    return batches, sum(int(c.item()) for c in bad), time.perf_counter() - t0


# negative control: one element of one reference wrong -> exactly the batches of (file 1, batch 2) are reported
broken = dict(ref)
broken[(1, 2)] = ref[(1, 2)].clone()
broken[(1, 2)][500, 7] += 1.0
n, wrong, _ = run(broken, 0.0, rounds=30)
expect = n // 12
print(f"negative control: {wrong} of {n} batches reported, {expect} expected")
if wrong != expect:
    sys.exit(2)
batches, wrong, dt = run(ref, seconds)
print(f"{batches} batches on {n_streams} streams in {dt:.1f} s ({batches * 937.5 / dt / 1e6:.2f} M windows/s incl. the comparisons), "
      f"mode {mode}, separable fusion {separable}: {wrong} batches differ from their idle-GPU result")
sys.exit(1 if wrong else 0)
