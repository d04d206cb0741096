"""Same-process A/B of two bd_set_fusion settings on bench.py's headline loop (one call per 1024-window chunk of a 1 h
recording, N analyzer streams with an engine each), alternating the settings so that box and clock drift cancel, plus the
logits digest of each setting (they must be equal: fused paths are bit-identical).  GPU box.

    python tools/fusion_ab.py [separable_a separable_b [streams]]        default: 1 3 4
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

a_code = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b_code = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 4

dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(n_streams)]
streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
hop, step = hop_samples(0.96), patch_step(0.96)
N = 57_600_000
g = torch.Generator(device="cpu").manual_seed(11)
files = [(torch.randn(N, generator=g) * 0.1).to(dev) for _ in range(3)]
edges = [(i * 1024 * hop, min((i + 1) * 1024 * hop, N)) for i in range(4)]
out = [torch.empty((3750, 13), device=dev) for _ in range(2)]


def recordings(n):
    k = 0
    for r in range(n):
        at = 0
        for b, (a, e) in enumerate(edges):
            j = (k + (r if os.environ.get("BD_AB_ROTATE") else 0)) % n_streams     # BD_AB_ROTATE=1: the short last batch of a
            k += 1                                                                   # recording moves from stream to stream
            w = 1024 if b < 3 else 678
            with torch.cuda.stream(streams[j]):
                engs[j].launch([files[r % 3][a:e]], hop, step, False, True, out=out[r % 2][at:at + w])
            at += w


def setting(e, code):
    """code < 100: bd_set_fusion separable code; 100 + v: default fusion with bd_set_pointwise_variant(13, v) (10 = 8-wave tile kernel)"""
    e.set_fusion(True, code if code < 100 else 1)
    e.set_pointwise_variant(13, code - 100 if code >= 100 else 0)


def rate(code, n=100):
    for e in engs:
        setting(e, code)
    recordings(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recordings(n)
    torch.cuda.synchronize()
    return n * 3750 / (time.perf_counter() - t0)


def digest(code):
    setting(engs[0], code)
    with torch.cuda.stream(streams[0]):
        engs[0].launch([files[0][:edges[0][1]]], hop, step, False, True, out=out[0][:1024])
    torch.cuda.synchronize()
    return hashlib.sha256(np.ascontiguousarray(out[0][:1024].cpu().numpy()).tobytes()).hexdigest()[:16]


print(f"digest separable={a_code}: {digest(a_code)}   separable={b_code}: {digest(b_code)}", flush=True)
ra, rb = [], []
for rep in range(5):
    ra.append(rate(a_code))
    rb.append(rate(b_code))
    print(f"rep {rep}: separable={a_code} {ra[-1] / 1e6:.3f} M windows/s   separable={b_code} {rb[-1] / 1e6:.3f} M windows/s", flush=True)
ma, mb = float(np.median(ra)), float(np.median(rb))
print(f"median: separable={a_code} {ma / 1e6:.3f} M   separable={b_code} {mb / 1e6:.3f} M   ratio {ma / mb:.4f}")
