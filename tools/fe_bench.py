"""Time the front-end kernel alone: 98 304 frames (the 1024-window batch), HIP events around N launches.
    python tools/fe_bench.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine  # noqa: E402
from oracle import yamnet_oracle as O  # noqa: E402
from buzzdetect_amd import weights as W  # noqa: E402

reps = int(sys.argv[-1]) if len(sys.argv) > 1 else 200
eng = HipEngine()
hop = 15360
n = hop * 1024
gen = torch.Generator(device="cuda").manual_seed(5)
xs = [0.1 * torch.randn(n, generator=gen, device="cuda") for _ in range(4)]
for x in xs:
    out = eng.frontend(x, hop)
torch.cuda.synchronize()
t = out.shape[0]
# correctness on a prefix against the f64 oracle
m = 15600 + 3 * hop
ref = O.log_mel(O.pad_waveform(xs[0][:m].cpu().numpy(), hop), W.load_mel("yamnet_k2"), np.float64)
got = eng.frontend(xs[0][:m].clone(), hop).cpu().numpy()
print(f"frames {t}; max|dlogmel| vs f64 oracle on {got.shape[0]} frames: {np.abs(got - ref).max():.2e}")
buf = torch.empty((t, 64), dtype=torch.float32, device="cuda")
import ctypes as C
from buzzdetect_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for i in range(reps):
        lib.bd_frontend(eng._handle, xs[i % 4].data_ptr(), n, hop, buf.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"logmel_kernel: {us:.1f} us per {t} frames = {t * 896 / us / 1e6:.2f} TB/s algorithmic "
          f"({t * 896 / us / 1e6 / 8 * 100:.1f} % of 8 TB/s)")
