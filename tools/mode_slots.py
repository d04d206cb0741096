"""Per-launch times of one arithmetic mode of the 1x1 convolutions (engine marker events), one 1024-window batch repeated.
GPU box.    python tools/mode_slots.py [f32|f16x3|f16] [repeats=20] [bd_set_fusion separable code]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = HipEngine(device=0)
eng.set_pointwise_mode(mode)
if len(sys.argv) > 3:                      # e.g. 6: the exact-f32 mode fused per layer
    eng.set_fusion(True, int(sys.argv[3]))
x = (torch.randn(15360 * 1023 + 15600, generator=torch.Generator().manual_seed(1)) * 0.1).cuda()
for _ in range(3):
    eng.predict(x, 0.96)
torch.cuda.synchronize()
eng.profile_read()
eng.profile_enable(True)
for _ in range(reps):
    eng.predict(x, 0.96)
torch.cuda.synchronize()
eng.profile_enable(False)
ms, n = eng.profile_read()
names = ["frontend", "conv1"] + [f"{'dw' if k % 2 == 0 else 'pw'}{2 + k // 2}" for k in range(26)] + ["pool+head"]
tot = 0.0
for i, (m, c) in enumerate(zip(ms, n)):
    if c:
        print(f"slot {i:2d} {names[i]:10s} {1e3 * m / c:8.1f} us x {c // reps} per batch")
        tot += m / reps
print(f"mode {mode}: {1e3 * tot:.1f} us per 1024-window batch = {1024 / tot / 1e3:.3f} M windows/s on one stream")
