"""Time the device resampler alone (HIP events): 48 kHz stereo s16 / 32 kHz mono s16 / 44.1 kHz stereo f32 -> 16 kHz mono,
and the rate-preserving s16 -> f32 conversion, on one 1024-window batch.    python tools/resample_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine  # noqa: E402

eng = HipEngine()
quality = sys.argv[1] if len(sys.argv) > 1 else "hq"
only = int(sys.argv[2]) if len(sys.argv) > 2 else None          # one input rate only (profiling runs)
eng.set_resample_quality(quality)
print(f"quality {quality}")
n16 = 15360 * 1024
gen = torch.Generator(device="cuda").manual_seed(3)
for name, rate, ch, dtype in (("48 kHz stereo s16", 48000, 2, torch.int16), ("32 kHz mono s16", 32000, 1, torch.int16),
                              ("48 kHz mono f32", 48000, 1, torch.float32), ("96 kHz stereo s16", 96000, 2, torch.int16),
                              ("44.1 kHz stereo f32", 44100, 2, torch.float32), ("24 kHz mono s16", 24000, 1, torch.int16),
                              ("16 kHz mono s16 (convert)", 16000, 1, torch.int16)):
    if only is not None and (rate != only or ch != 2):
        continue
    n = n16 * rate // 16000
    x = torch.randn((n, ch), generator=gen, device="cuda") * 0.2
    x = (x * 32768).clamp(-32768, 32767).to(torch.int16) if dtype == torch.int16 else x
    x = x.contiguous()
    for _ in range(2):
        y = eng.resample(x, rate)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        y = eng.resample(x, rate)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    nbytes = x.numel() * x.element_size() + y.numel() * 4
    print(f"{name:28s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  ({nbytes / us / 1e3 / 80:.1f} % of 8 TB/s)")
    del x, y
