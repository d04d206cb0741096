"""Board power and shader clock while one arithmetic mode runs for a few seconds on three streams (hwmon, as bench.py); with a
third argument, two fusion layouts alternate in the same process, three rounds: `ab` = exact-f32 layers 4-14 fused (bd_set_fusion
separable = 6) against the default; a number = that separable code against the default (9: exact-f32 without the layer-4 kernel).
GPU box.    python tools/mode_clock.py [f32|f16x3|f16] [seconds=3] [ab|code|-] [streams=3]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import bench  # noqa: E402
from buzzdetect_amd.engine import HipEngine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
NS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
engs = [HipEngine(device=0) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in engs]
for e in engs:
    e.set_pointwise_mode(mode)
x = (torch.randn(15360 * 1023 + 15600, generator=torch.Generator().manual_seed(1)) * 0.1).cuda()
for e in engs:
    e.predict(x, 0.96)
torch.cuda.synchronize()


def run(label):
    n = 0
    with bench.PowerWatch(torch, 0) as watch:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(30):
                with torch.cuda.stream(streams[n % NS]):
                    engs[n % NS].predict(x, 0.96)
                n += 1
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    w = watch.summary() or {}
    print(f"mode {mode} {label}: {n * 1024 / dt / 1e6:.3f} M windows/s on {NS} streams; {w.get('avg_W')} W, {w.get('sclk_MHz_avg')} MHz")


if len(sys.argv) > 3 and sys.argv[3] != "-":
    other = 6 if sys.argv[3] == "ab" else int(sys.argv[3])
    for rnd in range(3):
        for code in (other, 1):
            for e in engs:
                e.set_fusion(True, code)
            run(f"separable fusion {code}")
else:
    run("")
