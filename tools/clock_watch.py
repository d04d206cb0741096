"""Shader clock and board power while the headline loop runs (sysfs hwmon, 20 ms polls): is the steady state power-limited?
GPU box.    python tools/clock_watch.py [streams]"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def hwmon_of(device_index):
    props = torch.cuda.get_device_properties(device_index)
    want = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
    for card in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(card)).startswith(want):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if hw:
                return hw[0]
    raise RuntimeError(f"no hwmon for PCI {want}")


def read(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


hw = hwmon_of(0)
print(f"hwmon {hw}: power limit {read(os.path.join(hw, 'power1_cap')) / 1e6:.0f} W", flush=True)

dev = torch.device("cuda", 0)
engs = [HipEngine(device=0) for _ in range(n_streams)]
streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
hop, step = hop_samples(0.96), patch_step(0.96)
N = 57_600_000
files = [torch.randn(N, device=dev) * 0.1 for _ in range(3)]
edges = [(i * 1024 * hop, min((i + 1) * 1024 * hop, N)) for i in range(4)]
out = [torch.empty((3750, 13), device=dev) for _ in range(2)]
samples = []
stop = threading.Event()


def poll():
    t0 = time.perf_counter()
    while not stop.is_set():
        samples.append((time.perf_counter() - t0, read(os.path.join(hw, "freq1_input")), read(os.path.join(hw, "power1_input")),
                        read(os.path.join(hw, "temp2_input"))))
        time.sleep(0.02)


def recordings(n):
    k = 0
    for r in range(n):
        at = 0
        for b, (a, e) in enumerate(edges):
            j = k % n_streams
            k += 1
            w = 1024 if b < 3 else 678
            with torch.cuda.stream(streams[j]):
                engs[j].launch([files[r % 3][a:e]], hop, step, False, True, out=out[r % 2][at:at + w])
            at += w


th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(0.5)                       # idle samples
marks = []
for burst in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recordings(600)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    marks.append(dt)
    print(f"burst {burst}: 600 recordings in {dt:.3f} s = {600 * 3750 / dt / 1e6:.3f} M windows/s", flush=True)
time.sleep(0.5)
stop.set()
th.join()
for t, f, p, c in samples[::5]:
    print(f"{t:7.3f} s  sclk {(f or 0) / 1e6:6.0f} MHz  power {(p or 0) / 1e6:6.0f} W  temp {(c or 0) / 1000:.0f} C")
