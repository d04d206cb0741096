"""Sustained shader clock, board power and time per launch of each kernel of the hot path when it runs ALONE, back to back
(developer build: -DBD_KERNEL_TRACE, BD_REPEAT_SLOT / BD_REPEAT_N make one profile slot's kernel run N times per pass), and of
the whole path.  Answers: which kernels are held back by the board's power limit rather than by their own pipeline?  GPU box.

    hipcc ... -DBD_KERNEL_TRACE -o buzzdetect_amd/csrc/libtrace.so ...      (see tools/profile_round.sh)
    python tools/power_profile.py [seconds per kernel, default 2.0]
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# the default launch set (bd_set_fusion 3 / 1): every launch reads one buffer and writes another, so a repeated launch sees the same input
SLOTS = [(-1, "whole path"), (0, "frontend"), (5, "stem(1-3)"), (7, "sep4+dw5"), (13, "pw5-pw7"), (23, "sep8-12+dw13"),
         (25, "pw13+dw14"), (27, "pw14+pool"), (28, "head")]
REPEAT = 128


def hwmon_of(device_index):
    import torch
    props = torch.cuda.get_device_properties(device_index)
    want = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
    for card in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(card)).startswith(want):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if hw:
                return hw[0]
    raise RuntimeError(f"no hwmon for PCI {want}")


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def child(seconds):
    import torch
    sys.path.insert(0, ROOT)
    os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
    from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step
    hw = hwmon_of(0)
    eng = HipEngine(device=0)
    if os.environ.get("BD_POWER_PROFILE_FUSION"):          # e.g. "3,11": another launch set
        a, b = os.environ["BD_POWER_PROFILE_FUSION"].split(",")
        eng.set_fusion(int(a), int(b))
    hop, step = hop_samples(0.96), patch_step(0.96)
    x = torch.randn(1024 * hop + 240, device="cuda") * 0.1
    out = torch.empty((1024, 13), device="cuda")
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            samples.append((time.perf_counter(), read_int(os.path.join(hw, "freq1_input")), read_int(os.path.join(hw, "power1_input"))))
            time.sleep(0.01)

    def passes(n):
        for _ in range(n):
            eng.launch([x], hop, step, False, True, out=out)
        torch.cuda.synchronize()

    passes(3)
    t0 = time.perf_counter()
    passes(5)
    per_pass = (time.perf_counter() - t0) / 5
    n = max(10, int(seconds / per_pass))
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    passes(n)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    late = [s for s in samples if s[0] > t0 + 0.5 * (t1 - t0) and s[0] < t1 and s[1] and s[2]]
    cap = read_int(os.path.join(hw, "power1_cap"))
    print(json.dumps({"ms_per_pass": 1e3 * (t1 - t0) / n, "passes": n,
                      "sclk_MHz": sum(s[1] for s in late) / max(1, len(late)) / 1e6,
                      "power_W": sum(s[2] for s in late) / max(1, len(late)) / 1e6,
                      "power_cap_W": cap / 1e6 if cap else None, "samples": len(late)}))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    lib = os.path.join(ROOT, "buzzdetect_amd", "csrc", "libtrace.so")
    if not os.path.exists(lib):
        sys.exit("build the developer library first (libtrace.so, -DBD_KERNEL_TRACE)")
    rows = []
    base = None
    for slot, name in SLOTS:
        env = dict(os.environ, BUZZDETECT_HIP_LIB=lib, BD_REPEAT_SLOT=str(slot), BD_REPEAT_N=str(REPEAT))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seconds)], env=env, capture_output=True,
                           text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{name}: failed\n{r.stderr[-2000:]}", flush=True)
            continue
        d = json.loads(line[-1])
        if slot < 0:
            base = d["ms_per_pass"]
            d["us_per_launch"] = None
        else:
            d["us_per_launch"] = 1e3 * (d["ms_per_pass"] - base) / (REPEAT - 1) if base is not None else None
        d["slot"], d["kernel"] = slot, name
        rows.append(d)
        us = f"{d['us_per_launch']:7.1f} us/launch" if d["us_per_launch"] is not None else f"{1e3 * d['ms_per_pass']:7.1f} us/pass  "
        joule = f"{d['power_W'] * d['us_per_launch'] * 1e-3:6.1f} mJ/launch" if d["us_per_launch"] is not None else ""
        print(f"{name:12s} {us}  sclk {d['sclk_MHz']:6.0f} MHz  power {d['power_W']:6.0f} W (cap {d['power_cap_W']})  {joule}", flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(float(sys.argv[2]))
    else:
        main()
