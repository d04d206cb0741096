#!/usr/bin/env python3
"""Throughput of the analyze hot path on MI355X (BASELINE.json metric: YAMNet windows/s).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the whole hot path (PCM -> log-mel -> YAMNet -> dense head) over FIFTY synthetic 1 h
16 kHz mono recordings already resident in HBM, each fed exactly as BASELINE config 2 / SURVEY 8d say: batches
of 1024 windows (983.04 s = 15 728 640 samples), i.e. 1024 + 1024 + 1024 + 678 = 3750 windows per recording,
187 500 per step, yamnet_k2 embedder at hop 1.0 + model_general_v3 head.  The driver's `--steps 20` is therefore
BASELINE config 4's 1000 x 1 h: at N = 1 about 2.3 s of timed GPU work, at N = 8 125 full rounds.  K steps = 50 K
recordings in total, whatever N is (strong scaling): recording i goes to rank i mod N; what is left when the
count is not a multiple of N is dealt batch by batch; every round of N recordings ends in ONE RCCL gather of the
[3750, 13] logit blocks to rank 0, with sizes known on the host (no count exchange, no host synchronisation
inside the timed region).  `python bench.py --gpus N` starts its own N child ranks (torch.distributed.run) when
it is not already running under one.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# throughput of the architecture on synthetic data: random-init weights in the reference layout ("data" in the JSON line says
# so); the product refuses to run without real weights unless asked (buzzdetect_amd/weights.py)
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")

SAMPLE_RATE = 16000
WINDOWS_PER_BATCH = 1024
HOP_PROP = 1.0
FRAMELENGTH_S = 0.96
FILE_SECONDS = 3600
FILE_SAMPLES = FILE_SECONDS * SAMPLE_RATE                # 57 600 000
FILES_PER_STEP = 50                                      # --steps 20 = config 4's 1000 recordings

# SURVEY §8d / DESIGN.md: algorithmic work per window at hop 1.0
POINTWISE_FLOP_PER_WINDOW = 132_120_576          # 2 * 66 060 288 MAC in the thirteen 1x1 convolutions
CNN_FLOP_PER_WINDOW = 137_289_728
FRONTEND_BYTES_PER_WINDOW = 61_440 + 24_576      # f32 PCM in + f32 log-mel out
STFT_HOP_BYTES = 160 * 4                         # new PCM bytes a frame brings (the 240-sample overlap amortises to 0)
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0                    # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (the 2:1-sparse figure is never used)
PEAK_SPLIT_F16_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0   # ... / 3 MFMAs per f32-accurate product
PEAK_HBM_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E spec
PMC_TRAFFIC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
PMC_TRAFFIC_STRICT_FILE = os.path.join("profiles", "r06_pmc_traffic_strict_f32.json")
# What the board's 1400 W limit leaves of the paper peak: back-to-back v_mfma_f32_32x32x16_f16 on every SIMD, constant operands,
# held for 3 s, settles at 1987 TFLOP/s (2.0 GHz, 1345 W) - tools/ubench_power.hip, profiles/r03_ubench_power.txt.  The headline
# loop itself runs AT the limit (the `power` object of the line), so this is the ceiling its matrix work is priced against in
# `frac_of_sustained`; `frac` stays relative to the paper peak.
SUSTAINED_F16_MFMA_TFLOPS = 1987.0
PMC_FRONTEND_FILE = os.path.join("profiles", "r03_pmc_frontend_alone.csv")

# per-window HBM bytes each depthwise / conv1 launch must move (read input + write output, f32 NHWC)
_DEF = ((2, 32), (1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512), (1, 512),
        (1, 512), (1, 512), (2, 1024), (1, 1024))


def slot_plan(launches, stem_kernel="stem_reg_kernel", tail=True):
    """Map the 29 profile slots to (slot name, kernel family, per-window algorithmic bytes, per-window flops)
    for the launches that actually happened (fused kernels are timed in the pointwise slot of their layer).  Launch sets the engine
    has: the default (slots 0, 5, 7, 13, 23, 25, 27, 28), bd_set_fusion separable = 10 (+ 9, 11: layers 5-7 on their own kernels),
    one kernel per op (every slot); `tail` = layers 13 / 14 run on septail.hip's kernel (always, behind the on-chip run)."""
    plan = {0: ("frontend", "logmel_kernel", FRONTEND_BYTES_PER_WINDOW, 0)}
    h, w, c = 48, 32, 32
    conv1 = (96 * 64 * 4 + h * w * c * 4, 2 * 9 * h * w * c)
    if launches[1] > 0:
        plan[1] = ("conv1", "conv1_kernel", conv1[0], conv1[1])
    run = None            # layers 8-11 as one launch: [first layer, bytes, flops] of the layers that had no launch of their own
    # pointwise 5 -> layer 6 -> depthwise 7 -> pointwise 7 as one launch (sepmid.hip): nothing in slots 9 and 11, all of it in 13
    mid_mode = launches[7] > 0 and launches[8] == 0 and launches[9] == 0 and launches[10] == 0 and launches[11] == 0 and \
        launches[12] == 0 and launches[13] > 0
    mid = None            # [bytes in, flops] while the layers of that launch are walked
    for layer, (stride, cout) in enumerate(_DEF[1:], start=2):
        ho, wo = h // stride, w // stride
        dw_slot, pw_slot = 2 * layer - 2, 2 * layer - 1
        dw = ((h * w * c + ho * wo * c) * 4, 2 * 9 * ho * wo * c)
        pw = ((ho * wo * c + ho * wo * cout) * 4, 2 * ho * wo * c * cout)
        if tail and layer in (13, 14) and launches[dw_slot] == 0 and launches[pw_slot] > 0:
            # septail.hip: pointwise 13 with depthwise 14 in its epilogue (f16 planes in, f16 planes out: 4 bytes per element), pointwise
            # 14 with the average pool; depthwise 13 has run in the on-chip launch in front
            if layer == 13:
                nm, fam, nb, fl = plan[pw_slot - 2]
                plan[pw_slot - 2] = (nm + "+dw13", fam, nb - h * w * c * 4 + ho * wo * c * 4, fl + dw[1])
                plan[pw_slot] = ("pw13+dw14", "tail_gemm_kernel", (ho * wo * c + ho * wo * cout) * 4, pw[1] + 2 * 9 * ho * wo * cout)
            else:
                plan[pw_slot] = ("pw14+pool", "tail_gemm_kernel", (ho * wo * c + cout) * 4, pw[1] + ho * wo * cout)
            h, w, c = ho, wo, cout
            continue
        if mid_mode and layer in (5, 6, 7):
            if layer == 5:        # its stride-2 depthwise ran in the layer-4 kernel's epilogue (as below), its 1x1 opens the launch
                nm, fam, nb, fl = plan[7]
                plan[7] = (nm + "+dw5", "l4_window_kernel", nb - h * w * c * 4 + ho * wo * c * 4, fl + dw[1])
                mid = [ho * wo * c * 4, pw[1]]
            else:
                mid[1] += dw[1] + pw[1]
            if layer == 7:
                plan[13] = ("pw5-pw7", "sep_mid_kernel", mid[0] + ho * wo * cout * 4, mid[1])
            h, w, c = ho, wo, cout
            continue
        if stride == 1 and c == 512 and cout == 512 and launches[dw_slot] == 0 and launches[pw_slot] == 0 and launches[0] > 0:
            # timed (and launched) with the last layer of its run (sepchip.hip keeps the tiles between the layers on the CU)
            run = run or [layer, 0, 0]
            run[1] += (h * w * c + ho * wo * cout) * 4
            run[2] += dw[1] + pw[1]
            h, w, c = ho, wo, cout
            continue
        if layer == 2:
            stem_flops = dw[1] + pw[1]
        if launches[dw_slot] > 0:
            if layer == 3 and launches[1] == 0 and launches[3] == 0:
                # stem3: conv1 + layer 2 + depthwise 3 in one kernel: log-mel patch in, depthwise-3 output out
                plan[dw_slot] = ("stem3(1-3dw)", "stem3_kernel", 96 * 64 * 4 + ho * wo * c * 4,
                                 conv1[1] + stem_flops + dw[1])
            else:
                plan[dw_slot] = (f"dw{layer}", "depthwise_kernel", dw[0], dw[1])
        if layer == 3 and launches[pw_slot] > 0 and launches[dw_slot] == 0 and launches[1] == 0 and launches[3] == 0:
            # layers 1-3 in one kernel (stem3_kernel<true>): log-mel patch in, layer-3 output out
            plan[pw_slot] = ("stem(1-3)", stem_kernel, 96 * 64 * 4 + ho * wo * cout * 4,
                             conv1[1] + stem_flops + dw[1] + pw[1])
        elif launches[pw_slot] > 0:
            if launches[dw_slot] > 0 or (stride == 2 and layer >= 3):
                # (a stride-2 layer without a depthwise launch: the previous kernel applied it)
                # (engine rule, cnn.hip launch_pointwise_ws: K >= 128 and N % 256 == 0 run on the wave-specialised
                #  kernel with pass-through producers; the others on the plain split-f16 GEMM kernel)
                fam = "sep_ws_kernel" if (c >= 128 and c % 64 == 0 and cout % 256 == 0) else "pointwise_f16x3_kernel"
                if fam == "sep_ws_kernel" and c in (128, 256):
                    fam = "pw_res_kernel"            # layers 5 and 7: the weights live in registers
                plan[pw_slot] = (f"pw{layer}", fam, pw[0], pw[1])
                if launches[dw_slot] == 0 and layer >= 5 and (pw_slot - 2) in plan:
                    nm, fam, nb, fl = plan[pw_slot - 2]
                    # the fused kernel of the previous layer wrote this layer's depthwise output instead of its own
                    # (layer 4 + depthwise 5 have their own kernel: a window per workgroup, no overlapping bands; layer 6 + depthwise 7:
                    #  the 8-wave kernel; the on-chip run takes layer 12 and depthwise 13 along: sepchip.hip, NDW)
                    prev_fam = "l4_window_kernel" if layer == 5 else fam if fam == "sep_chip_kernel" else "sep_ws_kernel"
                    plan[pw_slot - 2] = (nm + f"+dw{layer}", prev_fam, nb - h * w * c * 4 + ho * wo * c * 4, fl + dw[1])
            else:                 # depthwise inside the product kernel: layer input in, layer output out
                if run and c == 512 and cout == 512:
                    # sepchip.hip: the tiles between the run's layers stay on the CU - the run's input in, its output out
                    plan[pw_slot] = (f"sep{run[0]}-{layer}", "sep_chip_kernel", (h * w * c + ho * wo * cout) * 4, run[2] + dw[1] + pw[1])
                    run = None
                else:             # layer 4 (its own kernel, named when layer 5 is walked) and layer 6 on the 8-wave kernel
                    plan[pw_slot] = (f"sep{layer}", "sep_ws_kernel", (h * w * c + ho * wo * cout) * 4, dw[1] + pw[1])
        h, w, c = ho, wo, cout
    if 27 in plan and plan[27][0] == "pw14+pool":
        plan[28] = ("head", "pool_head_kernel", (1024 + 13) * 4, 2 * 1024 * 13)
    else:
        plan[28] = ("pool_head", "pool_head_kernel", (6 * 1024 + 13) * 4, 6 * 1024 + 2 * 1024 * 13)
    return plan


def slot_plan_f32(launches, stem_kernel="stem_reg_f32_kernel"):
    """The exact-f32 mode's default launch set (bd_set_pointwise_mode 0, bd_set_fusion 3 / 1): slot -> (name, kernel, per-window
    algorithmic bytes, per-window flops).  Layers 1-3 are stem3_f32_kernel (slot 5), layer 4 + depthwise 5 l4_f32_kernel
    (slot 7); from layer 5 on every 1x1 convolution is pointwise_kernel with the NEXT layer's depthwise in its epilogue
    (timed in its own pointwise slot: depthwise-L output in, depthwise-(L+1) output out), layer 14 with the average pool; layers
    8-12 + depthwise 13 are ONE launch (sep_chip_f32_kernel, layer 12's slot) and pointwise 5 + layers 6-7 another
    (sep_mid_f32_kernel, layer 7's slot) when the on-chip runs are on (the default)."""
    plan = {0: ("frontend", "logmel_kernel", FRONTEND_BYTES_PER_WINDOW, 0)}
    dims = []                      # per layer 2..14: (h_in, w_in, c_in, h_out, w_out, c_out)
    h, w, c = 48, 32, 32
    for stride, cout in _DEF[1:]:
        dims.append((h, w, c, h // stride, w // stride, cout))
        h, w, c = h // stride, w // stride, cout
    dw_fl = lambda d: 2 * 9 * d[3] * d[4] * d[2]              # noqa: E731
    pw_fl = lambda d: 2 * d[3] * d[4] * d[2] * d[5]           # noqa: E731
    conv1_fl = 2 * 9 * 48 * 32 * 32
    if launches[5] > 0 and launches[1] == 0:
        d2, d3 = dims[0], dims[1]
        plan[5] = ("stem(1-3)", stem_kernel, 96 * 64 * 4 + d3[3] * d3[4] * d3[5] * 4,
                   conv1_fl + dw_fl(d2) + pw_fl(d2) + dw_fl(d3) + pw_fl(d3))
    if launches[7] > 0 and launches[6] == 0:
        d4, d5 = dims[2], dims[3]
        plan[7] = ("sep4+dw5", "l4_reg_f32_kernel" if stem_kernel == "stem_reg_f32_kernel" else "l4_f32_kernel", (d4[0] * d4[1] * d4[2] + d5[3] * d5[4] * d5[2]) * 4, dw_fl(d4) + pw_fl(d4) + dw_fl(d5))
    # layers 8-12 + depthwise 13 as one launch (sepchipf32.hip), timed in layer 12's slot: depthwise-8 output in, depthwise-13 out
    # (behind the middle run: layer-7 output in, the run applies depthwise 8 itself)
    chip = launches[23] > 0 and all(launches[s_] == 0 for s_ in (14, 15, 16, 17, 18, 19, 20, 21, 22))
    # pointwise 5 + layer 6 + layer 7 as one launch (sepmidf32.hip), timed in layer 7's slot: depthwise-5 output in, layer-7 output out
    mid = launches[13] > 0 and all(launches[s_] == 0 for s_ in (8, 9, 10, 11, 12))
    for layer in range(5, 15):
        d = dims[layer - 2]
        slot = 2 * layer - 1
        if launches[slot] == 0 or launches[slot - 1] > 0:
            continue
        rows_in = d[3] * d[4] * d[2] * 4
        if chip and layer == 12:
            d8, n13 = dims[6], dims[11]
            first_dw = 8 if mid else 9                             # behind the middle run depthwise 8 is this launch's work
            plan[slot] = ("sep8-12+dw13", "sep_chip_f32_kernel",
                          (d8[0] * d8[1] if mid else d8[3] * d8[4]) * d8[2] * 4 + n13[3] * n13[4] * n13[2] * 4,
                          sum(pw_fl(dims[L_ - 2]) for L_ in range(8, 13)) + sum(dw_fl(dims[L_ - 2]) for L_ in range(first_dw, 14)))
            continue
        if mid and layer == 7:
            d5 = dims[3]
            plan[slot] = ("pw5-pw7", "sep_mid_f32_kernel", d5[3] * d5[4] * d5[2] * 4 + d[3] * d[4] * d[5] * 4,
                          pw_fl(dims[3]) + dw_fl(dims[4]) + pw_fl(dims[4]) + dw_fl(dims[5]) + pw_fl(dims[5]))
            continue
        if layer == 14:
            plan[slot] = ("pw14+pool", "tail_gemm_f32_kernel" if chip else "pointwise_kernel", rows_in + d[5] * 4, pw_fl(d) + d[3] * d[4] * d[5])
        else:
            n = dims[layer - 1]
            # (layer 13 behind the on-chip run: septail.hip's kernel, like layer 14 above)
            plan[slot] = (f"pw{layer}+dw{layer + 1}", "tail_gemm_f32_kernel" if chip and layer == 13 else "pointwise_kernel",
                          rows_in + n[3] * n[4] * n[2] * 4, pw_fl(d) + dw_fl(n))
    plan[28] = ("head", "pool_head_kernel", (1024 + 13) * 4, 2 * 1024 * 13)
    return {k: v for k, v in plan.items() if launches[k] > 0}


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------------------- launch
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class PowerWatch:
    """Board power and shader clock of one GPU from its hwmon files (sysfs, no HIP call), polled every 10 ms by a thread."""

    def __init__(self, torch, dev_index: int):
        import glob
        self.dir = None
        self.samples = []
        self._stop = None
        try:
            props = torch.cuda.get_device_properties(dev_index)
            want = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
            for card in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(card)).startswith(want):
                    hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
        except (AttributeError, OSError):
            pass

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        if self.dir:
            self._stop = threading.Event()

            def poll():
                while not self._stop.is_set():
                    self.samples.append((time.perf_counter(), self._read("freq1_input"), self._read("power1_input")))
                    time.sleep(0.01)

            self._thread = threading.Thread(target=poll, daemon=True)
            self._t0 = time.perf_counter()
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._stop is not None:
            self._t1 = time.perf_counter()
            self._stop.set()
            self._thread.join()
        return False

    def summary(self):
        """Averages over the second half of the watched region (the sensor lags the load by ~0.3 s)."""
        if not self.dir or not self.samples:
            return None
        mid = 0.5 * (self._t0 + self._t1)
        late = [s for s in self.samples if s[0] >= mid and s[1] and s[2]]
        if not late:
            return None
        cap = self._read("power1_cap")
        watts = sum(s[2] for s in late) / len(late) / 1e6
        return {"avg_W": round(watts, 1), "cap_W": round(cap / 1e6, 1) if cap else None,
                "frac_of_cap": round(watts / (cap / 1e6), 3) if cap else None,
                "sclk_MHz_avg": round(sum(s[1] for s in late) / len(late) / 1e6, 1), "sclk_MHz_max": 2400, "samples": len(late),
                "source": "hwmon power1_input / freq1_input of this GPU, 10 ms polls, second half of the timed region",
                "what": "the board's power management holds the shader clock below its 2.4 GHz maximum for the whole timed "
                        "region (at the 1400 W limit on most boxes of the pool, 60-100 W below it on some): the hot path is "
                        "power-bound (DESIGN.md 7b; per-kernel figures in profiles/r03_power_profile.txt, what the limit "
                        "leaves of the paper peaks in profiles/r03_ubench_power.txt)"}


def visible_gpus():
    """GPUs this process may use, WITHOUT any HIP call: KFD topology nodes that have SIMDs (CPU nodes report simd_count 0) AND
    whose DRM render node this process can open - a container that leases one GPU of an eight-GPU host sees all eight in the
    topology but only its own /dev/dri/renderD* (round 5: `--gpus 8` started eight ranks on a one-GPU box) - cut down by a
    *_VISIBLE_DEVICES list if one is set.  None when the topology cannot be read."""
    import glob
    count = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for path in files:
        try:
            simds, minor = 0, None
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        simds = int(line.split()[1])
                    elif line.startswith("drm_render_minor"):
                        minor = int(line.split()[1])
            if simds <= 0:
                continue
            node = f"/dev/dri/renderD{minor}" if minor is not None and minor > 0 else None
            if node is None or not os.path.exists("/dev/dri") or os.access(node, os.R_OK | os.W_OK):
                count += 1                   # (no render-node information: counted, as before)
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            count = min(count, len([x for x in os.environ[var].split(",") if x.strip()]))
    return count


def runtime_gpu_count():
    """The HIP runtime's device count, asked of a short-lived CHILD process: the caller (self_launch's parent) must not open a
    GPU before its ranks exist.  None when the child cannot say."""
    code = "import torch; print(torch.cuda.device_count())"
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` from a plain shell: start N fresh child ranks BEFORE this process touches a
    GPU (no HIP call and no torch import here), let rank 0's JSON line through on stdout, hand back the exit code."""
    rehearsal = os.environ.get("BD_BENCH_REHEARSAL") == "1"
    visible = visible_gpus()
    if (visible is None or visible >= n) and not rehearsal:
        counted = runtime_gpu_count()           # second opinion: the runtime's own count, from a short-lived child
        visible = counted if counted is not None else visible
    if visible is not None and visible < n and not rehearsal:
        log(f"--gpus {n} but only {visible} GPU(s) visible (BD_BENCH_REHEARSAL=1 shares GPU 0 over gloo: control flow only)")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------------------- workload
def synthetic_audio(device, n_samples: int, seed: int):
    """0.1*N(0,1) noise + 0.3*sin(2*pi*220 t) bursts of 0.5 s every 5 s, clipped to [-1,1)."""
    import numpy as np
    import torch
    gen = torch.Generator(device=device).manual_seed(seed)
    x = 0.1 * torch.randn(n_samples, generator=gen, device=device, dtype=torch.float32)
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / SAMPLE_RATE
    burst = (torch.remainder(t, 5.0) < 0.5).to(torch.float32)
    x += 0.3 * torch.sin(2 * np.pi * 220.0 * t) * burst
    return x.clamp_(-1.0, 1.0 - 2.0 ** -23)


def file_batches(hop: int):
    """(first sample, samples, windows) of one recording's batches: chunk edges on multiples of 1024 hops."""
    out, at = [], 0
    while at < FILE_SAMPLES:
        n = min(WINDOWS_PER_BATCH * hop, FILE_SAMPLES - at)
        out.append((at, n))
        at += n
    return out


def cpu_baseline(engine, hop: int, step: int, windows: int):
    """The oracle timed on this box's host cores (checker + reported baseline, never the product).
    The sample is cut into the workload's own 1024-window chunks on both sides."""
    import numpy as np
    import torch
    from buzzdetect_amd import weights as W
    from oracle import yamnet_oracle as O
    from oracle.torch_baseline import TorchYamnet, usable_cores
    threads = usable_cores()
    torch.set_num_threads(threads)
    chunks = max(1, windows // WINDOWS_PER_BATCH)
    log(f"cpu_baseline: torch-CPU restatement on {chunks} x {WINDOWS_PER_BATCH} windows, {threads} threads")
    head = W.load_head()
    blob, mel = W.synthetic_embedder_blob(), W.load_mel("yamnet_k2")
    model = TorchYamnet(blob, mel, head.kernel, head.bias)
    waves = [O.synthetic_audio(hop * WINDOWS_PER_BATCH, seed=4321 + i) for i in range(chunks)]
    model.predict(waves[0][: hop * 64], hop, step)                      # warm-up
    passes = []
    cpu_logits = None
    for rep in range(5):
        t0 = time.perf_counter()
        outs = [model.predict(w, hop, step) for w in waves]
        passes.append(time.perf_counter() - t0)
        cpu_logits = outs
        log(f"cpu_baseline: pass {rep}: {chunks * WINDOWS_PER_BATCH / passes[-1]:.1f} windows/s")
    med = sorted(passes)[2]
    n_win = sum(o.shape[0] for o in cpu_logits)
    gpu_logits = [engine.predict(w, FRAMELENGTH_S * HOP_PROP).numpy() for w in waves]
    d32 = max(float(np.abs(g - c).max()) for g, c in zip(gpu_logits, cpu_logits))
    # f64 oracle on the first 8 windows: 15600 + 7*hop samples is exactly 8 windows with no zero padding,
    # so these rows are the same function of the audio as rows 0..7 of the long chunk
    ref64 = O.predict(waves[0][: 15600 + 7 * hop], blob, mel, head.kernel, head.bias, hop, step, np.float64)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(n_win / med, 2), "unit": "windows/s", "cores": threads, "kind": "port",
        "implementation": "CPU restatement (torch-CPU fp32), not TensorFlow",
        "sample": f"{chunks} chunks x {WINDOWS_PER_BATCH} windows ({n_win * 0.96:.0f} s of audio), "
                  f"1 warm-up + median of 5 passes",
        "seconds_per_pass": round(med, 3), "cpu_model": cpu_model,
        "max_abs_dlogit_gpu_vs_cpu_f32": d32,
        "max_abs_dlogit_gpu_vs_cpu_f64_first8": float(np.abs(gpu_logits[0][:8] - ref64).max()),
    }


# --------------------------------------------------------------------------------------------------- extra legs
def h2d_leg(engine, streams, device, hop: int, framehop_s: float, batches: int, s16: bool):
    """Config-2 batches that start on the HOST: pinned buffers -> async copy on a copy stream -> (s16: device-side
    conversion, bd_resample_s16 at 16 kHz -> 16 kHz) -> predict.  Copies are inside the timed region.
    A device buffer of the ring is reused once the HOST has seen its batch finish (event.synchronize(): the host runs
    four batches ahead of the GPU at most).  Making the copy stream wait for the compute stream's event on the device
    instead costs 15 % (tools/h2d_overlap_probe.py: 1.58 -> 1.26-1.37 M windows/s with nothing else changed): a copy
    that depends on a compute signal does not run at the copy engine's full rate."""
    import torch
    n = WINDOWS_PER_BATCH * hop
    dt = torch.int16 if s16 else torch.float32
    host = []
    for i in range(2):
        x = synthetic_audio(device, n, 777 + i)
        if s16:
            x = (x * 32768.0).round().clamp_(-32768, 32767).to(torch.int16)
        host.append(x.cpu().pin_memory())
    copy_stream = torch.cuda.Stream(device)
    ring = [torch.empty(n, dtype=dt, device=device) for _ in range(4)]
    copied = [torch.cuda.Event() for _ in ring]
    consumed = [None] * len(ring)
    spare = [torch.cuda.Event() for _ in ring]       # events are recorded again and again, never created per batch: a
                                                     # stream of fresh events makes the HIP runtime grow its signal pool

    def run(count):
        for i in range(count):
            slot = i % len(ring)
            if consumed[slot] is not None:
                consumed[slot].synchronize()
            with torch.cuda.stream(copy_stream):
                ring[slot].copy_(host[i % 2], non_blocking=True)
                copied[slot].record(copy_stream)
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                s.wait_event(copied[slot])
                pcm = engine[i % len(engine)].resample(ring[slot], SAMPLE_RATE, SAMPLE_RATE) if s16 else ring[slot]
                engine[i % len(engine)].predict(pcm, framehop_s)
                spare[slot].record(s)
                consumed[slot] = spare[slot]

    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(batches)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    rate = batches * WINDOWS_PER_BATCH / sec
    return round(rate, 1), round(batches * n * (2 if s16 else 4) / sec / 1e9, 2)


def analyze_leg(device_index: int, hours: int, chunklength: float, framehop_prop: float, engines=None, files: int = 1):
    """analyze() files -> CSV over `files` generated 16-bit mono WAVs of `hours` h each on tmpfs, in ONE call; audio-seconds
    per second (the reference's own rate definition, src/inference/worker.py:54-62, over the whole run) and the wall
    seconds each stage of the feeder spent working (Report.busy)."""
    import shutil
    import tempfile
    import wave
    import numpy as np
    import torch
    from buzzdetect_amd.analyze import analyze
    root = tempfile.mkdtemp(prefix="bd_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        audio, out = os.path.join(root, "audio"), os.path.join(root, "out")
        os.makedirs(audio)
        hour = synthetic_audio(torch.device("cuda", device_index), FILE_SAMPLES, 4242)
        block = (hour * 32768.0).round().clamp_(-32768, 32767).to(torch.int16).cpu().numpy().astype("<i2").tobytes()
        for i in range(files):
            with wave.open(os.path.join(audio, f"synthetic_{hours}h_{i:03d}.wav"), "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(SAMPLE_RATE)
                for _ in range(hours):
                    w.writeframes(block)
        del block, hour
        # read every file once before the clock starts: the FIRST pass over tmpfs pages that have just been written runs at
        # a quarter of the rate of any later one (12 vs 50 GB/s on the GPU box whatever the destination buffer -
        # tools/read_probe.py) - an artefact of generating the recordings a moment ago, not a cost of analyze()
        for name in sorted(os.listdir(audio)):
            with open(os.path.join(audio, name), "rb", buffering=0) as f:
                buf = bytearray(16 << 20)
                while f.readinto(buf):
                    pass
        secs, busy = [], []
        for call in range(2):      # the first call pins its staging buffers (host allocator cold), the second is the sustained rate
            t0 = time.perf_counter()
            rep = analyze("model_general_v3", classes_out="all", framehop_prop=framehop_prop, chunklength=chunklength,
                          dir_audio=audio, dir_out=f"{out}{call}", embeddername="yamnet_k2", engines=engines, rank=0, world_size=1)
            secs.append(time.perf_counter() - t0)
            busy.append({k: round(v, 4) for k, v in sorted(getattr(rep, "busy", {}).items())})
            assert rep.files_done == files, rep
            shutil.rmtree(f"{out}{call}", ignore_errors=True)
        sec = secs[0]
        return {"files": files, "hours_each": hours, "chunklength_s": chunklength, "framehop_prop": framehop_prop,
                "audio_s_per_s": round(rep.audio_seconds / sec, 1), "windows_per_s": round(rep.windows / sec, 1),
                "seconds": round(sec, 3), "windows": rep.windows, "chunks": rep.chunks,
                "pcm_GBps_s16": round(rep.audio_seconds * SAMPLE_RATE * 2 / sec / 1e9, 3), "stage_busy_s": busy[0],
                "second_call": {"audio_s_per_s": round(rep.audio_seconds / secs[1], 1),
                                "windows_per_s": round(rep.windows / secs[1], 1), "seconds": round(secs[1], 3),
                                "pcm_GBps_s16": round(rep.audio_seconds * SAMPLE_RATE * 2 / secs[1] / 1e9, 3),
                                "stage_busy_s": busy[1]}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


# --------------------------------------------------------------------------------------------------- main
def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps of 50 one-hour recordings each")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-windows", type=int, default=4096,
                    help="size of the bounded CPU-baseline sample (about 15-25 s of CPU work over 6 passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the host-resident (H2D-inclusive), analyze() and arithmetic-mode legs")
    ap.add_argument("--per-slot", action="store_true", help="print per-kernel times to stderr")
    ap.add_argument("--streams", type=int, default=2,
                    help="analyzer streams per GPU: batches are dealt round-robin, in issue order, to this many engines, each "
                         "on its own HIP stream (the reference's analyzers_gpu knob, src/analyze.py:218-253).  Two since round 5: "
                         "the on-chip kernels of layers 5-12 hold a whole CU each, so a third stream only adds queueing (same box, "
                         "alternating: 1.852 / 1.855 / 1.849 M windows/s with two, 1.835 / 1.826 / 1.832 M with three, 1.75 M with "
                         "four, 1.57 M with one; DESIGN.md 7)")
    ap.add_argument("--sep-variant", type=int, default=None, help="tuning: bd_set_fusion separable code (1 = default, 10 = layers 5-7 on four kernels)")
    ap.add_argument("--pointwise-mode", choices=["f16x3", "f32", "f16"], default=None,
                    help="profiling only: run the WHOLE bench in this arithmetic mode (the line then carries mode_override; "
                         "the driver's headline never uses it)")
    ap.add_argument("--stem", type=int, default=None,
                    help="tuning: bd_set_fusion stem code (3 = default: layer-2 tile handed over in registers, 5 = block stem)")
    ap.add_argument("--pw-variant", type=int, default=None, help="tuning: kernel variant of the plain 1x1 convolutions (layers 5-14)")
    ap.add_argument("--group-windows", type=int, default=0, help="windows per CNN pass (0 = library default)")
    ap.add_argument("--files-per-step", type=int, default=FILES_PER_STEP,
                    help="recordings per step (50: --steps 20 is config 4's 1000 x 1 h; profiling runs use fewer)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)

    import numpy as np
    import torch
    from buzzdetect_amd import sharding
    from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

    files_per_step = max(1, args.files_per_step)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        return 2
    if not torch.cuda.is_available():
        log("bench.py needs an MI355X; there is no CPU path for the product")
        return 2
    # BD_BENCH_REHEARSAL=1: every rank uses GPU 0 and the collectives run over gloo on host copies - only to
    # exercise the multi-rank control flow on a one-GPU box; never a measurement.
    rehearsal = os.environ.get("BD_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=dev_index)
               for _ in range(max(1, args.streams))]
    # every analyzer gets a stream of its own (same-box A/B against putting the first one on the default stream: no difference)
    streams = [torch.cuda.Stream(device) for _ in engines]
    comm_stream = torch.cuda.Stream(device)
    engine = engines[0]
    for e in engines:
        if args.group_windows:
            e.set_group_windows(args.group_windows)
        if args.sep_variant is not None or args.stem is not None:
            e.set_fusion(True if args.stem is None else args.stem, True if args.sep_variant is None else args.sep_variant)
        if args.pointwise_mode is not None:
            e.set_pointwise_mode(args.pointwise_mode)
        if args.pw_variant is not None:
            for layer in range(5, 15):
                e.set_pointwise_variant(layer, args.pw_variant)
    framehop_s = FRAMELENGTH_S * HOP_PROP
    hop, step = hop_samples(framehop_s), patch_step(framehop_s)
    batches = file_batches(hop)
    batch_windows = [engine.num_windows(n, hop, step) for _, n in batches]
    windows_per_file = sum(batch_windows)
    assert batch_windows == [1024, 1024, 1024, 678] and windows_per_file == 3750, batch_windows
    n_classes = engine.n_classes

    # a few distinct recordings so that a step never re-reads the PCM it has just processed
    files = [synthetic_audio(device, FILE_SAMPLES, 1234 + i) for i in range(3)]
    torch.cuda.synchronize()

    RING = 4
    blocks = {}            # rows -> ring of [rows, C] round buffers (+ gather targets on rank 0)

    def buffers(rows: int):
        if rows not in blocks:
            # (+ 4 spare rows behind the block: the first carries this rank's (rank + 1, windows) for `ranks_seen`)
            local = [torch.zeros((rows + sharding.ROW_ALIGN, n_classes), dtype=torch.float32, device=device) for _ in range(RING)]
            gath = None
            if world > 1 and rank == 0:
                gath = [torch.empty((world, rows + sharding.ROW_ALIGN, n_classes), dtype=torch.float32,
                                    device="cpu" if rehearsal else device) for _ in range(RING)]
            blocks[rows] = (local, gath, [None] * RING)
        return blocks[rows]

    issued = [0]
    dealt = [0]
    read_back = {"batches": 0, "rows": 0, "nonfinite": 0}
    # N > 1: what the gathers delivered, counted on rank 0 FROM the gathered blocks: every rank writes (rank + 1, windows of
    # the round) into the first spare row behind its block; rank 0 adds them up on the communication stream
    seen = torch.zeros((world, 2), dtype=torch.float64, device="cpu" if rehearsal else device) if world > 1 and rank == 0 else None

    def run_files(n_steps: int, use_streams: bool = True):
        """n_steps x 50 recordings through this rank's share of the rounds (see the module docstring).  Once per step (50
        recordings) one batch's result is read back on the host through DeviceResult.numpy() - the path the reference's
        writer takes (src/write/worker.py:69): the range verdict of its launch set is waited for and looked at, the rows
        cross PCIe - so the verdict path is inside the timed region."""
        rounds = sharding.plan_rounds(n_steps * files_per_step, batch_windows, world)
        per_step = max(1, len(rounds) // max(n_steps, 1))
        for ri, rnd in enumerate(rounds):
            local, gath, reusable = buffers(rnd.rows)
            slot = issued[0] % RING
            issued[0] += 1
            mine = rnd.units[rank]
            used = set()
            probe = None
            for u, at in zip(mine, sharding.unit_offsets(mine)):
                j = (dealt[0] % len(engines)) if use_streams else 0       # batches dealt round-robin in issue order
                dealt[0] += 1
                s = streams[j]
                if j not in used and reusable[slot] is not None:
                    s.wait_event(reusable[slot])           # the gather that last read this block is done
                used.add(j)
                first, n = batches[u.batch]
                with torch.cuda.stream(s):
                    res = engines[j].predict(files[u.file % len(files)][first:first + n], framehop_s,
                                             out=local[slot][at:at + u.windows])
                if probe is None:
                    probe = res
            if world > 1:
                # the marker row is written in EVERY round, also by a rank that has no units in it (windows = 0): a reused
                # ring slot would otherwise hand rank 0 the marker of the slot's previous round a second time
                mark_stream = streams[next(iter(used))] if used else streams[0]
                if not used and reusable[slot] is not None:
                    mark_stream.wait_event(reusable[slot])
                used.add(streams.index(mark_stream))
                with torch.cuda.stream(mark_stream):
                    local[slot][rnd.rows, 0] = float(rank + 1)
                    local[slot][rnd.rows, 1] = float(sum(u.windows for u in mine))
            if probe is not None and ri % per_step == per_step - 1:
                rows = probe.numpy()                       # waits for that launch set's verdict, repeats in f32 if raised
                read_back["batches"] += 1
                read_back["rows"] += int(rows.shape[0])
                read_back["nonfinite"] += int((~np.isfinite(rows)).sum())
            if world > 1:         # ONE gather per round, after both analyzer streams, off their critical path
                for j in used:
                    comm_stream.wait_stream(streams[j])
                with torch.cuda.stream(comm_stream):
                    nrow = rnd.rows + sharding.ROW_ALIGN
                    if rehearsal:
                        sharding.gather_round(local[slot].cpu(), nrow, dst=0, out=gath[slot] if gath else None)
                    else:
                        sharding.gather_round(local[slot], nrow, dst=0, out=gath[slot] if gath else None)
                    if seen is not None:                   # per sending rank: rounds it took part in, windows it delivered
                        mark = gath[slot][:, rnd.rows, :2].to(torch.float64)
                        ok = (mark[:, 0] == torch.arange(1, world + 1, dtype=torch.float64, device=mark.device))
                        seen[:, 0] += ok.to(torch.float64)
                        seen[:, 1] += mark[:, 1] * ok.to(torch.float64)
                    ev = torch.cuda.Event()
                    ev.record(comm_stream)
                    reusable[slot] = ev

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps: int, use_streams: bool = True) -> float:
        fence()
        t0 = time.perf_counter()
        run_files(steps, use_streams)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    if rank == 0:
        log(f"warm-up {args.warmup} steps, then {args.steps} timed steps ({files_per_step} x 1 h recordings each) over {world} GPU(s)")
    run_files(max(args.warmup, 0))
    fence()

    # region 1: the number reported as `value` — nothing but the hot path (and, for N > 1, the gathers)
    for e in engines:
        e.profile_enable(False)
    read_back.update(batches=0, rows=0, nonfinite=0)
    reruns0 = sum(e.overflow_reruns for e in engines)
    if world > 1:
        fence()                    # (every rank: fence() holds a collective) the warm-up's gathers are done ...
        if seen is not None:
            seen.zero_()           # ... so the count below is the timed region's alone
    with PowerWatch(torch, dev_index) as watch:
        elapsed = timed_region(args.steps)
    timed_read_back = dict(read_back, exact_f32_repeats=sum(e.overflow_reruns for e in engines) - reruns0)
    ranks_seen = None
    if seen is not None:
        ranks_seen = [{"rank": r, "rounds": int(seen[r, 0].item()), "windows": int(seen[r, 1].item())} for r in range(world)]
    windows_per_step = windows_per_file * files_per_step
    total_windows = windows_per_step * args.steps
    value = total_windows / elapsed
    if rank == 0:
        log(f"{value:.0f} windows/s ({1e3 * elapsed / args.steps:.3f} ms/step of {windows_per_step} windows, "
            f"timed region {elapsed:.3f} s)")

    # region 2 (rank 0's own share, one stream): the same batches with every kernel bracketed by HIP events on
    # its stream (costs a few % of wall time, which is why `value` does not come from this region)
    events_on = not args.no_kernel_events
    ms = launches = None
    elapsed_events = None
    ev_steps = 10                    # recordings in the event region
    if events_on:
        engine.profile_read()                                # drop anything recorded so far
        engine.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for i in range(ev_steps):
                for first, n in batches:
                    engine.predict(files[i % len(files)][first:first + n], framehop_s)
        torch.cuda.synchronize()
        elapsed_events = time.perf_counter() - t0
        engine.profile_enable(False)
        ms, launches = engine.profile_read()

    # region 3 (single GPU): `value`'s workload once more with every product an exact f32 product - the reference's own
    # precision.  The SAME run_files() loop (50 distinct recordings per step as 1024/1024/1024/678-window batches, three
    # analyzer streams, one read-back per step), then the same kernel-event region, in the exact-f32 mode.
    strict = None
    if world == 1 and not args.no_extras and "f32" in getattr(engine, "POINTWISE_MODES", ()):
        strict_steps = max(5, min(args.steps, 8))
        for e in engines:
            e.set_pointwise_mode("f32")
        try:
            run_files(1)
            read_back.update(batches=0, rows=0, nonfinite=0)
            s_elapsed = timed_region(strict_steps)
            strict = {"value": round(windows_per_step * strict_steps / s_elapsed, 1), "steps": strict_steps,
                      "ms_per_step": round(1e3 * s_elapsed / strict_steps, 4), "read_back": dict(read_back)}
            log(f"exact-f32 mode, same loop: {strict['value']:.0f} windows/s ({strict['ms_per_step']:.3f} ms/step, {strict_steps} steps)")
            if events_on:
                engine.profile_read()
                engine.profile_enable(True)
                torch.cuda.synchronize()
                with torch.cuda.stream(streams[0]):
                    for i in range(ev_steps):
                        for first, n in batches:
                            engine.predict(files[i % len(files)][first:first + n], framehop_s)
                torch.cuda.synchronize()
                engine.profile_enable(False)
                strict["ms"], strict["launches"] = engine.profile_read()
        finally:
            for e in engines:
                e.set_pointwise_mode("f16x3")

    rc = 0
    if rank == 0:
        out = {
            "metric": "yamnet_windows_per_s", "value": round(value, 1), "unit": "windows/s",
            "audio_seconds_per_s": round(value * framehop_s, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 (split-f16x3 MFMA products, f32 accumulate)",
            "dtype_note": "f32 in / f32 accumulate; the 1x1-conv products of `value` are NOT f32 products: split-f16 MFMA (hi+lo "
                          "halves of operands scaled by exact powers of two - per weight row at load, per layer from an "
                          "exact-f32 calibration pass - so that both halves are normal f16: 22-bit operands at any weight "
                          "scale; 3 MFMAs per product), inside north_star's 1e-4 on the logits against the f64 oracle.  "
                          "value_strict_f32 is the same workload on exact-f32 MFMA products (the reference's own precision); "
                          "value_end_to_end_f32_host is `value`'s workload starting from float32 PCM in pinned HOST memory "
                          "(the reference's dtype_in), PCIe copies inside the timed region (SURVEY 8d's end-to-end number)",
            "value_strict_f32": None, "value_end_to_end_f32_host": None,
            "value_scope": "PCM resident in HBM when the timed region starts; per step one batch's logits are read back on the "
                           "host through DeviceResult.numpy() (range verdict waited for and checked, D2H copy)",
            "timed_read_back": timed_read_back,
            "timed_region_s": round(elapsed, 4),
            "data": "synthetic" + (" (REHEARSAL: ranks share GPU 0, gloo)" if rehearsal else ""),
            "config": {"workload": f"config 2 x {files_per_step} per step: synthetic 1 h 16 kHz mono recordings, each fed as batches "
                                   "of 1024 windows (1024+1024+1024+678 = 3750 windows, 983.04 s chunks), embedder yamnet_k2 "
                                   "(mel Const of embedders/yamnet_k2) hop 1.0 + model_general_v3 head; embedder weights "
                                   "seeded synthetic in the reference layout, head weights real; --steps 20 = config 4's "
                                   "1000 x 1 h",
                       "recordings_per_step": files_per_step, "windows_per_recording": windows_per_file,
                       "windows_per_step": windows_per_step, "samples_per_step": FILE_SAMPLES * files_per_step,
                       "batch_windows": batch_windows, "hop_samples": hop, "patch_step": step,
                       "analyzer_streams": len(engines),
                       "sharding": (f"config 4 shape: {args.steps * files_per_step} recordings in total, recording i -> rank i mod {world}, "
                                    f"remainder dealt per batch; one RCCL gather of the [rows,13] logit blocks to rank 0 "
                                    f"per round of {world} recordings, sizes known on the host") if world > 1 else "single GPU",
                       "timing": "value from K clean steps; per-kernel HIP-event times from a second region "
                                 f"({ev_steps} recordings, one stream)"},
        }
        if args.pointwise_mode is not None:
            out["mode_override"] = args.pointwise_mode
        if ranks_seen is not None:
            out["ranks_seen"] = ranks_seen
            out["ranks_seen_note"] = ("counted on rank 0 from the gathered blocks of the timed region: rounds in which the block "
                                      "of that rank arrived with its marker, windows it announced; planned total "
                                      f"{total_windows} windows")
        power = watch.summary() if rank == 0 else None
        if power:
            out["power"] = power
            log(f"board power {power['avg_W']} W of {power['cap_W']} W, shader clock {power['sclk_MHz_avg']} MHz")
        stem_kernel = {None: "stem_reg_kernel", 3: "stem_reg_kernel"}.get(args.stem, "stem3_kernel")
        if events_on and launches.sum() > 0 and args.per_slot:
            plan = (slot_plan_f32(launches, "stem_reg_f32_kernel" if args.stem in (None, 3) else "stem3_f32_kernel") if args.pointwise_mode == "f32"
                    else slot_plan(launches, stem_kernel=stem_kernel))
            for slot, (nm, fam, nb, fl) in sorted(plan.items()):
                us = 1e3 * ms[slot] / max(int(launches[slot]), 1)
                wl = windows_per_file * ev_steps / max(int(launches[slot]), 1)     # windows per launch on average
                log(f"slot {slot:2d} {nm:10s} {fam:24s} {us:8.1f} us  {nb * wl / us / 1e6:6.2f} TB/s  "
                    f"{fl * wl / us / 1e6:7.1f} TFLOP/s   (avg {wl:.0f} windows/launch)")
        if events_on and launches.sum() > 0:
            out["ms_per_recording_with_kernel_events"] = round(1e3 * elapsed_events / ev_steps, 4)
            out["ms_per_recording"] = round(1e3 * elapsed / (args.steps * files_per_step), 4)
            fams = {}
            for slot, (nm, fam, nb, fl) in slot_plan(launches, stem_kernel=stem_kernel).items():
                f = fams.setdefault(fam, {"ms": 0.0, "launches": 0, "bytes": 0, "flops": 0, "slots": []})
                f["ms"] += ms[slot]
                f["launches"] += int(launches[slot])
                f["bytes"] += nb * windows_per_file * ev_steps
                f["flops"] += fl * windows_per_file * ev_steps
                f["slots"].append(nm)
            total_ms = float(ms.sum())
            mfma_fams = ("pointwise_f16x3_kernel", "sep_ws_kernel", "sep_chip_kernel", "sep_mid_kernel", "tail_gemm_kernel", "stem3_kernel", "stem_reg_kernel",
                         "pw_res_kernel", "l4_window_kernel")
            dom = max(fams, key=lambda k: fams[k]["ms"])
            d = fams[dom]
            sec = d["ms"] * 1e-3
            if dom in mfma_fams:
                achieved = d["flops"] / sec / 1e12
                out["roofline"] = {
                    "kernel": f"{dom} ({', '.join(d['slots'])}: {d['launches'] // (ev_steps * len(batches))} launches per batch)",
                    "bound": "mfma", "achieved": round(achieved, 2), "peak": round(PEAK_SPLIT_F16_TFLOPS, 1),
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_SPLIT_F16_TFLOPS, 4), "traffic": None,
                    "peak_note": "achieved = algorithmic f32-equivalent FLOP/s (1x1 conv + depthwise of the layers this "
                                 "kernel runs); every product is 3 f16 MFMAs, so peak = 2500 TFLOP/s dense f16 / 3",
                    "executed_f16_mfma_tflops": round(3 * achieved, 1),
                    "peak_sustained": round(SUSTAINED_F16_MFMA_TFLOPS / 3.0, 1),
                    "frac_of_sustained": round(3 * achieved / SUSTAINED_F16_MFMA_TFLOPS, 4),
                    "peak_sustained_note": "back-to-back f16 MFMAs alone hold 1987 TFLOP/s at the 1400 W board limit (2.0 GHz; "
                                           "profiles/r03_ubench_power.txt), not the 2500 of the 2.4 GHz paper peak",
                    "hbm_GBps_algorithmic": round(d["bytes"] / sec / 1e9, 1),
                }
            else:
                gbs = d["bytes"] / sec / 1e9
                out["roofline"] = {"kernel": f"{dom} ({', '.join(d['slots'])})", "bound": "hbm",
                                   "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
            try:      # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs)
                with open(os.path.join(REPO, PMC_TRAFFIC_FILE)) as f:
                    pmc = json.load(f)["per_kernel_family"].get(dom)
                if pmc:
                    out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = PMC_TRAFFIC_FILE + " ((2*FETCH_SIZE + WRITE_SIZE)*1024 per launch)"
            except (OSError, ValueError, KeyError):
                pass
            # the whole path, not only its best kernel: f16 MFMA FLOP/s executed at `value` (3 MFMAs per 1x1-conv product)
            # over the 2.5 PFLOP/s dense f16 peak
            out["roofline"]["pipeline_frac"] = round(3.0 * POINTWISE_FLOP_PER_WINDOW * value / world / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)
            out["roofline"]["pipeline_frac_note"] = ("executed f16 MFMA FLOP/s of the whole hot path at `value` per GPU (3 x 132.12 "
                                                     "MFLOP per window) / 2500 TFLOP/s: every kernel, launch gap and the "
                                                     "front end included")
            out["roofline"].update({"avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                                    "launches": d["launches"],
                                    "flop_per_launch_avg": d["flops"] // d["launches"],
                                    "bytes_per_launch_avg": d["bytes"] // d["launches"],
                                    "share_of_step_time": round(d["ms"] / total_ms, 4)})
            stages = {}
            for fam, k in sorted(fams.items(), key=lambda kv: -kv[1]["ms"]):
                sec = k["ms"] * 1e-3
                stages[fam] = {"slots": k["slots"], "ms_per_recording": round(k["ms"] / ev_steps, 4),
                               "share": round(k["ms"] / total_ms, 4),
                               "GBps_algorithmic": round(k["bytes"] / sec / 1e9, 1),
                               "frac_hbm_peak": round(k["bytes"] / sec / 1e9 / PEAK_HBM_GBS, 4),
                               "TFLOPs_algorithmic": round(k["flops"] / sec / 1e12, 2)}
            out["stages"] = stages

        if strict is not None:
            out["value_strict_f32"] = strict["value"]
            out["strict_f32"] = {"value": strict["value"], "unit": "windows/s", "steps": strict["steps"],
                                 "ms_per_step": strict["ms_per_step"], "timed_read_back": strict["read_back"],
                                 "workload": out["config"]["workload"],
                                 "what": "`value`'s workload and loop (run_files: the same recordings, batches, analyzer streams and "
                                         "per-step read-back) with bd_set_pointwise_mode 0: every 1x1-convolution product on "
                                         "v_mfma_f32_32x32x2_f32, exact f32 products"}
            tfw = strict["value"] * CNN_FLOP_PER_WINDOW / 1e12
            out["strict_f32"]["pipeline_frac"] = round(tfw / PEAK_F32_MFMA_TFLOPS, 4)
            if "ms" in strict and strict["launches"].sum() > 0:
                sms, sl = strict["ms"], strict["launches"]
                sf = {}
                for slot, (nm, fam, nb, fl) in slot_plan_f32(sl, "stem_reg_f32_kernel" if args.stem in (None, 3) else "stem3_f32_kernel").items():
                    f = sf.setdefault(fam, {"ms": 0.0, "launches": 0, "bytes": 0, "flops": 0, "slots": []})
                    f["ms"] += sms[slot]
                    f["launches"] += int(sl[slot])
                    f["bytes"] += nb * windows_per_file * ev_steps
                    f["flops"] += fl * windows_per_file * ev_steps
                    f["slots"].append(nm)
                    if args.per_slot:
                        us = 1e3 * sms[slot] / max(int(sl[slot]), 1)
                        wl = windows_per_file * ev_steps / max(int(sl[slot]), 1)
                        log(f"f32 slot {slot:2d} {nm:12s} {fam:20s} {us:8.1f} us  {fl * wl / us / 1e6:7.1f} TFLOP/s")
                sdom = max(sf, key=lambda k: sf[k]["ms"])
                d = sf[sdom]
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                out["roofline_strict"] = {
                    "kernel": f"{sdom} ({', '.join(d['slots'])})", "bound": "mfma", "achieved": round(ach, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2), "launches": d["launches"],
                    "flop_per_launch_avg": d["flops"] // d["launches"], "bytes_per_launch_avg": d["bytes"] // d["launches"],
                    "share_of_step_time": round(d["ms"] / float(sms.sum()), 4),
                    "peak_note": "v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md); achieved = algorithmic FLOP of the 1x1 "
                                 "convolutions + the depthwise in their epilogues over HIP-event time of their launches",
                    "stages": {fam: {"slots": k["slots"], "ms_per_recording": round(k["ms"] / ev_steps, 4),
                                     "share": round(k["ms"] / float(sms.sum()), 4),
                                     "TFLOPs_algorithmic": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2)}
                               for fam, k in sorted(sf.items(), key=lambda kv: -kv[1]["ms"])}}
                try:
                    with open(os.path.join(REPO, PMC_TRAFFIC_STRICT_FILE)) as f:
                        pmc = json.load(f)["per_kernel_family"].get(sdom)
                    if pmc:
                        out["roofline_strict"]["traffic"] = pmc["hbm_bytes_per_launch"]
                        out["roofline_strict"]["traffic_source"] = PMC_TRAFFIC_STRICT_FILE
                except (OSError, ValueError, KeyError):
                    pass
        if world == 1 and not args.no_extras:
            try:
                extras(out, args, engines, streams, device, dev_index, hop, step, framehop_s)
            except Exception as exc:                     # an extra leg must never take the headline line down
                log(f"extras failed: {type(exc).__name__}: {exc}")
                out["extras_error"] = f"{type(exc).__name__}: {exc}"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(engine, hop, step, args.cpu_windows)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def extras(out, args, engines, streams, device, dev_index, hop, step, framehop_s):
    """Numbers reported BESIDE `value` (never as it): host-resident input, file -> CSV, other arithmetic modes."""
    import numpy as np
    import torch
    k = 300                            # batches per leg: ~0.2 s of GPU work each (a 20-batch leg measures launch jitter)
    # two analyzer streams for the host-resident legs: more only get in the copy engine's way (1.52 M with two, 1.31 M with four)
    v, gbs = h2d_leg(engines[:2], streams[:2], device, hop, framehop_s, k, s16=True)
    out["value_h2d_s16"] = {"value": v, "unit": "windows/s", "pcie_GBps": gbs,
                            "what": "config-2 batches starting in pinned HOST memory as 16-bit PCM: async H2D on a copy "
                                    "stream (4-deep device ring, reused when the host has seen the batch finish) + "
                                    "device-side s16->f32 + predict, copies inside the timed region"}
    log(f"host-resident s16 batches: {v:.0f} windows/s ({gbs} GB/s over PCIe)")
    v, gbs = h2d_leg(engines[:2], streams[:2], device, hop, framehop_s, k, s16=False)
    out["value_h2d_f32"] = {"value": v, "unit": "windows/s", "pcie_GBps": gbs,
                            "what": "the same with float32 PCM on the host (the reference's dtype_in)"}
    out["value_end_to_end_f32_host"] = v
    log(f"host-resident f32 batches: {v:.0f} windows/s ({gbs} GB/s over PCIe)")

    # the front-end kernel alone on one 1024-window batch (98 304 frames), HIP events on its stream: the north star's
    # ">= 60 % of the HBM roofline" is a statement about THIS kernel; in situ it is ~6 % of a step
    x_fe = synthetic_audio(device, WINDOWS_PER_BATCH * hop, 4711)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        lm = engines[0].frontend(x_fe, hop)
    torch.cuda.synchronize()
    reps = 100
    e0.record()
    for _ in range(reps):
        lm = engines[0].frontend(x_fe, hop)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    fe_bytes = lm.shape[0] * (STFT_HOP_BYTES + 64 * 4)
    # issue bound from the kernel's own counters (rocprofv3 --pmc SQ_INSTS_VALU on tools/fe_bench.py, committed under
    # profiles/): vector instructions per launch / 1024 SIMDs x ~4 cycles per instruction (v_pk_*_f32 measure 3.3-4.5
    # cycles per SIMD, tools/ubench_lane_ops.hip) at the ~2.1 GHz the kernel runs at - a floor no memory-side change beats
    fe_frames = lm.shape[0]
    valu = None
    try:
        import csv
        with open(os.path.join(REPO, PMC_FRONTEND_FILE)) as f:
            rows = [r for r in csv.DictReader(f) if r["kernel"] == "logmel_kernel"]
        valu = max(float(r["SQ_INSTS_VALU"]) for r in rows)
    except (OSError, ValueError, KeyError):
        pass
    # how close the kernel is to its OWN instruction floor.  Lane-instructions a packed-f32 16 x 16 FFT
    # front end needs per 400-sample frame (a v_pk_*_f32 on a (re, im) pair counts as one):
    #   Hann window                          200   (200 packed complex points)
    #   32 DFT-16 (16 over n1, 16 over n2)  2432   (8 radix-4 butterflies x 8 packed add/sub + 12 for the W16 twiddles, each)
    #   twiddles between the two passes      450   (225 non-trivial complex multiplies x 2)
    #   real-input split X[k], X[256 - k]    768   (128 bin pairs x 6)
    #   |X| = sqrt(re^2 + im^2)              771   (257 bins x 3)
    #   banded mel matrix                    461   (the matrix's non-zeros, one FMA each)
    #   log(x + 0.001)                       192   (64 bands x 3)
    fe_needed = 200 + 2432 + 450 + 768 + 771 + 461 + 192
    issue = None
    valu_floor_frac = None
    if valu:
        fe_issued = valu * 64.0 / fe_frames
        valu_floor_frac = round(fe_needed / fe_issued, 4)
        floor_us = valu / 1024.0 * 4.0 / 2.1e3
        issue = {"valu_wave_instructions_per_launch": valu, "source": PMC_FRONTEND_FILE, "simds": 1024,
                 "cycles_per_instruction": 4.0, "clock_GHz": 2.1, "issue_floor_us": round(floor_us, 1),
                 "frac_of_launch": round(floor_us / us, 3),
                 "what": "vector-instruction issue time alone; the rest of a launch is LDS transposes (LDS ~50 % busy) and two "
                         "workgroup barriers per 64 frames: the kernel is issue / latency bound, not HBM bound (DESIGN.md 4.3)"}
    out["frontend_roofline"] = {
        "kernel": "logmel_kernel alone, 98 304 frames (one 1024-window batch), PCM resident in HBM",
        "bound": "hbm", "achieved": round(fe_bytes / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": round(fe_bytes / us / 1e3 / PEAK_HBM_GBS, 4), "avg_launch_us": round(us, 2), "bytes_per_launch": fe_bytes,
        "traffic": None, "north_star_gate_frac": 0.60, "gate_met": bool(fe_bytes / us / 1e3 / PEAK_HBM_GBS >= 0.60),
        "valu_floor_frac": valu_floor_frac,
        "valu_floor": {"lane_instructions_per_frame_needed": fe_needed,
                       "lane_instructions_per_frame_issued": round(valu * 64.0 / fe_frames, 1) if valu else None,
                       "what": "arithmetic a packed-f32 16 x 16 FFT + window + split + |.| + 461-FMA mel + log needs per frame "
                               "(breakdown in bench.py) over what the kernel issues (SQ_INSTS_VALU x 64 lanes / frames): the "
                               "kernel is within ~10-15 % of its own instruction floor, and that floor alone is ~0.5 of the "
                               "launch - the 0.60 HBM gate is out of reach for an f32 vector FFT at 5e-5 (DESIGN.md 10)"},
        "issue_bound": issue}
    log(f"front end alone: {us:.1f} us per {fe_frames} frames = {fe_bytes / us / 1e3:.0f} GB/s "
        f"({100 * fe_bytes / us / 1e3 / PEAK_HBM_GBS:.1f} % of the HBM peak)")
    del x_fe, lm

    legs = {}
    for name, hours, chunk, hp, nfiles in (("config2_1h_hop1.0", 1, WINDOWS_PER_BATCH * FRAMELENGTH_S, 1.0, 1),
                                           ("config3_24h_600s_hop1.0", 24, 600.0, 1.0, 1),
                                           ("config3_24h_600s_hop0.5", 24, 600.0, 0.5, 1),
                                           # config 4's single-rank share in shape: many one-hour recordings in one call
                                           ("config4_share_50x1h_hop1.0", 1, WINDOWS_PER_BATCH * FRAMELENGTH_S, 1.0, 50),
                                           ("config4_share_50x1h_hop0.5", 1, 600.0, 0.5, 50)):
        legs[name] = analyze_leg(dev_index, hours, chunk, hp, engines=engines[:2], files=nfiles)
        log(f"analyze() {name}: {legs[name]['audio_s_per_s']:.0f} audio-s/s, {legs[name]['windows_per_s']:.0f} windows/s "
            f"(second call: {legs[name]['second_call']['audio_s_per_s']:.0f} audio-s/s, {legs[name]['second_call']['windows_per_s']:.0f} "
            f"windows/s); busy {legs[name]['second_call']['stage_busy_s']}")
    # (round 5's `sustained_hop1.0` - (24 h - 1 h) / (t_24h - t_1h) - was a difference of two FIRST calls and measured the
    #  page-locking of that round's staging ring; the sustained rate of a leg is its `second_call`)
    out["analyze_audio_s_per_s"] = {"what": "analyze(): 16-bit WAV on tmpfs -> reference-format CSV, wall clock of the whole "
                                            "call (file read, H2D, device conversion, hot path, D2H, CSV) with two prebuilt "
                                            "engines (2 analyzer threads, 6 reader threads)", **legs}

    # config 5 on one GPU: 48 kHz stereo 16-bit PCM resident in HBM -> channel mean + 3:1 polyphase resample in one kernel
    # -> hot path in plain-f16 mode; and that kernel alone against its own HBM roofline
    n48 = WINDOWS_PER_BATCH * hop * 3
    st = torch.stack([synthetic_audio(device, n48, 31), synthetic_audio(device, n48, 32)], 1)
    st = (st * 32768.0).round().clamp_(-32768, 32767).to(torch.int16).contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        mono = engines[0].resample(st, 48000, SAMPLE_RATE)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(k):
        mono = engines[0].resample(st, 48000, SAMPLE_RATE)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / k
    rs_bytes = st.numel() * 2 + mono.numel() * 4
    # the reference's filter class on the matrix cores (resample.hip): 569 taps, K = 704 per 32 outputs incl. the band's zero
    # corners, 3 f16 MFMAs per product -> executed f16 FLOP per output = 2 * 3 * 704; algorithmic f32-equivalent = 2 * 569
    out["resample_roofline"] = {"kernel": "fir_mfma_kernel<short, 11, 4> (48 kHz stereo s16 -> 16 kHz mono f32; soxr_hq-class "
                                          "569-tap low-pass as a Toeplitz product in split-f16 on the matrix cores)",
                                "bound": "hbm", "achieved": round(rs_bytes / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(rs_bytes / us / 1e3 / PEAK_HBM_GBS, 4), "avg_launch_us": round(us, 2),
                                "bytes_per_launch": rs_bytes,
                                "mfma_executed_f16_tflops": round(mono.numel() * 2 * 3 * 704 / us / 1e6, 1),
                                "mfma_frac": round(mono.numel() * 2 * 3 * 704 / us / 1e6 / PEAK_F16_MFMA_TFLOPS, 4),
                                "bound_note": "both ceilings are within 2x of each other for this stage (HBM floor ~50 us at the "
                                              "6.3 TB/s a copy reaches, matrix floor ~30 us); as vector code the 569-tap filter "
                                              "would be compute bound at ~180 us.  The 61-tap filter of rounds 1-3 "
                                              "(bd_set_resample_quality 0) is resample_roofline_scipy"}
    engines[0].set_resample_quality("scipy")
    for _ in range(3):
        mono = engines[0].resample(st, 48000, SAMPLE_RATE)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(k):
        mono = engines[0].resample(st, 48000, SAMPLE_RATE)
    e1.record()
    torch.cuda.synchronize()
    engines[0].set_resample_quality("hq")
    us0 = 1e3 * e0.elapsed_time(e1) / k
    out["resample_roofline_scipy"] = {"kernel": "decimate_kernel<short, 3> (61-tap scipy.signal.resample_poly default: NOT the "
                                                "reference's filter class)", "bound": "hbm",
                                      "achieved": round(rs_bytes / us0 / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(rs_bytes / us0 / 1e3 / PEAK_HBM_GBS, 4), "avg_launch_us": round(us0, 2)}
    log(f"resample 48k stereo s16 -> 16k mono: {us:.1f} us per batch = {rs_bytes / us / 1e3:.0f} GB/s")
    for e in engines:
        e.set_pointwise_mode("f16")
    for i in range(4):
        engines[i % len(engines)].predict(engines[i % len(engines)].resample(st, 48000, SAMPLE_RATE), framehop_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        with torch.cuda.stream(streams[i % len(streams)]):
            e = engines[i % len(engines)]
            e.predict(e.resample(st, 48000, SAMPLE_RATE), framehop_s)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    for e in engines:
        e.set_pointwise_mode("f16x3")
    out["config5_1gpu"] = {"value": round(k * WINDOWS_PER_BATCH / sec, 1), "unit": "windows/s",
                           "what": "BASELINE config 5 on ONE GPU: 48 kHz stereo 16-bit PCM resident in HBM, downmix + 3:1 resample "
                                   "in-kernel, plain-f16 MFMA mode (value_mode2_f16 has its max|dlogit|); never `value`"}
    log(f"config 5 on one GPU (48 kHz stereo in, plain f16): {out['config5_1gpu']['value']:.0f} windows/s")
    del st, mono

    # the other arithmetic modes of the 1x1 convolutions, same K batches of 1024 windows, reported beside `value`
    x = synthetic_audio(device, WINDOWS_PER_BATCH * hop, 99)
    ref = engines[0].predict(x, framehop_s).numpy().copy()
    modes = {}
    for mode in getattr(engines[0], "POINTWISE_MODES", ("f32",)):
        if mode == "f16x3":
            continue
        for e in engines:
            e.set_pointwise_mode(mode)
        got = engines[0].predict(x, framehop_s).numpy()
        for i in range(4):
            engines[i % len(engines)].predict(x, framehop_s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            with torch.cuda.stream(streams[i % len(streams)]):
                engines[i % len(engines)].predict(x, framehop_s)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        modes[mode] = {"value": round(k * WINDOWS_PER_BATCH / sec, 1), "unit": "windows/s",
                       "max_abs_dlogit_vs_default_mode": float(np.abs(got - ref).max())}
        log(f"pointwise mode {mode}: {modes[mode]['value']:.0f} windows/s, max|dlogit| vs default {modes[mode]['max_abs_dlogit_vs_default_mode']:.2e}")
    for e in engines:
        e.set_pointwise_mode("f16x3")
    if "f32" in modes:
        out["value_mode0_f32"] = {**modes["f32"], "what": "1x1 convolutions on v_mfma_f32_32x32x2_f32 (exact f32 products); k x ONE "
                                                          "repeated 1024-window batch, no read-back (value_strict_f32 is the headline loop)"}
        if out.get("value_strict_f32") is None:
            out["value_strict_f32"] = modes["f32"]["value"]
        tf = modes["f32"]["value"] * CNN_FLOP_PER_WINDOW / 1e12
        out["roofline_mode0"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                                 "what": "whole CNN (137.29 MFLOP per window) at value_mode0_f32's rate against the dense "
                                         "f32-MFMA peak; the strict-f32 number's own fraction"}
    if "f16" in modes:
        out["value_mode2_f16"] = {**modes["f16"], "what": "config 5 arithmetic: plain f16 operands, one MFMA per product, f32 "
                                                          "accumulate; outside the 1e-4 gate by design, never `value`"}


if __name__ == "__main__":
    sys.exit(main())
