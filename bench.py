#!/usr/bin/env python3
"""Throughput of the analyze hot path on MI355X (BASELINE.json metric: YAMNet windows/s).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the whole hot path (PCM -> log-mel -> YAMNet -> dense head) over one batch
of synthetic 16 kHz mono audio already resident in HBM: BASELINE config 2's batch of 1024 windows
(983.04 s = 15 728 640 samples, yamnet_k2 embedder at hop 1.0, model_general_v3 head).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank runs its own batch per step — the
round-robin file sharding of config 4 — and the per-window logits are gathered to rank 0 over RCCL
inside the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from buzzdetect_amd import sharding  # noqa: E402
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step  # noqa: E402

SAMPLE_RATE = 16000
WINDOWS_PER_BATCH = 1024
HOP_PROP = 1.0
FRAMELENGTH_S = 0.96

# SURVEY §8d / DESIGN.md: algorithmic work per window at hop 1.0
POINTWISE_FLOP_PER_WINDOW = 132_120_576          # 2 * 66 060 288 MAC in the thirteen 1x1 convolutions
CNN_FLOP_PER_WINDOW = 137_289_728
FRONTEND_BYTES_PER_WINDOW = 61_440 + 24_576      # f32 PCM in + f32 log-mel out
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_SPLIT_F16_TFLOPS = 2500.0 / 3.0             # dense f16 MFMA peak / 3 MFMAs per f32-accurate product
PEAK_HBM_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E spec

# per-window HBM bytes each depthwise / conv1 launch must move (read input + write output, f32 NHWC)
_DEF = ((2, 32), (1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512), (1, 512),
        (1, 512), (1, 512), (2, 1024), (1, 1024))


def slot_plan(launches, pool_fused=True):
    """Map the 29 profile slots to (slot name, kernel family, per-window algorithmic bytes, per-window flops)
    for the launches that actually happened (fused kernels are timed in the pointwise slot of their layer)."""
    plan = {0: ("frontend", "logmel_kernel", FRONTEND_BYTES_PER_WINDOW, 0)}
    h, w, c = 48, 32, 32
    conv1 = (96 * 64 * 4 + h * w * c * 4, 2 * 9 * h * w * c)
    if launches[1] > 0:
        plan[1] = ("conv1", "conv1_kernel", conv1[0], conv1[1])
    for layer, (stride, cout) in enumerate(_DEF[1:], start=2):
        ho, wo = h // stride, w // stride
        dw_slot, pw_slot = 2 * layer - 2, 2 * layer - 1
        dw = ((h * w * c + ho * wo * c) * 4, 2 * 9 * ho * wo * c)
        pw = ((ho * wo * c + ho * wo * cout) * 4, 2 * ho * wo * c * cout)
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
            plan[pw_slot] = ("stem(1-3)", "stem3_kernel", 96 * 64 * 4 + ho * wo * cout * 4,
                             conv1[1] + stem_flops + dw[1] + pw[1])
        elif launches[pw_slot] > 0:
            if launches[dw_slot] > 0 or (stride == 2 and layer >= 3):
                # (a stride-2 layer without a depthwise launch: the previous kernel applied it)
                # (engine rule, cnn.hip launch_pointwise_ws: K >= 128 and N % 256 == 0 run on the wave-specialised
                #  kernel with pass-through producers; the others on the plain split-f16 GEMM kernel)
                fam = "sep_ws_kernel" if (c >= 128 and c % 64 == 0 and cout % 256 == 0) else "pointwise_f16x3_kernel"
                plan[pw_slot] = (f"pw{layer}", fam, pw[0], pw[1])
                if launches[dw_slot] == 0 and layer >= 5 and (pw_slot - 2) in plan:
                    nm, fam, nb, fl = plan[pw_slot - 2]
                    # the fused kernel of the previous layer wrote this layer's depthwise output instead of its own
                    # (epilogue fusion exists only in the 8-wave kernel)
                    plan[pw_slot - 2] = (nm + f"+dw{layer}", "sep_ws_kernel", nb - h * w * c * 4 + ho * wo * c * 4, fl + dw[1])
            elif layer == 2:      # fused stem: log-mel patch in, layer-2 output out
                plan[pw_slot] = ("stem(1-2)", "stem_kernel", 96 * 64 * 4 + ho * wo * cout * 4,
                                 conv1[1] + dw[1] + pw[1])
            else:                 # depthwise inside the GEMM: layer input in, layer output out
                # fused stride-1 layers run the wave-specialised kernel; 512 -> 512 channels its 12-wave form (default path)
                fam = "sep_w12_kernel" if (pool_fused and c == 512 and cout == 512) else "sep_ws_kernel"
                if layer == 14 and pool_fused:    # the average pool rides in the epilogue: [1024] out per window
                    plan[pw_slot] = ("sep14+pool", fam, (h * w * c + cout) * 4, dw[1] + pw[1] + ho * wo * cout)
                else:
                    plan[pw_slot] = (f"sep{layer}", fam, (h * w * c + ho * wo * cout) * 4, dw[1] + pw[1])
        h, w, c = ho, wo, cout
    if pool_fused and 27 in plan and plan[27][0] == "sep14+pool":
        plan[28] = ("head", "pool_head_kernel", (1024 + 13) * 4, 2 * 1024 * 13)
    else:
        plan[28] = ("pool_head", "pool_head_kernel", (6 * 1024 + 13) * 4, 2 * 1024 * 13)
    return plan


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def synthetic_batch(device, n_samples: int, seed: int) -> torch.Tensor:
    """0.1*N(0,1) noise + 0.3*sin(2*pi*220 t) bursts of 0.5 s every 5 s, clipped to [-1,1)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    x = 0.1 * torch.randn(n_samples, generator=gen, device=device, dtype=torch.float32)
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / SAMPLE_RATE
    burst = (torch.remainder(t, 5.0) < 0.5).to(torch.float32)
    x += 0.3 * torch.sin(2 * np.pi * 220.0 * t) * burst
    return x.clamp_(-1.0, 1.0 - 2.0 ** -23)


def cpu_baseline(engine: HipEngine, hop: int, step: int, windows: int):
    """The oracle timed on this box's host cores (checker + reported baseline, never the product).
    The sample is cut into the workload's own 1024-window chunks on both sides."""
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
    for rep in range(3):
        t0 = time.perf_counter()
        outs = [model.predict(w, hop, step) for w in waves]
        passes.append(time.perf_counter() - t0)
        cpu_logits = outs
        log(f"cpu_baseline: pass {rep}: {chunks * WINDOWS_PER_BATCH / passes[-1]:.1f} windows/s")
    med = sorted(passes)[1]
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
                  f"1 warm-up + median of 3 passes",
        "seconds_per_pass": round(med, 3), "cpu_model": cpu_model,
        "max_abs_dlogit_gpu_vs_cpu_f32": d32,
        "max_abs_dlogit_gpu_vs_cpu_f64_first8": float(np.abs(gpu_logits[0][:8] - ref64).max()),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cpu-windows", type=int, default=4096,
                    help="size of the bounded CPU-baseline sample (about 10-20 s of CPU work over 4 passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--per-slot", action="store_true", help="print per-kernel times to stderr")
    ap.add_argument("--streams", type=int, default=2,
                    help="analyzer streams per GPU: steps are dealt round-robin to this many engines, each on "
                         "its own HIP stream (the reference's analyzers_gpu knob, src/analyze.py:218-253)")
    ap.add_argument("--sep-variant", type=int, default=None, help="fused separable-layer kernel variant (tuning)")
    ap.add_argument("--pw-variant", type=int, default=None, help="tuning: kernel variant of the plain 1x1 convolutions (layers 5-14)")
    ap.add_argument("--frontend-variant", type=int, default=None, help="front-end FFT formulation (tuning)")
    ap.add_argument("--group-windows", type=int, default=0, help="windows per CNN pass (0 = library default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU path for the product")
    # BD_BENCH_REHEARSAL=1: every rank uses GPU 0 and the collectives run over gloo on host copies - only to
    # exercise the multi-rank control flow on a one-GPU box; never a measurement.
    rehearsal = os.environ.get("BD_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=dev_index)
               for _ in range(max(1, args.streams))]
    streams = [torch.cuda.current_stream(device)] + [torch.cuda.Stream(device) for _ in engines[1:]]
    engine = engines[0]
    for e in engines:
        if args.group_windows:
            e.set_group_windows(args.group_windows)
        if args.sep_variant is not None:
            e.set_fusion(True, args.sep_variant)
        if args.pw_variant is not None:
            for layer in range(5, 15):
                e.set_pointwise_variant(layer, args.pw_variant)
        if args.frontend_variant is not None:
            e.set_frontend_variant(args.frontend_variant)
    framehop_s = FRAMELENGTH_S * HOP_PROP
    hop, step = hop_samples(framehop_s), patch_step(framehop_s)
    n_samples = hop * WINDOWS_PER_BATCH                      # 15 728 640
    assert engine.num_windows(n_samples, hop, step) == WINDOWS_PER_BATCH

    # a few distinct batches so that a step never re-reads the PCM it has just processed
    batches = [synthetic_batch(device, n_samples, 1234 + rank * 16 + i) for i in range(4)]
    torch.cuda.synchronize()

    def one_step(i: int):
        j = i % len(engines)
        with torch.cuda.stream(streams[j]):
            res = engines[j].predict(batches[i % len(batches)], framehop_s)
        gathered = None
        if world > 1:             # one communicator: collectives are issued in step order on stream 0
            streams[0].wait_stream(streams[j])
            with torch.cuda.stream(streams[0]):
                gathered = sharding.gather_rows(res.tensor.cpu() if rehearsal else res.tensor, dst=0)
                res.tensor.record_stream(streams[0])
        return res, gathered

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"warm-up {args.warmup} steps, then {args.steps} timed steps on {world} GPU(s)")
    for i in range(args.warmup):
        one_step(i)
    fence()
    def timed_region(steps: int, single_stream: bool = False) -> float:
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            if single_stream:     # per-kernel event timing needs the kernels of one stream back to back
                engine.predict(batches[i % len(batches)], framehop_s)
            else:
                one_step(i)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # region 1: the number reported as `value` — nothing but the hot path on the stream
    engine.profile_enable(False)
    elapsed = timed_region(args.steps)
    # region 2: the same K steps again with every kernel bracketed by HIP events on its own stream
    # (costs a few % of wall time, which is why it is not the region `value` comes from)
    events_on = not args.no_kernel_events
    ms = launches = None
    elapsed_events = None
    if events_on:
        engine.profile_read()                                # drop anything recorded so far
        engine.profile_enable(True)
        elapsed_events = timed_region(args.steps, single_stream=True)
        engine.profile_enable(False)
        ms, launches = engine.profile_read()

    total_windows = WINDOWS_PER_BATCH * args.steps * world
    value = total_windows / elapsed
    if rank == 0:
        log(f"{value:.0f} windows/s ({1e3 * elapsed / args.steps:.3f} ms/step)")

    if rank == 0:
        out = {
            "metric": "yamnet_windows_per_s", "value": round(value, 1), "unit": "windows/s",
            "audio_seconds_per_s": round(value * framehop_s, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share GPU 0, gloo)" if rehearsal else ""),
            "config": {"workload": "config 2: 16 kHz mono, batches of 1024 windows (983.04 s), yamnet_k2 hop 1.0 "
                                   "+ model_general_v3 head; embedder weights seeded synthetic in the reference "
                                   "layout, head weights real",
                       "windows_per_step_per_gpu": WINDOWS_PER_BATCH, "samples_per_step_per_gpu": n_samples,
                       "hop_samples": hop, "patch_step": step, "analyzer_streams": len(engines), "sharding": "round-robin batches per rank, "
                       "RCCL gather of [W,13] logits to rank 0 each step" if world > 1 else "single GPU",
                       "timing": "value from K clean steps; per-kernel HIP-event times from a second identical K-step region"},
        }
        if events_on and launches.sum() > 0 and args.per_slot:
            for slot, (nm, fam, nb, fl) in sorted(slot_plan(launches, pool_fused=args.sep_variant is None).items()):
                us = 1e3 * ms[slot] / max(int(launches[slot]), 1)
                log(f"slot {slot:2d} {nm:10s} {fam:24s} {us:8.1f} us  {nb * WINDOWS_PER_BATCH / us / 1e6:6.2f} TB/s  "
                    f"{fl * WINDOWS_PER_BATCH / us / 1e6:7.1f} TFLOP/s")
        if events_on and launches.sum() > 0:
            out["ms_per_step_with_kernel_events"] = round(1e3 * elapsed_events / args.steps, 4)
            fams = {}
            for slot, (nm, fam, nb, fl) in slot_plan(launches, pool_fused=args.sep_variant is None).items():
                f = fams.setdefault(fam, {"ms": 0.0, "launches": 0, "bytes": 0, "flops": 0, "slots": []})
                f["ms"] += ms[slot]
                f["launches"] += int(launches[slot])
                f["bytes"] += nb * WINDOWS_PER_BATCH * args.steps
                f["flops"] += fl * WINDOWS_PER_BATCH * args.steps
                f["slots"].append(nm)
            total_ms = float(ms.sum())
            mfma_fams = ("pointwise_f16x3_kernel", "sep_s1_kernel", "sep_ws_kernel", "sep_w12_kernel", "stem_kernel", "stem3_kernel")
            dom = max(fams, key=lambda k: fams[k]["ms"])
            d = fams[dom]
            sec = d["ms"] * 1e-3
            if dom in mfma_fams:
                achieved = d["flops"] / sec / 1e12
                out["roofline"] = {
                    "kernel": f"{dom} ({', '.join(d['slots'])}: {d['launches'] // args.steps} launches per step)",
                    "bound": "mfma", "achieved": round(achieved, 2), "peak": round(PEAK_SPLIT_F16_TFLOPS, 1),
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_SPLIT_F16_TFLOPS, 4), "traffic": None,
                    "peak_note": "achieved = algorithmic f32-equivalent FLOP/s (1x1 conv + depthwise of the layers this "
                                 "kernel runs); every product is 3 f16 MFMAs, so peak = 2500 TFLOP/s dense f16 / 3",
                    "executed_f16_mfma_tflops": round(3 * achieved, 1),
                    "hbm_GBps_algorithmic": round(d["bytes"] / sec / 1e9, 1),
                }
            else:
                gbs = d["bytes"] / sec / 1e9
                out["roofline"] = {"kernel": f"{dom} ({', '.join(d['slots'])})", "bound": "hbm",
                                   "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
            try:      # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs)
                with open(os.path.join(REPO, "profiles", "r01_pmc_traffic_1p3M.json")) as f:
                    pmc = json.load(f)["per_kernel_family"].get(dom)
                if pmc:
                    out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic_1p3M.json ((2*FETCH_SIZE + WRITE_SIZE)*1024 per launch)"
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"].update({"avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                                    "launches": d["launches"],
                                    "flop_per_launch_avg": d["flops"] // d["launches"],
                                    "bytes_per_launch_avg": d["bytes"] // d["launches"],
                                    "share_of_step_time": round(d["ms"] / total_ms, 4)})
            stages = {}
            for fam, k in sorted(fams.items(), key=lambda kv: -kv[1]["ms"]):
                sec = k["ms"] * 1e-3
                stages[fam] = {"slots": k["slots"], "ms_per_step": round(k["ms"] / args.steps, 4),
                               "share": round(k["ms"] / total_ms, 4),
                               "GBps_algorithmic": round(k["bytes"] / sec / 1e9, 1),
                               "frac_hbm_peak": round(k["bytes"] / sec / 1e9 / PEAK_HBM_GBS, 4),
                               "TFLOPs_algorithmic": round(k["flops"] / sec / 1e12, 2)}
            out["stages"] = stages
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(engine, hop, step, args.cpu_windows)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
