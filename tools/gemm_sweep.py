#!/usr/bin/env python3
"""Developer tool (GPU box): time every pointwise-GEMM tile variant on each YAMNet layer shape."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buzzdetect_amd import _lib

SHAPES = [  # (layer, rows per window, K, N)
    (2, 1536, 32, 64), (3, 384, 64, 128), (4, 384, 128, 128), (5, 96, 128, 256), (6, 96, 256, 256),
    (7, 24, 256, 512), (8, 24, 512, 512), (13, 6, 512, 1024), (14, 6, 1024, 1024)]

def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(1, 10))
    mode = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
    lib = _lib.load()
    dev = torch.device("cuda")
    stream = torch.cuda.current_stream().cuda_stream
    results = {}
    for layer, rpw, K, N in SHAPES:
        M = rpw * windows
        g = torch.Generator(device=dev).manual_seed(layer)
        A = torch.rand((M, K), generator=g, device=dev) * 2 - 0.5
        Wt = torch.randn((N, K), generator=g, device=dev) * (2.0 / K) ** 0.5
        bias = torch.randn(N, generator=g, device=dev) * 0.1
        ref = torch.relu(A[:4096].double() @ Wt.double().T + bias.double()).float()
        reft = torch.relu(A[-300:].double() @ Wt.double().T + bias.double()).float()
        C = torch.empty((M, N), device=dev)
        Whi = Wt.to(torch.float16); Wlo = (Wt - Whi.float()).to(torch.float16)
        def call(v):
            if mode == "f32":
                return lib.bd_debug_pointwise(A.data_ptr(), Wt.data_ptr(), bias.data_ptr(), C.data_ptr(), M, N, K, v, stream)
            return lib.bd_debug_pointwise_f16x3(A.data_ptr(), Whi.data_ptr(), Wlo.data_ptr(), bias.data_ptr(), C.data_ptr(), M, N, K, v, stream)
        flops = 2.0 * M * K * N
        line = []
        for v in variants:
            C.fill_(-1.0)
            rc = call(v)
            if rc != 0:
                line.append((v, None, None)); continue
            torch.cuda.synchronize()
            err = max((C[:4096] - ref).abs().max().item(), (C[-300:] - reft).abs().max().item())
            for _ in range(3):
                call(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                call(v)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            line.append((v, us, err))
        results[layer] = line
        txt = "  ".join(f"v{v}:{'--' if us is None else f'{us:7.1f}us {flops/us/1e6:5.1f}TF e{err:.0e}'}" for v, us, err in line)
        print(f"L{layer:2d} M={M:8d} K={K:4d} N={N:4d} | {txt}", flush=True)
    best = {l: min((x for x in r if x[1] is not None), key=lambda x: x[1]) for l, r in results.items()}
    mult = {2: 1, 3: 1, 4: 1, 5: 1, 6: 1, 7: 1, 8: 5, 13: 1, 14: 1}
    print("best per layer:", {l: (b[0], round(b[1], 1)) for l, b in best.items()})
    print("sum of best over 13 launches: %.1f us" % sum(best[l][1] * mult[l] for l in best))

if __name__ == "__main__":
    main()
