#!/usr/bin/env python3
"""Host-to-device copy rate from pinned memory: one copy stream vs the same bytes split over two / four streams.

    python tools/h2d_bench.py
"""
import time

import torch


def main():
    dev = torch.device("cuda", 0)
    nbytes = 31_457_280                      # one 1024-window batch of 16-bit PCM
    host = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
    for h in host:
        h.random_(0, 255)
    for parts in (1, 2, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        dst = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(4)]
        piece = nbytes // parts

        def run(n):
            for i in range(n):
                h, d = host[i % 4], dst[i % 4]
                for p, s in enumerate(streams):
                    with torch.cuda.stream(s):
                        d[p * piece:(p + 1) * piece].copy_(h[p * piece:(p + 1) * piece], non_blocking=True)

        run(4)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{parts} stream(s): {64 * nbytes / dt / 1e9:.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
