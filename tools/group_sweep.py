#!/usr/bin/env python3
"""Windows per CNN pass (bd_set_group_windows): a whole synthetic 1 h recording (3750 windows in four chunks) through
ONE predict_batch launch set, two engines on two streams, for several pass sizes.

    python tools/group_sweep.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
import torch

import bench
from buzzdetect_amd.engine import HipEngine, hop_samples


def main():
    dev = torch.device("cuda", 0)
    hop = hop_samples(0.96)
    batches = bench.file_batches(hop)
    files = [bench.synthetic_audio(dev, bench.FILE_SAMPLES, 1234 + i) for i in range(3)]
    for group in (512, 1024, 2048, 4096):
        engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3", device=0) for _ in range(2)]
        streams = [torch.cuda.Stream(dev) for _ in engines]
        for e in engines:
            e.set_group_windows(group)

        def run(n):
            for i in range(n):
                j = i % 2
                with torch.cuda.stream(streams[j]):
                    f = files[i % 3]
                    engines[j].predict_batch([f[a:a + m] for a, m in batches], 0.96)

        run(4)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run(40)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"pass of <= {group} windows: {40 * 3750 / dt:,.0f} windows/s", flush=True)
        del engines


if __name__ == "__main__":
    main()
