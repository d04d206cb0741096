"""Developer aid: layer-3 output of the walking stem (bd_set_fusion stem = 4) against the block stem, where they differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step
from oracle import yamnet_oracle as O
eng = HipEngine(device=0)
hop, step = hop_samples(0.96), patch_step(0.96)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = O.synthetic_audio(hop * (n - 1) + 15600, seed=55)
eng.set_fusion(3, False)
a = eng.stage_tap(x, hop, step, 4, n).cpu().numpy()
eng.set_fusion(4, False)
b = eng.stage_tap(x, hop, step, 4, n).cpu().numpy()
print("shape", a.shape, "equal", np.array_equal(a, b), "max|d|", float(np.abs(a - b).max()))
d = a != b
if d.any():
    print("windows with differences", np.unique(np.nonzero(d)[0])[:20])
    print("rows (of 24) with differences", np.unique(np.nonzero(d)[1]))
    print("cols (of 16) with differences", np.unique(np.nonzero(d)[2]))
    ch = np.unique(np.nonzero(d)[3])
    print("channels with differences", len(ch), ch[:16])
    for r in range(24):
        print("row", r, "differing elements", int(d[:, r].sum()), "max|d|", float(np.abs(a[:, r] - b[:, r]).max()))
