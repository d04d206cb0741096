"""Developer aid: embeddings of the default launch set against one kernel per op (bd_set_fusion 0, 0)."""
import os
import sys

os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")

import numpy as np

from buzzdetect_amd.engine import HipEngine
from oracle import yamnet_oracle as O

HOP = 15360


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    mode = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
    eng = HipEngine(embeddername="yamnet_k2", modelname="model_general_v3")
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=windows)
    eng.set_pointwise_mode(mode)
    eng.set_fusion(False, False)
    ref = eng.embed(x, 0.96).numpy()
    eng.set_fusion(True, True)
    got = eng.embed(x, 0.96).numpy()
    d = np.abs(got - ref)
    print("windows", windows, "mode", mode, "max |d|", d.max(), "of", np.abs(ref).max(), "wrong", int((got != ref).sum()), "of", got.size)
    bad_w = np.nonzero((got != ref).any(axis=1))[0]
    bad_c = np.nonzero((got != ref).any(axis=0))[0]
    print("windows wrong:", bad_w[:40], "...", len(bad_w))
    print("channels wrong:", bad_c[:64], "...", len(bad_c))
    for w in bad_w[:3]:
        c = np.nonzero(got[w] != ref[w])[0][:8]
        print(" w", w, "c", c, "got", got[w, c], "ref", ref[w, c])


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "pattern"):
    main()


def pattern():
    """(window, channel mod 64) of every wrong embedding value, as a small table."""
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    eng = HipEngine(embeddername="yamnet_k2", modelname="model_general_v3")
    x = O.synthetic_audio(HOP * (windows - 1) + 15600, seed=windows)
    if len(sys.argv) > 3:
        eng.set_pointwise_mode(sys.argv[3])
    eng.set_fusion(False, False)
    ref = eng.embed(x, 0.96).numpy()
    eng.set_fusion(True, True)
    got = eng.embed(x, 0.96).numpy()
    bad = got != ref
    for w in range(windows):
        cs = sorted(set(int(c) % 64 for c in np.nonzero(bad[w])[0]))
        print("window", w, "wrong channels mod 64:", cs, " count", int(bad[w].sum()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pattern":
    pattern()
