#!/usr/bin/env python3
"""Developer check on a GPU box: every stage of the HIP path against the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buzzdetect_amd import weights as Wt
from buzzdetect_amd.engine import HipEngine
from oracle import yamnet_oracle as O

def main():
    eng = HipEngine()
    blob = Wt.synthetic_embedder_blob(); mel = Wt.load_mel(); head = Wt.load_head()
    hop, step = 15360, 96
    nwin = 5
    x = O.synthetic_audio(15360 * nwin + 240 - 7)
    lm_o = O.log_mel(O.pad_waveform(x, hop), mel, np.float64)
    lm_g = eng.frontend(x, hop).cpu().numpy()
    print("logmel", lm_g.shape, lm_o.shape, "max|d|", np.abs(lm_g - lm_o).max())
    taps = []
    pat = O.frame_patches(lm_o, step)
    emb_o = O.yamnet_body(pat, blob, np.float64, taps)
    for s, t in enumerate(taps):
        g = eng.stage_tap(x, hop, step, s, nwin).cpu().numpy()
        d = np.abs(g - t).max(); sc = np.abs(t).max()
        print(f"stage {s:2d} shape {g.shape} max|d| {d:.3e} max|ref| {sc:.3e}")
    emb_g, log_g = eng.run(x, hop, step, True, True)
    log_o = O.dense_head(emb_o, head.kernel, head.bias, np.float64)
    print("emb max|d|", np.abs(emb_g.cpu().numpy() - emb_o).max(), "logits max|d|", np.abs(log_g.cpu().numpy() - log_o).max(), "max|logit|", np.abs(log_o).max())
    # halfhop
    hop2, step2 = 7680, 48
    l2 = eng.run(x, hop2, step2, False, True)[1].cpu().numpy()
    o2 = O.predict(x, blob, mel, head.kernel, head.bias, hop2, step2, np.float64)
    print("halfhop", l2.shape, o2.shape, np.abs(l2 - o2).max())
    # timing of a 1024-window batch
    n = 15360 * 1024
    xb = torch.from_numpy(O.synthetic_audio(n)).cuda()
    for g in (1024, 256, 128):
        eng.set_group_windows(g)
        eng.run(xb, hop, step, False, True); torch.cuda.synchronize()
        t = time.time()
        for _ in range(5): eng.run(xb, hop, step, False, True)
        torch.cuda.synchronize(); dt = (time.time() - t) / 5
        print(f"group {g}: {dt*1e3:.2f} ms / 1024 windows -> {1024/dt:.0f} windows/s")
    eng.set_group_windows(1024)
    eng.profile_enable(True)
    for _ in range(3): eng.run(xb, hop, step, False, True)
    ms, cnt = eng.profile_read()
    for i in range(29):
        print(f"slot {i:2d} {ms[i]/max(cnt[i],1):8.4f} ms x{cnt[i]}")
    print("sum per call", ms.sum()/3)

if __name__ == "__main__":
    main()
