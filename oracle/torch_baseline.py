"""torch-CPU restatement of the analyze hot path — TEST INFRASTRUCTURE ONLY.

Two jobs (only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it):

* an independent second opinion on ``yamnet_oracle`` (different FFT, different conv
  implementation — ``torch.fft`` / ``F.conv2d`` instead of pocketfft / NumPy slicing);
* the CPU baseline timed beside the GPU number.  It is labelled everywhere as
  **"CPU restatement (torch-CPU fp32), not TensorFlow"**: the reference's own TF path cannot
  run here or on the GPU box (no TensorFlow, no embedder weights — SURVEY §8c/§8d).

Follows the same reference lines as the NumPy oracle: embedders/yamnet/features.py:22-108,
embedders/yamnet/yamnet.py:26-106, models/model_general_v3/model.py:18-31, configured to TF
semantics: explicit framing (no centring), periodic Hann, zero-pad 400 -> 512 on the right,
asymmetric SAME padding via ``F.pad``, BatchNorm ``(x-mean)*rsqrt(var+1e-4)+beta``.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import yamnet_oracle as O


class TorchYamnet:
    def __init__(self, blob: np.ndarray, mel: np.ndarray, head_kernel: np.ndarray, head_bias: np.ndarray,
                 dtype=torch.float32):
        self.dtype = dtype
        t = O.split_blob(np.asarray(blob))
        cv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)  # noqa: E731
        self.mel = cv(mel)
        self.window = cv(O.hann_periodic(np.float32 if dtype == torch.float32 else np.float64))
        self.head_kernel = cv(head_kernel)
        self.head_bias = cv(head_bias)

        def bn(k):
            p = f"layer_with_weights-{k}/"
            return cv(t[p + "beta"]), cv(t[p + "moving_mean"]), cv(t[p + "moving_variance"])

        # torch conv weight layout: [Cout, Cin/groups, kh, kw]
        self.conv1 = cv(t["layer_with_weights-0/kernel"]).permute(3, 2, 0, 1).contiguous()
        self.bn1 = bn(1)
        self.layers: List[Dict] = []
        k = 2
        for stride, _ in O.LAYER_DEFS[1:]:
            dw = cv(t[f"layer_with_weights-{k}/depthwise_kernel"]).permute(2, 3, 0, 1).contiguous()   # [C,1,3,3]
            pw = cv(t[f"layer_with_weights-{k + 2}/kernel"]).permute(3, 2, 0, 1).contiguous()        # [Cout,Cin,1,1]
            self.layers.append({"stride": stride, "dw": dw, "bn_dw": bn(k + 1), "pw": pw, "bn_pw": bn(k + 3)})
            k += 4

    # ---- front end ----
    def log_mel(self, wave: np.ndarray, hop: int) -> torch.Tensor:
        x = torch.from_numpy(O.pad_waveform(np.asarray(wave, dtype=np.float32), hop)).to(self.dtype)
        frames = x.unfold(0, O.STFT_WINDOW, O.STFT_HOP) * self.window
        frames = F.pad(frames, (0, O.FFT_LENGTH - O.STFT_WINDOW))
        mag = torch.fft.rfft(frames, n=O.FFT_LENGTH, dim=1).abs()
        return torch.log(mag @ self.mel + O.LOG_OFFSET)

    @staticmethod
    def _same(x: torch.Tensor, stride: int) -> torch.Tensor:
        _, pt, pb = O._same_pad(x.shape[2], 3, stride)
        _, pl, pr = O._same_pad(x.shape[3], 3, stride)
        return F.pad(x, (pl, pr, pt, pb))

    @staticmethod
    def _bn_relu(x, bn):
        beta, mean, var = bn
        inv = torch.rsqrt(var + O.BN_EPS)
        return torch.relu((x - mean[None, :, None, None]) * inv[None, :, None, None] + beta[None, :, None, None])

    def body(self, patches: torch.Tensor) -> torch.Tensor:
        x = patches[:, None, :, :]                                   # NCHW, C = 1
        x = self._bn_relu(F.conv2d(self._same(x, 2), self.conv1, stride=2), self.bn1)
        for L in self.layers:
            x = self._bn_relu(F.conv2d(self._same(x, L["stride"]), L["dw"], stride=L["stride"],
                                       groups=L["dw"].shape[0]), L["bn_dw"])
            x = self._bn_relu(F.conv2d(x, L["pw"]), L["bn_pw"])
        return x.mean(dim=(2, 3))

    @torch.no_grad()
    def predict(self, wave: np.ndarray, hop: int = 15360, step: int = 96, batch: int = 64) -> np.ndarray:
        lm = self.log_mel(wave, hop)
        t = lm.shape[0]
        w = 1 + (t - O.PATCH_FRAMES) // step if t >= O.PATCH_FRAMES else 0
        idx = (torch.arange(w) * step)[:, None] + torch.arange(O.PATCH_FRAMES)[None, :]
        out = []
        for i in range(0, w, batch):
            emb = self.body(lm[idx[i:i + batch]])
            out.append(emb @ self.head_kernel + self.head_bias)
        if not out:
            return np.zeros((0, self.head_bias.numel()), dtype=np.float32)
        return torch.cat(out, 0).numpy()


def usable_cores(cap: int = 32) -> int:
    """Cores this process may really use: affinity mask, cgroup CPU quota, and a cap (a GPU box
    exposes all host cores to ``os.cpu_count()`` but grants only a share of them)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def time_cpu_baseline(model: TorchYamnet, wave: np.ndarray, hop: int, step: int, repeats: int = 3,
                      threads: int = None) -> Dict:
    """windows/s of the torch-CPU restatement on ``wave`` (1 warm-up + median of ``repeats``)."""
    import time
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    w = model.predict(wave, hop, step).shape[0]
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        model.predict(wave, hop, step)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"windows": int(w), "seconds": med, "windows_per_s": w / med, "threads": threads}
