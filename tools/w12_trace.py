"""Developer aid: run a few 1024-window passes with a -DBD_KERNEL_TRACE build (BUZZDETECT_HIP_LIB) so that the kernel traces
selected by BD_WS_TRACE / BD_L4_TRACE / BD_STEM_TRACE print.  argv[1] = bd_set_fusion separable code (default 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd.engine import HipEngine, hop_samples, patch_step

code = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = HipEngine(device=0)
eng.set_fusion(True, code)
hop, step = hop_samples(0.96), patch_step(0.96)
x = torch.randn(1024 * hop + 240, device="cuda") * 0.1
out = torch.empty((1024, 13), device="cuda")
for _ in range(12):
    eng.launch([x], hop, step, False, True, out=out)
torch.cuda.synchronize()
