"""Where the host time of one predict() goes (run on the GPU box): enqueue-only wall time per call, pieces timed alone."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")      # developer tool: timing on the seeded stand-in weights
from buzzdetect_amd import _lib
from buzzdetect_amd.engine import HipEngine, LaunchVerdict, hop_samples, patch_step


def per_call(fn, n=200):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n
    return 1e6 * host, 1e6 * total


def main():
    eng = HipEngine()
    dev = eng.device
    hop, step = hop_samples(0.96), patch_step(0.96)
    for windows in (1024, 64):
        x = torch.randn(hop * windows, device=dev) * 0.1
        out = torch.empty((windows, eng.n_classes), device=dev)
        print(f"--- {windows} windows per call")
        print("predict()                       host %7.1f us   incl. device %7.1f us" % per_call(lambda: eng.predict(x, 0.96, out=out)))
        print("launch(), no verdict            host %7.1f us   incl. device %7.1f us" % per_call(lambda: eng.launch([x], hop, step, False, True, out=out)))
        v = LaunchVerdict(torch.cuda.current_stream())
        print("launch(), one reused verdict    host %7.1f us   incl. device %7.1f us" % per_call(lambda: eng.launch([x], hop, step, False, True, out=out, verdict=v)))
        lib = _lib.load()
        ws_bytes = lib.bd_workspace_bytes(eng._handle, x.numel(), hop, step)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        ptrs = (C.c_void_p * 1)(x.data_ptr())
        lens = (C.c_int64 * 1)(x.numel())
        print("bd_predict (C)                  host %7.1f us   incl. device %7.1f us" % per_call(
            lambda: lib.bd_predict(eng._handle, x.data_ptr(), x.numel(), hop, step, ws.data_ptr(), ws.numel(), None, out.data_ptr(), s)))
        print("bd_predict_chunks (C), no word  host %7.1f us   incl. device %7.1f us" % per_call(
            lambda: lib.bd_predict_chunks(eng._handle, ptrs, lens, 1, hop, step, ws.data_ptr(), ws.numel(), None, out.data_ptr(), -1, None, s)))
        print("bd_predict_chunks (C), word     host %7.1f us   incl. device %7.1f us" % per_call(
            lambda: lib.bd_predict_chunks(eng._handle, ptrs, lens, 1, hop, step, ws.data_ptr(), ws.numel(), None, out.data_ptr(), -1, v.word.data_ptr(), s)))
    print("LaunchVerdict()                 host %7.1f us" % per_call(lambda: LaunchVerdict(torch.cuda.current_stream()))[0])
    print("torch.cuda.Event()+record       host %7.1f us" % per_call(lambda: torch.cuda.Event().record())[0])


if __name__ == "__main__":
    main()


def two_streams():
    """Two engines on two streams, 1024-window calls dealt round-robin (bench.py's main region), variant by variant."""
    engs = [HipEngine(), HipEngine()]
    dev = engs[0].device
    hop, step = hop_samples(0.96), patch_step(0.96)
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
    x = torch.randn(hop * 1024, device=dev) * 0.1
    outs = [torch.empty((1024, engs[0].n_classes), device=dev) for _ in range(2)]
    vs = [LaunchVerdict(s) for s in streams]

    def loop(fn, n=60):
        for i in range(4):
            with torch.cuda.stream(streams[i % 2]):
                fn(i % 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % 2]):
                fn(i % 2)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n

    print("--- two streams, us per 1024-window call")
    print("launch, no verdict      %7.1f" % loop(lambda j: engs[j].launch([x], hop, step, False, True, out=outs[j])))
    print("launch, reused verdict  %7.1f" % loop(lambda j: engs[j].launch([x], hop, step, False, True, out=outs[j], verdict=vs[j])))
    print("predict()               %7.1f" % loop(lambda j: engs[j].predict(x, 0.96, out=outs[j])))
    keep = []
    print("predict(), results kept %7.1f" % loop(lambda j: keep.append(engs[j].predict(x, 0.96, out=outs[j]))))
    print("launch, no verdict      %7.1f" % loop(lambda j: engs[j].launch([x], hop, step, False, True, out=outs[j])))


if __name__ == "__main__" and "--two" in sys.argv:
    two_streams()
