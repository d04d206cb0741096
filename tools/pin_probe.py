"""What page-locking the feeder's staging buffers costs on the GPU box, by the way it is done.

    python tools/pin_probe.py [slots] [MB per slot]

Times, for `slots` buffers of `MB` each: torch.empty(pin_memory=True) (the caching host allocator rounds sizes up to a
power of two), hipHostMalloc of the exact size, hipHostRegister of anonymous memory (plain, pre-faulted, and with
MADV_HUGEPAGE), each sequentially and from 6 threads at once; then one host-to-device copy out of each kind."""
import ctypes
import mmap
import sys
import threading
import time

import torch


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    mb = float(sys.argv[2]) if len(sys.argv) > 2 else 19.2
    nbytes = int(mb * 1e6)
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    torch.empty(1, device=dev)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostFree.argtypes = [ctypes.c_void_p]
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
    libc = ctypes.CDLL("libc.so.6", use_errno=True)
    libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def torch_pin():
        return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)

    def host_malloc(flags=0):
        p = ctypes.c_void_p()
        rc = hip.hipHostMalloc(ctypes.byref(p), nbytes, flags)
        assert rc == 0, rc
        return p

    def register(prefault=False, huge=False):
        size = (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1)
        m = mmap.mmap(-1, size + (2 << 20))
        addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
        base = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
        if huge:
            libc.madvise(base, size, 14)          # MADV_HUGEPAGE
        if prefault:
            ctypes.memset(base, 0, size)
        rc = hip.hipHostRegister(base, size, 0)
        assert rc == 0, rc
        return m, base

    def timed(label, fn, threads=1):
        keep = []
        lock = threading.Lock()
        per = slots // threads

        def work():
            mine = [fn() for _ in range(per)]
            with lock:
                keep.extend(mine)

        t0 = time.perf_counter()
        ts = [threading.Thread(target=work) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        gb = per * threads * nbytes / 1e9
        print(f"{label:52s} threads={threads}: {dt * 1e3:8.1f} ms for {gb:.2f} GB = {dt / gb:.3f} s/GB", flush=True)
        return keep

    for threads in (1, 6):
        k = timed("torch.empty(pin_memory=True)", torch_pin, threads)
        del k
        torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
        k = timed("hipHostMalloc(exact, default flags)", host_malloc, threads)
        for p in k:
            hip.hipHostFree(p)
        k = timed("hipHostMalloc(exact, hipHostMallocNonCoherent)", lambda: host_malloc(0x80000000), threads)
        for p in k:
            hip.hipHostFree(p)
        k = timed("hipHostRegister(anonymous mmap)", register, threads)
        for m, base in k:
            hip.hipHostUnregister(base)
        del k
        k = timed("hipHostRegister(pre-faulted)", lambda: register(True), threads)
        for m, base in k:
            hip.hipHostUnregister(base)
        del k
        k = timed("hipHostRegister(MADV_HUGEPAGE)", lambda: register(False, True), threads)
        for m, base in k:
            hip.hipHostUnregister(base)
        del k
    # second allocation of the same size out of torch's cache (what a second analyze() call sees)
    k = timed("torch.empty(pin_memory=True), first", torch_pin, 1)
    del k
    k = timed("torch.empty(pin_memory=True), again (cached)", torch_pin, 1)
    # copy rate out of one such buffer
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(3):
        d.copy_(k[0], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in k:
        d.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"H2D out of torch-pinned buffers: {len(k) * nbytes / dt / 1e9:.1f} GB/s")
    try:
        print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
    except OSError:
        pass


if __name__ == "__main__":
    main()
