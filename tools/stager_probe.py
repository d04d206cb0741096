"""How fast reader threads move a file onto the device through bd_stager_read, by piece size, buffers and threads.

    python tools/stager_probe.py [GB=2.5]

Writes one 16-bit PCM-sized file on tmpfs, reads it once (fresh tmpfs pages are slow on their first pass), then for each
(threads, stage MB, buffers): every thread takes 19.2 MB chunks of the file round-robin through its own stager into its own
device buffers.  Also the old way for comparison: whole chunks into chunk-sized pinned buffers, then one copy per chunk."""
import ctypes as C
import os
import sys
import tempfile
import threading
import time

import torch

from buzzdetect_amd import _lib


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    lib = _lib.load()
    chunk = 19_200_000
    n_chunks = int(gb * 1e9 // chunk)
    root = tempfile.mkdtemp(prefix="bd_probe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(root, "blob.bin")
    block = os.urandom(1 << 20)
    with open(path, "wb") as f:
        for _ in range(n_chunks * chunk // len(block) + 1):
            f.write(block)
    with open(path, "rb", buffering=0) as f:
        buf = bytearray(16 << 20)
        while f.readinto(buf):
            pass
    fd = os.open(path, os.O_RDONLY)
    dev = torch.device("cuda", 0)
    torch.empty(1, device=dev)
    try:
        for threads in (1, 2, 4, 6, 8):
            for stage_mb, bufs in ((2, 2), (4, 2), (8, 2), (8, 3), (16, 2), (32, 1)):
                stagers = []
                for _ in range(threads):
                    h = C.c_void_p()
                    _lib.check(lib.bd_stager_create(C.byref(h), 0, stage_mb << 20, bufs))
                    stagers.append(h)
                streams = [torch.cuda.Stream(dev) for _ in range(threads)]
                dst = [[torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(4)] for _ in range(threads)]
                torch.cuda.synchronize()

                def work(t):
                    for k, c in enumerate(range(t, n_chunks, threads)):
                        got = lib.bd_stager_read(stagers[t], fd, c * chunk, chunk, dst[t][k & 3].data_ptr(), streams[t].cuda_stream)
                        assert got == chunk, got

                t0 = time.perf_counter()
                ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"threads {threads}  stage {stage_mb:2d} MB x {bufs}: {n_chunks * chunk / dt / 1e9:6.1f} GB/s", flush=True)
                for h in stagers:
                    lib.bd_stager_destroy(h)
            # the old way: chunk-sized pinned buffers (pre-pinned here), one copy per chunk
            pinned = [[torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(4)] for _ in range(threads)]
            streams = [torch.cuda.Stream(dev) for _ in range(threads)]
            dst = [[torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(4)] for _ in range(threads)]
            evs = [[None] * 4 for _ in range(threads)]
            torch.cuda.synchronize()

            def work_old(t):
                for k, c in enumerate(range(t, n_chunks, threads)):
                    b = k & 3
                    if evs[t][b] is not None:
                        evs[t][b].synchronize()
                    view = memoryview(pinned[t][b].numpy())
                    got = 0
                    while got < chunk:
                        r = os.preadv(fd, [view[got:]], c * chunk + got)
                        if r <= 0:
                            break
                        got += r
                    with torch.cuda.stream(streams[t]):
                        dst[t][b].copy_(pinned[t][b], non_blocking=True)
                        e = evs[t][b] or torch.cuda.Event()
                        e.record(streams[t])
                        evs[t][b] = e

            t0 = time.perf_counter()
            ts = [threading.Thread(target=work_old, args=(t,)) for t in range(threads)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"threads {threads}  whole chunks, 4 pinned buffers each: {n_chunks * chunk / dt / 1e9:6.1f} GB/s", flush=True)
    finally:
        os.close(fd)
        os.remove(path)
        os.rmdir(root)


if __name__ == "__main__":
    main()
