"""Developer probe: why is the FIRST pass over a freshly written tmpfs file slower than the second (analyze()'s first call)?
Writes a file to /dev/shm, then reads it twice with 6 threads (os.preadv into pinned / pageable buffers), prints GB/s per pass."""
import os, sys, time, threading
import numpy as np
import torch

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
path = "/dev/shm/bd_read_probe.bin"
chunk = 32 << 20
n_chunks = int(GB * (1 << 30)) // chunk
block = np.random.default_rng(0).integers(0, 255, chunk, dtype=np.uint8).tobytes()
t0 = time.perf_counter()
with open(path, "wb") as f:
    for _ in range(n_chunks):
        f.write(block)
print(f"write {n_chunks * chunk / 1e9:.2f} GB in {time.perf_counter() - t0:.2f} s", flush=True)

def run(bufs, label):
    fd = os.open(path, os.O_RDONLY)
    nxt = [0]
    lock = threading.Lock()
    def work(b):
        view = memoryview(b.numpy())
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= n_chunks:
                return
            got = 0
            while got < chunk:
                r = os.preadv(fd, [view[got:]], i * chunk + got)
                if r <= 0: break
                got += r
    ts = [threading.Thread(target=work, args=(b,)) for b in bufs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    os.close(fd)
    print(f"{label}: {n_chunks * chunk / dt / 1e9:.1f} GB/s ({dt:.3f} s)", flush=True)

pageable = [torch.empty(chunk, dtype=torch.uint8) for _ in range(6)]
for b in pageable: b.zero_()
run(pageable, "pass 1 pageable (touched) buffers")
run(pageable, "pass 2 pageable")
t0 = time.perf_counter()
pinned = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(6)]
print(f"pin 6 x 32 MB: {time.perf_counter() - t0:.3f} s", flush=True)
run(pinned, "pass 3 pinned fresh")
run(pinned, "pass 4 pinned")
many = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(24)]
fd = os.open(path, os.O_RDONLY)
os.close(fd)
run(many[:6], "pass 5 other fresh pinned")
run(many[6:12], "pass 6 other fresh pinned")
os.remove(path)
# a second, freshly written file: is the first pass over NEW file pages the slow one?
with open(path, "wb") as f:
    for _ in range(n_chunks):
        f.write(block)
run(pinned, "new file pass 1 (used pinned buffers)")
run(pinned, "new file pass 2")
os.remove(path)
