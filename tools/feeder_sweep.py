"""analyze() on one generated 24 h 16-bit recording (600 s chunks, hop 1.0) by number of reader threads: audio-seconds per
second of the first and second call per setting.   python tools/feeder_sweep.py [hours=24] [readers ...]"""
import os
import shutil
import sys
import tempfile
import time
import wave

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BUZZDETECT_SYNTHETIC_WEIGHTS", "1")
import bench                                                        # noqa: E402
from buzzdetect_amd.analyze import analyze                          # noqa: E402
from buzzdetect_amd.engine import HipEngine                         # noqa: E402


def main():
    hours = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    readers = [int(a) for a in sys.argv[2:]] or [4, 6, 8, 10, 12]
    root = tempfile.mkdtemp(prefix="bd_sweep_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        audio = os.path.join(root, "audio")
        os.makedirs(audio)
        hour = bench.synthetic_audio(torch.device("cuda", 0), bench.FILE_SAMPLES, 4242)
        block = (hour * 32768.0).round().clamp_(-32768, 32767).to(torch.int16).cpu().numpy().astype("<i2").tobytes()
        with wave.open(os.path.join(audio, "day.wav"), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            for _ in range(hours):
                w.writeframes(block)
        with open(os.path.join(audio, "day.wav"), "rb", buffering=0) as f:
            buf = bytearray(16 << 20)
            while f.readinto(buf):
                pass
        engines = [HipEngine(embeddername="yamnet_k2", modelname="model_general_v3") for _ in range(2)]
        for n in readers + readers[:1]:
            for call in range(2):
                out = os.path.join(root, f"out_{n}_{call}_{time.monotonic_ns()}")
                t0 = time.perf_counter()
                rep = analyze("model_general_v3", classes_out="all", framehop_prop=1.0, chunklength=600.0, dir_audio=audio,
                              dir_out=out, embeddername="yamnet_k2", engines=list(engines), n_streamers=n, rank=0, world_size=1)
                dt = time.perf_counter() - t0
                print(f"readers {n:2d} call {call}: {rep.audio_seconds / dt / 1e6:.3f} M audio-s/s ({dt * 1e3:.1f} ms)  busy "
                      f"{ {k: round(v, 3) for k, v in sorted(rep.busy.items())} }", flush=True)
                shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
