#!/bin/bash
# Developer build of the library with the in-kernel traces compiled in (-DBD_KERNEL_TRACE): buzzdetect_amd/csrc/libtrace.so.
# Never shipped, never loaded unless BUZZDETECT_HIP_LIB points at it (tools/w12_trace.py, tools/chip_tune.sh, tools/power_profile.py).
# A failed compile stops the script with the compiler's message; a stale libtrace.so never survives a failed build.
set -e
cd "$(dirname "$0")/../buzzdetect_amd/csrc"
rm -f libtrace.so
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
pids=()
for src in *.hip; do
  f=${src%.hip}
  flags=""; case $f in sepchip|sepmid|sepchipf32|sepmidf32) flags="-mllvm -amdgpu-sched-strategy=iterative-maxocc";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE $flags -c "$src" -o "$obj/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { echo "build_trace.sh: a compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libtrace.so "$obj"/*.o
ls -la libtrace.so
