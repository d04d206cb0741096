#!/bin/bash
# Developer build of the library with the in-kernel traces compiled in (-DBD_KERNEL_TRACE): buzzdetect_amd/csrc/libtrace.so.
# Never shipped, never loaded unless BUZZDETECT_HIP_LIB points at it (tools/w12_trace.py, tools/chip_tune.sh, tools/power_profile.py).
set -e
cd "$(dirname "$0")/../buzzdetect_amd/csrc"
obj=$(mktemp -d)
for f in engine frontend resample sepf32 cnn rowfmt stemreg stemregf32 l4regf32; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE -c $f.hip -o $obj/$f.o 2>/dev/null &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE -mllvm -amdgpu-sched-strategy=iterative-maxocc -c sepmid.hip -o $obj/sepmid.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE -mllvm -amdgpu-sched-strategy=iterative-maxocc -c sepchipf32.hip -o $obj/sepchipf32.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE -mllvm -amdgpu-sched-strategy=iterative-maxocc -c sepmidf32.hip -o $obj/sepmidf32.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBD_KERNEL_TRACE -mllvm -amdgpu-sched-strategy=iterative-maxocc -c sepchip.hip -o $obj/sepchip.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o libtrace.so $obj/*.o
rm -rf $obj
ls -la libtrace.so
