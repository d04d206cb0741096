#!/bin/bash
# Developer build of the library with extra compiler flags: tools/build_flags.sh libname.so -DFOO=1 ...  (buzzdetect_amd/csrc/libname.so)
set -e
out=$1; shift
cd "$(dirname "$0")/../buzzdetect_amd/csrc"
rm -f "$out"
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
pids=()
for src in *.hip; do
  f=${src%.hip}
  flags=""; case $f in sepchip|sepmid|sepchipf32|sepmidf32) flags="-mllvm -amdgpu-sched-strategy=iterative-maxocc";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" $flags -c "$src" -o "$obj/$f.o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { echo "build_flags.sh: a compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/*.o
ls -la "$out"
