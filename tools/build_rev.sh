#!/bin/bash
# Developer build for same-box A/Bs across revisions: the library with ONE source file taken from a git revision,
# everything else from the working tree -> buzzdetect_amd/csrc/libprev.so (never shipped; BUZZDETECT_HIP_LIB selects it).
#   bash tools/build_rev.sh sepf32.hip HEAD~1        (run here, in the container: the GPU box has no git history)
set -e
file=$1; rev=${2:-HEAD~1}
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/buzzdetect_amd/csrc"
obj=$(mktemp -d)
git -C "$root" show "$rev:buzzdetect_amd/csrc/$file" > "$obj/$file"
cp bd_internal.h "$obj/"; mkdir -p "$obj/../../include" 2>/dev/null || true
for f in engine frontend resample sepf32 sepchip sepchipf32 sepmid sepmidf32 stemreg stemregf32 l4regf32 cnn rowfmt; do
  src=$f.hip; [ "$f.hip" = "$file" ] && src="$obj/$file"
  flags=""; case $f in sepchip|sepmid|sepchipf32|sepmidf32) flags="-mllvm -amdgpu-sched-strategy=iterative-maxocc";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I"$root/buzzdetect_amd/csrc" $flags -c "$src" -o "$obj/$f.o" 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o libprev.so "$obj"/*.o
rm -rf "$obj"
ls -la libprev.so
