#!/bin/bash
# Developer build for same-box A/Bs across revisions: the library with ONE (or several, space separated) source file taken
# from a git revision, everything else from the working tree -> buzzdetect_amd/csrc/libprev.so (never shipped;
# BUZZDETECT_HIP_LIB selects it).
#   bash tools/build_rev.sh sepf32.hip HEAD~1        (run here, in the container: the GPU box has no git history)
#   bash tools/build_rev.sh "sepmid.hip sepchip.hip" HEAD
# A failed compile stops the script with the compiler's message; a stale libprev.so never survives a failed build.
set -e
files=$1; rev=${2:-HEAD~1}
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/buzzdetect_amd/csrc"
rm -f libprev.so
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
for file in $files; do git -C "$root" show "$rev:buzzdetect_amd/csrc/$file" > "$obj/$file"; done
cp bd_internal.h "$obj/"
pids=()
for src in *.hip; do
  f=${src%.hip}
  for file in $files; do [ "$src" = "$file" ] && src="$obj/$file"; done
  flags=""; case $f in sepchip|sepmid|sepchipf32|sepmidf32) flags="-mllvm -amdgpu-sched-strategy=iterative-maxocc";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I"$root/buzzdetect_amd/csrc" $flags -c "$src" -o "$obj/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { echo "build_rev.sh: a compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libprev.so "$obj"/*.o
ls -la libprev.so
