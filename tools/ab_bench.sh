#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by ~10 %): buzzdetect_amd/csrc/libA.so and libB.so,
# alternating, per-slot HIP-event times of the kernels named in $1 (egrep pattern on the slot lines).
#   gpurun -- 'bash tools/ab_bench.sh "stem|sep4|sep8"'
pat=${1:-"slot"}
for round in 1 2; do
  for v in A B; do
    BUZZDETECT_HIP_LIB=$PWD/buzzdetect_amd/csrc/lib$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 3 --files-per-step 1 --per-slot --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "$pat|windows/s \(" | sed "s/^/[$v$round] /"
  done
done
