#!/bin/bash
# Same-box A/B of two builds of the library (the shipped one against $1, e.g. buzzdetect_amd/csrc/libprev.so from
# tools/build_rev.sh), alternating three times, with bench.py flags $2: per-slot HIP-event times matching $3 and the rate.
#   gpurun -- 'bash tools/ab_lib.sh buzzdetect_amd/csrc/libprev.so "--pointwise-mode f32" "slot  7"'
lib=$1; flags=$2; pat=${3:-"windows/s"}
for round in 1 2 3; do
  for arm in new prev; do
    if [ $arm = prev ]; then export BUZZDETECT_HIP_LIB=$lib; else unset BUZZDETECT_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --per-slot --no-cpu-baseline --no-extras $flags 2>&1 >/dev/null | grep -E "$pat|windows/s \(" | sed "s/^/[$arm r$round] /"
  done
done
