#!/bin/bash
# Same-box A/B of two sets of bench.py flags on ONE build, alternating three times: per-slot HIP-event times of the slots matching
# $3 and the three-stream rate.    gpurun -- 'bash tools/ab_flags.sh "" "--stem 4" "stem|windows/s"'
a=$1; b=$2; pat=${3:-"windows/s"}
for round in 1 2 3; do
  for arm in A B; do
    flags=$a; [ $arm = B ] && flags=$b
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --per-slot --no-cpu-baseline --no-extras $flags 2>&1 >/dev/null | grep -E "$pat|windows/s \(" | sed "s/^/[$arm:$flags r$round] /"
  done
done
