#!/bin/bash
# Same-box A/B of two launch sets of ONE library: bench.py --sep-variant $1 ("old") against the default ("new"), alternating
# three times; prints the rate and the per-slot HIP-event lines matching $2.
#   gpurun -- 'bash tools/ab_variant.sh 11 "slot 2[57]"'
old=$1; pat=${2:-"windows/s"}; flags=$3
for round in 1 2 3; do
  for arm in new old; do
    if [ $arm = old ]; then v="--sep-variant $old"; else v=""; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --per-slot --no-cpu-baseline --no-extras $v $flags 2>&1 >/dev/null | grep -E "$pat|windows/s \(" | sed "s/^/[$arm r$round] /"
  done
done
