#!/bin/bash
# Same-box A/B of two fusion variants of ONE build: the default path against `--sep-variant $1` (e.g. 7 = layers 8-11 handed
# over through global memory), alternating, per-slot HIP-event times of the slots matching $2 and the three-stream rate.
#   gpurun -- 'bash tools/ab_variant.sh 7 "sep8|sep11"'
v=${1:-7}
pat=${2:-"sep"}
for round in 1 2 3; do
  for arm in default "$v"; do
    extra=""; [ "$arm" != default ] && extra="--sep-variant $arm"
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --per-slot --no-cpu-baseline --no-extras $extra 2>&1 >/dev/null | grep -E "$pat|windows/s" | sed "s/^/[$arm r$round] /"
  done
done
