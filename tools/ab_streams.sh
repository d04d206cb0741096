#!/bin/bash
# Same box: the default launch set and --sep-variant $1 with 1, 2 and 3 analyzer streams, two rounds.
for round in 1 2; do
  for st in 1 2 3; do
    for arm in new old; do
      if [ $arm = old ]; then v="--sep-variant $1"; else v=""; fi
      timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events --streams $st $v 2>&1 >/dev/null | grep -E "windows/s \(" | sed "s/^/[$arm streams=$st r$round] /"
    done
  done
done
