#!/bin/bash
# Same-box A/B of two whole TREES (this working tree against an exported older one, e.g. tools/_r05 = `git archive <rev>` built
# in place): `python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2` alternating three times, each in its own tree.
#   gpurun -- 'bash tools/ab_round.sh tools/_r05 "--pointwise-mode f32"'
old=$1; flags=$2
for round in 1 2 3; do
  for arm in new old; do
    dir=.; [ $arm = old ] && dir=$old
    (cd $dir && timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $flags 2>&1 >/dev/null | grep -E "windows/s \(" | sed "s/^/[$arm r$round] /")
  done
done
