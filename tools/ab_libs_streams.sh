#!/bin/bash
# Same box, job level: the shipped library ("cur") and every library given, with $STREAMS (default "2 3") analyzer streams, two rounds.
for round in 1 2; do
  for st in ${STREAMS:-2 3}; do
    for lib in cur "$@"; do
      if [ $lib = cur ]; then unset BUZZDETECT_HIP_LIB; else export BUZZDETECT_HIP_LIB=$lib; fi
      timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events --streams $st $FLAGS 2>&1 >/dev/null | grep -E "windows/s \(" | sed "s|^|[$(basename $lib) streams=$st r$round] |"
    done
  done
done
