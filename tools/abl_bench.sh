#!/bin/bash
# Same-box comparison of developer builds buzzdetect_amd/csrc/libabl_<n>.so (n from "$2", default "0 1 2 3 4 5"): per-slot HIP-event times
# of the kernels matching the egrep pattern $1, two rounds, alternating.  The ablated builds give WRONG results: timing only.
pat=${1:-"slot"}
for round in 1 2; do
  for n in ${2:-0 1 2 3 4 5}; do
    BUZZDETECT_HIP_LIB=$PWD/buzzdetect_amd/csrc/libabl_$n.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --files-per-step 1 --streams 1 --per-slot --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "$pat|windows/s \(" | sed "s/^/[abl $n, round $round] /"
  done
done
