#!/bin/bash
# Developer loop of the on-chip run: the priority policies of sep_chip_kernel (BD_CHIP_TUNE, developer build libtrace.so only)
# side by side on one box: phase trace of workgroup 0 and the three-stream rate of the whole path.  $1 = log tag, $2.. = tunes.
tag=${1:-x}; shift
tunes=${@:-"0 1 2"}
mkdir -p gpurun_out/r05
export BUZZDETECT_HIP_LIB=$PWD/buzzdetect_amd/csrc/libtrace.so
for t in $tunes; do
  BD_CHIP_TUNE=$t BD_WS_TRACE=7 timeout -k 10 200 python tools/w12_trace.py 1 2>&1 | grep trace | sed "s/^/[tune $t] /"
done > gpurun_out/r05/tune_trace_$tag.log
cat gpurun_out/r05/tune_trace_$tag.log
for round in 1 2; do
  for t in $tunes; do
    BD_CHIP_TUNE=$t timeout -k 10 300 python bench.py --steps 10 --warmup 3 --per-slot --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "sep8|windows/s" | sed "s/^/[tune $t r$round] /"
  done
done > gpurun_out/r05/tune_bench_$tag.log
cat gpurun_out/r05/tune_bench_$tag.log
