#!/bin/bash
# One GPU call of the on-chip run's development loop: bit-identity tests, the phase trace of workgroup 0 (developer build
# libtrace.so, BD_WS_TRACE=7), then the same-box A/B against the round-3 form (tools/ab_variant.sh 7).  $1 = tag of the log files.
tag=${1:-x}
mkdir -p gpurun_out/r05
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8_to_11 or fused_separable" > gpurun_out/r05/t_chip$tag.log 2>&1
echo "rc=$?" >> gpurun_out/r05/t_chip$tag.log
tail -3 gpurun_out/r05/t_chip$tag.log
grep -q "rc=0" gpurun_out/r05/t_chip$tag.log || exit 1
BD_WS_TRACE=7 BUZZDETECT_HIP_LIB=$PWD/buzzdetect_amd/csrc/libtrace.so timeout -k 10 300 python tools/w12_trace.py 1 2>&1 | grep trace > gpurun_out/r05/trace_chip$tag.log
cat gpurun_out/r05/trace_chip$tag.log
bash tools/ab_variant.sh ${AB_VARIANT:-7} "sep8|sep11|sep12" > gpurun_out/r05/ab_chip$tag.log 2>&1
tail -12 gpurun_out/r05/ab_chip$tag.log
