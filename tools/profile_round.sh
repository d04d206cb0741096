#!/bin/bash
# Collect the round's profiling artefacts on the GPU box (run through gpurun from the repository root):
#   kernel trace + stats, HBM traffic counters (two passes), SQ busy/wait counters (two passes); then trace + traffic of the
#   exact-f32 mode.
# Output under gpurun_out/prof_*; summarise with tools/summarize_rocprof.py, tools/pmc_traffic.py, tools/summarize_pmc.py.
set -o pipefail
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --files-per-step 1 --no-cpu-baseline --no-kernel-events --no-extras --streams 1"
S="python3 $R/bench.py --steps 2 --warmup 1 --files-per-step 1 --no-cpu-baseline --no-kernel-events --no-extras --streams 1"
F="python3 $R/bench.py --steps 1 --warmup 1 --files-per-step 1 --no-cpu-baseline --no-kernel-events --no-extras --streams 1 --pointwise-mode f32"
rm -rf $R/gpurun_out/prof_trace $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_sq1 $R/gpurun_out/prof_sq2 $R/gpurun_out/prof_fetch_f32 $R/gpurun_out/prof_write_f32 $R/gpurun_out/prof_trace_f32
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace --output-format csv -- $B > $R/gpurun_out/prof_trace.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch --output-format csv -- $S > $R/gpurun_out/prof_fetch.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write --output-format csv -- $S > $R/gpurun_out/prof_write.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof_sq1 --output-format csv -- $S > $R/gpurun_out/prof_sq1.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $R/gpurun_out/prof_sq2 --output-format csv -- $S > $R/gpurun_out/prof_sq2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace_f32 --output-format csv -- $F > $R/gpurun_out/prof_trace_f32.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch_f32 --output-format csv -- $F > $R/gpurun_out/prof_fetch_f32.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write_f32 --output-format csv -- $F > $R/gpurun_out/prof_write_f32.log 2>&1
