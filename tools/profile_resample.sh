#!/bin/bash
# PMC counters of the resampler alone (tools/resample_bench.py hq 48000): two SQ passes + FETCH/WRITE passes.
# Run through gpurun from the repository root; output under gpurun_out/prof_rs_*; summarise with tools/summarize_pmc.py.
set -o pipefail
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
S="python3 $R/tools/resample_bench.py hq 48000"
rm -rf $R/gpurun_out/prof_rs_sq1 $R/gpurun_out/prof_rs_sq2 $R/gpurun_out/prof_rs_fetch $R/gpurun_out/prof_rs_write
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/gpurun_out/prof_rs_sq1 --output-format csv -- $S > $R/gpurun_out/prof_rs_sq1.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY -d $R/gpurun_out/prof_rs_sq2 --output-format csv -- $S > $R/gpurun_out/prof_rs_sq2.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_rs_fetch --output-format csv -- $S > $R/gpurun_out/prof_rs_fetch.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_rs_write --output-format csv -- $S > $R/gpurun_out/prof_rs_write.log 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rs_trace --output-format csv -- $S > $R/gpurun_out/prof_rs_trace.log 2>&1
