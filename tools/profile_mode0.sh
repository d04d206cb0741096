#!/bin/bash
# PMC counters of the exact-f32 mode's kernels (tools/mode_slots.py f32 5): two SQ passes; summarise with tools/summarize_pmc.py
# or read the per-kernel rows of counter_collection.csv.  Run through gpurun from the repository root.
set -o pipefail
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
S="python3 $R/tools/mode_slots.py f32 5"
rm -rf $R/gpurun_out/prof_m0_sq1 $R/gpurun_out/prof_m0_sq2
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/gpurun_out/prof_m0_sq1 --output-format csv -- $S > $R/gpurun_out/prof_m0_sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY -d $R/gpurun_out/prof_m0_sq2 --output-format csv -- $S > $R/gpurun_out/prof_m0_sq2.log 2>&1
