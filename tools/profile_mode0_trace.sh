#!/bin/bash
# rocprofv3 --kernel-trace --stats of the exact-f32 mode (tools/mode_slots.py f32 5): per-kernel average durations.
# Run through gpurun from the repository root; the stats csv lands under gpurun_out/prof_m0_trace.
set -o pipefail
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m0_trace
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m0_trace --output-format csv -- python3 $R/tools/mode_slots.py f32 5 > $R/gpurun_out/prof_m0_trace.log 2>&1
