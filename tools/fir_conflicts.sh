#!/bin/bash
# Developer probe: which LDS access of fir_mfma_kernel carries the bank conflicts?  The shipped build and
# three ablated ones (libabl1.so: no fragment reads, 2: no staging writes, 3: no partial-tile writes; -DBD_FIR_ABLATE), each
# under rocprofv3 --pmc on tools/resample_bench.py hq 48000.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  lib=$R/buzzdetect_amd/csrc/libabl$v.so; [ $v = 0 ] && lib=$R/buzzdetect_amd/csrc/libbuzzdetect_hip.so
  rm -rf $R/gpurun_out/prof_fir_$v
  BUZZDETECT_HIP_LIB=$lib timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_fir_$v --output-format csv -- python3 $R/tools/resample_bench.py hq 48000 > $R/gpurun_out/prof_fir_$v.log 2>&1 || exit 1
done
