R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 90 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmcv9_$tag --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --no-kernel-events --sep-variant 9 > $R/gpurun_out/pmc_$tag.log 2>&1; }
run a GRBM_GUI_ACTIVE TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES && run b TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS && run c TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ && run d TCP_TCC_READ_REQ_LATENCY TCP_READ_TAGCONFLICT_STALL_CYCLES && run e TCC_HIT TCC_MISS && run f TCC_REQ TCC_TAG_STALL
