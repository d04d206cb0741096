#!/bin/bash
# Same-box comparison of several builds of the library at job level: the shipped one ("cur") and every library given, in turn,
# three rounds; bench.py flags in $FLAGS.   gpurun -- 'bash tools/ab_libs.sh buzzdetect_amd/csrc/libv00.so ...'
for round in 1 2 3; do
  for lib in cur "$@"; do
    if [ $lib = cur ]; then unset BUZZDETECT_HIP_LIB; else export BUZZDETECT_HIP_LIB=$lib; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events $FLAGS 2>&1 >/dev/null | grep -E "windows/s \(" | sed "s|^|[$(basename $lib) r$round] |"
  done
done
