#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b15
S=gpurun_out/b15/summary.txt
for l in s_maxilp s_iterilp s_memclause; do
  [ -f build/ab/$l.so ] || continue
  VRT_LIB=$PWD/build/ab/$l.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/$l parity: /" | tee -a $S
done
LIBS="volumeraytracer_amd/lib/libvrt_hip.so"
for l in s_maxilp s_iterilp s_memclause; do [ -f build/ab/$l.so ] && LIBS="$LIBS build/ab/$l.so"; done
CONFIGS="3 5 4" REPS=2 tools/ab.sh $LIBS 2>&1 | tee -a $S
