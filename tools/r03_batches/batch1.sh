#!/bin/bash
# round-3 batch 1: parity of the new base, then A/B of build variants
set -o pipefail
mkdir -p gpurun_out/b1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/b1/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/b1/summary.txt
tail -3 gpurun_out/b1/pytest.log | tee -a gpurun_out/b1/summary.txt
CONFIGS="3 4 5" REPS=2 tools/ab.sh build/ab/a_base.so build/ab/d_ldsfield.so 2>&1 | tee -a gpurun_out/b1/summary.txt
CONFIGS="8" REPS=2 tools/ab.sh build/ab/a_base.so build/ab/b_nokeep.so build/ab/c_occ5.so 2>&1 | tee -a gpurun_out/b1/summary.txt
for lds in 12288 16384; do
  BENCH_ARGS="--lds-bytes $lds" CONFIGS="5 3" REPS=1 tools/ab.sh build/ab/e_exp.so 2>&1 | sed "s/^/lds=$lds /" | tee -a gpurun_out/b1/summary.txt
done
