#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/b2/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/b2/summary.txt
tail -15 gpurun_out/b2/pytest.log | tee -a gpurun_out/b2/summary.txt
for v in base ldsfield; do
  echo "== phases $v config 5" | tee -a gpurun_out/b2/summary.txt
  VRT_LIB=$PWD/build/ab/ph_$v.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 tools/phases.py --config 5 2>&1 | tail -12 | tee -a gpurun_out/b2/summary.txt
  echo "== stats $v config 5" | tee -a gpurun_out/b2/summary.txt
  VRT_LIB=$PWD/build/ab/st_$v.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 bench.py --config 5 --cpu-seconds 0 --steps 2 --warmup 1 2>&1 | grep -i "debug stats" | tee -a gpurun_out/b2/summary.txt
done
echo "== phases base config 3" | tee -a gpurun_out/b2/summary.txt
VRT_LIB=$PWD/build/ab/ph_base.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 tools/phases.py --config 3 2>&1 | tail -12 | tee -a gpurun_out/b2/summary.txt
CONFIGS="3 4 5" REPS=1 tools/ab.sh volumeraytracer_amd/lib/libvrt_hip.so 2>&1 | tee -a gpurun_out/b2/summary.txt
