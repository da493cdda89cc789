#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b6
S=gpurun_out/b6/summary.txt
VRT_LIB=$PWD/build/ab/f_face.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/b6/pytest.log 2>&1; echo "pytest (face lib) rc $?" | tee -a $S
tail -4 gpurun_out/b6/pytest.log | tee -a $S
CONFIGS="3 5 4 2" REPS=2 tools/ab.sh build/ab/pre_wavesort.so build/ab/f_noface.so build/ab/f_face.so 2>&1 | tee -a $S
echo "== phases face config 5" | tee -a $S
VRT_LIB=$PWD/build/ab/ph_face.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 tools/phases.py --config 5 2>&1 | tail -12 | tee -a $S
