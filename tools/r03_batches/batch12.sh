#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b12
S=gpurun_out/b12/summary.txt
VRT_LIB=$PWD/build/ab/w_win.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_engine.py tests/test_gpu_scene.py -m gpu -x -q > gpurun_out/b12/pytest.log 2>&1; echo "pytest (lane-window lib) rc $?" | tee -a $S
tail -6 gpurun_out/b12/pytest.log | tee -a $S
CONFIGS="5 8" REPS=2 tools/ab.sh build/ab/w_nowin.so build/ab/w_win.so build/ab/w_win_noshare.so 2>&1 | tee -a $S
echo "== phases lane windows config 5" | tee -a $S
VRT_LIB=$PWD/build/ab/ph_win.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 tools/phases.py --config 5 2>&1 | tail -12 | tee -a $S
