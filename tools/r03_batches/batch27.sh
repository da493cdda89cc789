#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b27
S=gpurun_out/b27/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_scene.py -m gpu -x -q 2>&1 | tail -1 | tee -a $S
echo "== scene build profile 512 (rotate for float + short division)" | tee -a $S
tools/profile_scene.sh r03scene 512 2>&1 | tee -a $S
echo "== same with the IEEE division" | tee -a $S
VRT_LIB=$PWD/build/ab/sc_ieee.so tools/profile_scene.sh r03scene_ieee 512 2>&1 | tee -a $S
