#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b32
S=gpurun_out/b32/summary.txt
for seed in 71 72; do timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 240 --seed $seed 2>&1 | tail -1 | tee -a $S; done
for seed in 73 74; do timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 240 --seed $seed --big 2>&1 | tail -1 | tee -a $S; done
