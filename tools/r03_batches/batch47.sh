#!/bin/bash
# batch 47: scene-build fuzz (new mode of tools/fuzz_gpu.py) + march fuzz on the final code
set -o pipefail
mkdir -p gpurun_out/b47
S=gpurun_out/b47/summary.txt
timeout -k 10 300 python3 tools/fuzz_gpu.py --scenes --seconds 200 --seed 81 2>&1 | tail -4 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --scenes --seconds 200 --seed 82 2>&1 | tail -4 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 200 --seed 83 2>&1 | tail -2 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 200 --seed 84 --big 2>&1 | tail -2 | tee -a $S
