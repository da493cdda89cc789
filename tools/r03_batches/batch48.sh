#!/bin/bash
# batch 48: soak: march fuzz (small + big cases, multi-engine handles, both ray orders) and scene fuzz on the final code
set -o pipefail
mkdir -p gpurun_out/b48
S=gpurun_out/b48/summary.txt
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 240 --seed 91 2>&1 | tail -1 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 240 --seed 92 --big 2>&1 | tail -1 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --scenes --seconds 240 --seed 93 2>&1 | tail -1 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 240 --seed 94 --big 2>&1 | tail -1 | tee -a $S
