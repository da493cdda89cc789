#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b23
S=gpurun_out/b23/summary.txt
VRT_LIB=$PWD/build/ab/t_s640.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -1 | tee -a $S
cat > /tmp/cases.txt <<'EOC'
base c5 | VRT_LIB=build/ab/t_base.so | --config 5 --warmup 5
s400 c5 | VRT_LIB=build/ab/t_s400.so | --config 5 --warmup 5
s640 c5 | VRT_LIB=build/ab/t_s640.so | --config 5 --warmup 5
s1000 c5 | VRT_LIB=build/ab/t_s1000.so | --config 5 --warmup 5
s640w4 c5 | VRT_LIB=build/ab/t_s640w4.so | --config 5 --warmup 5
base c3 | VRT_LIB=build/ab/t_base.so | --config 3 --warmup 5
s400 c3 | VRT_LIB=build/ab/t_s400.so | --config 3 --warmup 5
s640 c3 | VRT_LIB=build/ab/t_s640.so | --config 3 --warmup 5
s1000 c3 | VRT_LIB=build/ab/t_s1000.so | --config 3 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
