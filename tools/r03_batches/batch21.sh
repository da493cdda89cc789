#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b21
S=gpurun_out/b21/summary.txt
cat > /tmp/cases.txt <<'EOC'
c64 c5 | VRT_LIB=build/ab/r_reps16.so | --config 5 --warmup 5
c32 c5 | VRT_LIB=build/ab/c_coarse32.so | --config 5 --warmup 5
c48 c5 | VRT_LIB=build/ab/c_coarse48.so | --config 5 --warmup 5
c128 c5 | VRT_LIB=build/ab/c_coarse128.so | --config 5 --warmup 5
c64 c3 | VRT_LIB=build/ab/r_reps16.so | --config 3 --warmup 5
c32 c3 | VRT_LIB=build/ab/c_coarse32.so | --config 3 --warmup 5
c48 c3 | VRT_LIB=build/ab/c_coarse48.so | --config 3 --warmup 5
c128 c3 | VRT_LIB=build/ab/c_coarse128.so | --config 3 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
