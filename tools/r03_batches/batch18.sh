#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b18
S=gpurun_out/b18/summary.txt
cat > /tmp/cases.txt <<'EOC'
reps4 c5 | VRT_LIB=build/ab/r_reps4.so | --config 5 --warmup 5
reps8 c5 | VRT_LIB=build/ab/r_reps8.so | --config 5 --warmup 5
reps16 c5 | VRT_LIB=build/ab/r_reps16.so | --config 5 --warmup 5
reps4 c3 | VRT_LIB=build/ab/r_reps4.so | --config 3 --warmup 5
reps8 c3 | VRT_LIB=build/ab/r_reps8.so | --config 3 --warmup 5
reps16 c3 | VRT_LIB=build/ab/r_reps16.so | --config 3 --warmup 5
reps4 c4 | VRT_LIB=build/ab/r_reps4.so | --config 4 --warmup 5
reps8 c4 | VRT_LIB=build/ab/r_reps8.so | --config 4 --warmup 5
EOC
REPS=3 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
