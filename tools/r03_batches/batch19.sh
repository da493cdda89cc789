#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b19
S=gpurun_out/b19/summary.txt
cat > /tmp/cases.txt <<'EOC'
reps16 c5 | VRT_LIB=build/ab/r_reps16.so | --config 5 --warmup 5
reps32 c5 | VRT_LIB=build/ab/r_reps32.so | --config 5 --warmup 5
reps64 c5 | VRT_LIB=build/ab/r_reps64.so | --config 5 --warmup 5
reps16 c3 | VRT_LIB=build/ab/r_reps16.so | --config 3 --warmup 5
reps32 c3 | VRT_LIB=build/ab/r_reps32.so | --config 3 --warmup 5
reps64 c3 | VRT_LIB=build/ab/r_reps64.so | --config 3 --warmup 5
reps16 c2 | VRT_LIB=build/ab/r_reps16.so | --config 2 --warmup 5
reps64 c2 | VRT_LIB=build/ab/r_reps64.so | --config 2 --warmup 5
reps4 c2 | VRT_LIB=build/ab/r_reps4.so | --config 2 --warmup 5
reps16 c4 | VRT_LIB=build/ab/r_reps16.so | --config 4 --warmup 5
reps64 c4 | VRT_LIB=build/ab/r_reps64.so | --config 4 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
