#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b20
S=gpurun_out/b20/summary.txt
cat > /tmp/cases.txt <<'EOC'
r16 p100 c5 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=100 | --config 5 --warmup 5
r16 p115 c5 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=115 | --config 5 --warmup 5
r16 p130 c5 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=130 | --config 5 --warmup 5
r16 p90 c5 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=90 | --config 5 --warmup 5
r16 p75 c5 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=75 | --config 5 --warmup 5
r16 p100 c3 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=100 | --config 3 --warmup 5
r16 p115 c3 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=115 | --config 3 --warmup 5
r16 p130 c3 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=130 | --config 3 --warmup 5
r16 p90 c3 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=90 | --config 3 --warmup 5
r16 p75 c3 | VRT_LIB=build/ab/r_reps16.so VRT_SLICE_PERCENT=75 | --config 3 --warmup 5
r4 p115 c5 | VRT_LIB=build/ab/r_reps4.so | --config 5 --warmup 5
r4 p130 c5 | VRT_LIB=build/ab/r_reps4.so VRT_SLICE_PERCENT=130 | --config 5 --warmup 5
r4 p150 c5 | VRT_LIB=build/ab/r_reps4.so VRT_SLICE_PERCENT=150 | --config 5 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
