#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b4
S=gpurun_out/b4/summary.txt
cat > /tmp/cases.txt <<'EOC'
old c3 s3 | VRT_LIB=build/ab/old_r2.so | --config 3 --schedule 3 --warmup 3
v1 c3 s3 | VRT_LIB=build/ab/v1_fold.so | --config 3 --schedule 3 --warmup 3
v2 c3 s3 | VRT_LIB=build/ab/v2_stride.so | --config 3 --schedule 3 --warmup 3
v3 c3 s3 | VRT_LIB=build/ab/v3_oldcons.so | --config 3 --schedule 3 --warmup 3
old c5 s3 | VRT_LIB=build/ab/old_r2.so | --config 5 --schedule 3 --warmup 3
v1 c5 s3 | VRT_LIB=build/ab/v1_fold.so | --config 5 --schedule 3 --warmup 3
v2 c5 s3 | VRT_LIB=build/ab/v2_stride.so | --config 5 --schedule 3 --warmup 3
v3 c5 s3 | VRT_LIB=build/ab/v3_oldcons.so | --config 5 --schedule 3 --warmup 3
old c4 s3 | VRT_LIB=build/ab/old_r2.so | --config 4 --schedule 3 --warmup 3
v1 c4 s3 | VRT_LIB=build/ab/v1_fold.so | --config 4 --schedule 3 --warmup 3
v2 c4 s3 | VRT_LIB=build/ab/v2_stride.so | --config 4 --schedule 3 --warmup 3
old c3 s2 | VRT_LIB=build/ab/old_r2.so | --config 3 --schedule 2 --warmup 3
v1 c3 s2 | VRT_LIB=build/ab/v1_fold.so | --config 3 --schedule 2 --warmup 3
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
