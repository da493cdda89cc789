#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b11
S=gpurun_out/b11/summary.txt
cat > /tmp/cases.txt <<'EOC'
c4 6KiB auto | A=1 | --config 4 --warmup 5
c4 10KiB | A=1 | --config 4 --warmup 5 --lds-bytes 10240
c8 6KiB auto | A=1 | --config 8 --warmup 5
c8 10KiB | A=1 | --config 8 --warmup 5 --lds-bytes 10240
c7 auto | A=1 | --config 7 --warmup 5
c7 6KiB | A=1 | --config 7 --warmup 5 --lds-bytes 6144
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
