#!/bin/bash
# batch 38: at which granularity does the row alignment matter?  rows of 516 (32 B int16 / 64 B float), 520 (64 / 128), 528 (128 / 256), 544 (256 / 512) voxels
set -o pipefail
mkdir -p gpurun_out/b38
S=$GRAFT_REPO_ROOT/gpurun_out/b38/summary.txt
for n in 518 522 530 546 514 520; do
  echo "== N $n" | tee -a $S
  tools/profile_scene.sh b38_$n $n 2>&1 | grep -v "^wrote" | tee -a $S
done
