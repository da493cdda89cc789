#!/bin/bash
# batch 37: does the row alignment of the field bind the scene build?  N = 512 (rows of 510 voxels), 514 (512: aligned), 516 (514)
set -o pipefail
mkdir -p gpurun_out/b37
S=$GRAFT_REPO_ROOT/gpurun_out/b37/summary.txt
for rep in 1 2; do
for n in 512 514 516; do
  echo "== N $n (rep $rep)" | tee -a $S
  tools/profile_scene.sh b37_$n $n 2>&1 | grep -v "^wrote" | tee -a $S
done
done
