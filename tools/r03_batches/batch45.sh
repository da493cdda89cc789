#!/bin/bash
# batch 45: refresh the scene-build and config-5 profiles with the final code
set -o pipefail
mkdir -p gpurun_out/b45
REPO=$GRAFT_REPO_ROOT
S=$REPO/gpurun_out/b45/summary.txt
tools/profile_scene.sh r03scene 512 2>&1 | grep -v "^wrote" | tee -a $S
O=$REPO/gpurun_out/prof_r03lut
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 $REPO/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $O/bench.json 2> $O/err.txt ) || echo "lut trace failed" | tee -a $S
python3 tools/summarize_prof.py $O/t $O/summary.json | tee -a $S; rm -rf $O/t
tools/profile.sh r03c5 --config 5 > gpurun_out/b45/profile_c5.log 2>&1
echo "config 5 profiled: $(tail -1 gpurun_out/b45/profile_c5.log)" | tee -a $S
