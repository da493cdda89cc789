#!/bin/bash
# round-3 profiles: kernel trace + counter passes for configs 3, 4, 5
set -o pipefail
mkdir -p gpurun_out/b8
for c in 3 4 5; do
  tools/profile.sh r03c$c --config $c > gpurun_out/b8/profile_c$c.log 2>&1
  echo "config $c profiled: $(tail -1 gpurun_out/b8/profile_c$c.log)" | tee -a gpurun_out/b8/summary.txt
done
ls gpurun_out/prof_r03c3 gpurun_out/prof_r03c4 gpurun_out/prof_r03c5 | tee -a gpurun_out/b8/summary.txt
