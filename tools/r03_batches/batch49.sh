#!/bin/bash
# batch 49: rocprofv3 trace + counter passes of configs 3 and 4 on the final code (kernel signature with the FAST flag)
set -o pipefail
mkdir -p gpurun_out/b49
S=gpurun_out/b49/summary.txt
for c in 3 4; do
  tools/profile.sh r03c$c --config $c > gpurun_out/b49/profile_c$c.log 2>&1
  echo "config $c profiled: $(tail -1 gpurun_out/b49/profile_c$c.log)" | tee -a $S
done
