#!/bin/bash
# tools/ab_env.sh -- A/B of run-time knobs on one GPU box: each line of stdin is "label | env assignments | bench args";
# prints kernel ms per line, REPS times interleaved.   usage: REPS=2 tools/ab_env.sh < cases.txt
REPS=${REPS:-2}
mapfile -t CASES
for rep in $(seq $REPS); do
  for line in "${CASES[@]}"; do
    [ -z "$line" ] && continue
    label=$(echo "$line" | cut -d'|' -f1); envs=$(echo "$line" | cut -d'|' -f2); args=$(echo "$line" | cut -d'|' -f3)
    env $envs timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 5 --warmup 5 $args 2>>gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'step_ms %.3f'%d['ms_per_step'], 'value %.4e'%d['value'], flush=True)" || echo "$label FAILED"
  done
done
