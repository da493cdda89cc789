#!/bin/bash
# tools/ab.sh -- A/B of engine builds on one GPU box (run through gpurun): kernel ms per bench config.
#   usage: CONFIGS="3 4 5" REPS=2 tools/ab.sh path/to/libA.so path/to/libB.so ...
# Builds are selected through VRT_LIB (volumeraytracer_amd/capi.py); interleaved so that box drift hits all alike.
CONFIGS=${CONFIGS:-"3 4 5"}
REPS=${REPS:-2}
for rep in $(seq $REPS); do
for lib in "$@"; do
  for c in $CONFIGS; do
    VRT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --config $c --cpu-seconds 0 --steps 5 --warmup 5 $BENCH_ARGS 2>>gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'cfg', $c, 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'step_ms %.3f'%d['ms_per_step'], 'value %.4e'%d['value'], flush=True)" || exit 1
  done
done
done
