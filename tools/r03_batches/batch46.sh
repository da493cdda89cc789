#!/bin/bash
# batch 46: full-size check of config 5 on the final code: GPU scene build + march against the CPU port's, every ray
set -o pipefail
mkdir -p gpurun_out/b46
S=gpurun_out/b46/summary.txt
( while true; do sleep 60; echo "alive $(date +%s)" >> gpurun_out/b46/alive.txt; done ) &
KA=$!
timeout -k 10 1000 python3 bench.py --config 5 --cpu-seconds 60 > gpurun_out/b46/bench5.json 2> gpurun_out/b46/bench5.err; echo "bench config 5 rc $?" | tee -a $S
kill $KA
tail -1 gpurun_out/b46/bench5.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); cb=d.get('cpu_baseline',{})
print('value %.4e' % d['value'], 'cpu', {k:cb.get(k) for k in ('value','cores','sample','gpu_matches_bit_exact','rays_checked')})" | tee -a $S
