#!/bin/bash
# full-size validation of the final round-3 code + longer randomised differential runs
set -o pipefail
mkdir -p gpurun_out/b24
S=gpurun_out/b24/summary.txt
for c in 4 5; do
  timeout -k 10 1000 python3 bench.py --config $c --cpu-seconds 60 > gpurun_out/b24/full_config$c.json 2> gpurun_out/b24/full_config$c.err; echo "bench config $c full rc $?" | tee -a $S
  tail -1 gpurun_out/b24/full_config$c.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); cb=d['cpu_baseline']
print('  value %.4e | cpu sample: %s | gpu_matches_bit_exact %s' % (d['value'], cb['sample'], cb['gpu_matches_bit_exact']))" | tee -a $S
done
for seed in 61 62 63; do
  timeout -k 10 400 python3 tools/fuzz_gpu.py --seconds 240 --seed $seed 2>&1 | tail -1 | tee -a $S
done
for seed in 64 65 66; do
  timeout -k 10 400 python3 tools/fuzz_gpu.py --seconds 240 --seed $seed --big 2>&1 | tail -1 | tee -a $S
done
