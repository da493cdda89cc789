#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b25
S=gpurun_out/b25/summary.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/b25/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -5 gpurun_out/b25/pytest.log | tee -a $S
for c in 3 5 2 4; do
  timeout -k 10 600 python3 bench.py --config $c --cpu-seconds 0 > gpurun_out/b25/bench_config$c.json 2> gpurun_out/b25/bench_config$c.err; echo "bench config $c rc $?" | tee -a $S
  tail -1 gpurun_out/b25/bench_config$c.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  exact value %.4e ms/step %.3f kernel_ms %.3f | fast_math %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], json.dumps(d.get('fast_math'))[:520]))" | tee -a $S
done
