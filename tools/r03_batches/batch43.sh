#!/bin/bash
# batch 43: evidence with the final code: the whole GPU suite, bench lines of configs 3 / 2 / 4 / 5
set -o pipefail
mkdir -p gpurun_out/b43
S=gpurun_out/b43/summary.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/b43/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -3 gpurun_out/b43/pytest.log | tee -a $S
for c in 3 2 4 5; do
  timeout -k 10 300 python3 bench.py --config $c > gpurun_out/b43/bench_config$c.json 2> gpurun_out/b43/bench_config$c.err; echo "bench config $c rc $?" | tee -a $S
  tail -1 gpurun_out/b43/bench_config$c.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; cb=d.get('cpu_baseline',{})
print('  value %.4e ms/step %.3f kernel_ms %.3f clock %.3f frac %s sched %s | cpu %.3e (%s thr) match %s | ref %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['clock_ghz'] or 0, r['frac'], r.get('schedule_observed'), cb.get('value',0), cb.get('cores'), cb.get('gpu_matches_bit_exact'), json.dumps(cb.get('reference'))[:300]))" | tee -a $S
done
