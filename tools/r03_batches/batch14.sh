#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b14
S=gpurun_out/b14/summary.txt
for seed in 31 32; do
  timeout -k 10 400 python3 tools/fuzz_gpu.py --seconds 200 --seed $seed 2>&1 | tail -3 | tee -a $S
done
for seed in 33 34; do
  timeout -k 10 400 python3 tools/fuzz_gpu.py --seconds 200 --seed $seed --big 2>&1 | tail -3 | tee -a $S
done
echo "== rehearse 2 ranks (final code)" | tee -a $S
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --rehearse --config 2 --steps 3 --warmup 2 > gpurun_out/b14/rehearse.json 2> gpurun_out/b14/rehearse.err; echo "rehearse rc $?" | tee -a $S
tail -1 gpurun_out/b14/rehearse.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gathered_matches_oracle', d.get('gathered_matches_oracle'), d.get('gathered_check'))" | tee -a $S
mkdir -p gpurun_out/prof_r03order
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --cpu-seconds 0 --steps 4 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/err.txt )
python3 tools/summarize_prof.py gpurun_out/prof_r03order/t gpurun_out/prof_r03order/trace_summary.json; rm -rf gpurun_out/prof_r03order/t
