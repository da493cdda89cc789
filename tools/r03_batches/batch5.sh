#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b5
S=gpurun_out/b5/summary.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/b5/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -22 gpurun_out/b5/pytest.log | tee -a $S
echo "== default bench (config 3, cpu baseline + reference leg)" | tee -a $S
timeout -k 10 600 python3 bench.py > gpurun_out/b5/bench3.json 2> gpurun_out/b5/bench3.err; echo "bench rc $?" | tee -a $S
python3 - <<'PY' | tee -a $S
import json
d=json.loads(open("gpurun_out/b5/bench3.json").read().strip().splitlines()[-1])
print("value %.4e ms_per_step %.3f kernel_ms %.3f frac %s sched %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("schedule_observed")))
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:900])
PY
echo "== rehearse: 2 ranks on one GPU over gloo (N>1 code path incl. self-check)" | tee -a $S
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse --config 2 --steps 3 --warmup 2 > gpurun_out/b5/rehearse.json 2> gpurun_out/b5/rehearse.err; echo "rehearse rc $?" | tee -a $S
tail -1 gpurun_out/b5/rehearse.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gathered_matches_oracle', d.get('gathered_matches_oracle'), d.get('gathered_check')); print(json.dumps(d.get('ranks'))[:600])" | tee -a $S
echo "== wave sort A/B" | tee -a $S
CONFIGS="3 5" REPS=2 tools/ab.sh build/ab/pre_wavesort.so volumeraytracer_amd/lib/libvrt_hip.so 2>&1 | tee -a $S
echo "== scene build profile 512 tiled" | tee -a $S
tools/profile_scene.sh r3scene_tiled 512 2>&1 | tee -a $S
mode=tiled
mkdir -p gpurun_out/prof_r3lut_$mode
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/err.txt )
python3 tools/summarize_prof.py gpurun_out/prof_r3lut_$mode/t gpurun_out/prof_r3lut_$mode/summary.json; rm -rf gpurun_out/prof_r3lut_$mode/t
python3 - <<PY | tee -a $S
import json
j=json.load(open("gpurun_out/prof_r3lut_$mode/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k or "absmax" in k or "wave_sort" in k or "wave_work" in k: print(k[:90], v["launches"], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
