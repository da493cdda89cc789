#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b26
S=gpurun_out/b26/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/b26/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -4 gpurun_out/b26/pytest.log | tee -a $S
echo "== scene build profile 512" | tee -a $S
tools/profile_scene.sh r03scene 512 2>&1 | tee -a $S
mkdir -p gpurun_out/prof_r03lut
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut/err.txt )
python3 tools/summarize_prof.py gpurun_out/prof_r03lut/t gpurun_out/prof_r03lut/summary.json; rm -rf gpurun_out/prof_r03lut/t
python3 - <<'PY' | tee -a $S
import json
j=json.load(open("gpurun_out/prof_r03lut/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k or "absmax" in k: print(k[:90], v["launches"], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
