#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b7
S=gpurun_out/b7/summary.txt
VRT_LIB=$PWD/build/ab/g_prefetch.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scene.py -m gpu -x -q > gpurun_out/b7/pytest.log 2>&1; echo "pytest (prefetch lib, incl. new scene build) rc $?" | tee -a $S
tail -4 gpurun_out/b7/pytest.log | tee -a $S
CONFIGS="5 3" REPS=2 tools/ab.sh build/ab/f_noface.so build/ab/g_prefetch.so 2>&1 | tee -a $S
echo "== phases prefetch config 5" | tee -a $S
VRT_LIB=$PWD/build/ab/ph_prefetch.so VRT_DEBUG_STATS=1 timeout -k 10 300 python3 tools/phases.py --config 5 2>&1 | tail -12 | tee -a $S
echo "== scene build profile 512 tiled (depth-4 prefetch)" | tee -a $S
VRT_LIB=$PWD/build/ab/g_prefetch.so tools/profile_scene.sh r3scene_tiled 512 2>&1 | tee -a $S
mode=tiled
mkdir -p gpurun_out/prof_r3lut_$mode
( cd /tmp && export TMPDIR=/tmp && VRT_LIB=$GRAFT_REPO_ROOT/build/ab/g_prefetch.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/err.txt )
python3 tools/summarize_prof.py gpurun_out/prof_r3lut_$mode/t gpurun_out/prof_r3lut_$mode/summary.json; rm -rf gpurun_out/prof_r3lut_$mode/t
python3 - <<PY | tee -a $S
import json
j=json.load(open("gpurun_out/prof_r3lut_$mode/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k or "absmax" in k: print(k[:90], v["launches"], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
