#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b29
S=gpurun_out/b29/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1 | tee -a $S
for v in main terms; do
  lib=$PWD/volumeraytracer_amd/lib/libvrt_hip.so; [ $v = terms ] && lib=$PWD/build/ab/sc_terms.so
  echo "== $v: scene build 512" | tee -a $S
  VRT_LIB=$lib tools/profile_scene.sh r03scene_$v 512 2>&1 | grep -v "^wrote" | tee -a $S
  mkdir -p gpurun_out/prof_r03lut_$v
  ( cd /tmp && export TMPDIR=/tmp && VRT_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut_$v/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut_$v/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03lut_$v/err.txt )
  python3 tools/summarize_prof.py gpurun_out/prof_r03lut_$v/t gpurun_out/prof_r03lut_$v/summary.json > /dev/null; rm -rf gpurun_out/prof_r03lut_$v/t
  python3 - <<PY | tee -a $S
import json
j=json.load(open("gpurun_out/prof_r03lut_$v/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k: print("$v LUT 1024^3:", k[:70], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
done
