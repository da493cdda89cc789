#!/bin/bash
# batch 34: scene build, fixed-point family: running partials (sc_part), + branch-free division (sc_part2) vs HEAD
set -o pipefail
mkdir -p gpurun_out/b34
S=gpurun_out/b34/summary.txt
VRT_LIB=$PWD/build/ab/sc_part2.so timeout -k 10 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee -a $S
for rep in 1 2; do
for v in main sc_part sc_part2; do
  lib=$PWD/volumeraytracer_amd/lib/libvrt_hip.so; [ $v != main ] && lib=$PWD/build/ab/$v.so
  echo "== $v: scene build 512 (rep $rep)" | tee -a $S
  VRT_LIB=$lib tools/profile_scene.sh b34_$v 512 2>&1 | grep -v "^wrote" | tee -a $S
  mkdir -p gpurun_out/prof_b34lut_$v
  ( cd /tmp && export TMPDIR=/tmp && VRT_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b34lut_$v/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_b34lut_$v/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_b34lut_$v/err.txt )
  python3 tools/summarize_prof.py gpurun_out/prof_b34lut_$v/t gpurun_out/prof_b34lut_$v/summary.json > /dev/null; rm -rf gpurun_out/prof_b34lut_$v/t
  python3 - <<PY | tee -a $S
import json
j=json.load(open("gpurun_out/prof_b34lut_$v/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k: print("$v LUT 1024^3:", k[:70], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
done
done
