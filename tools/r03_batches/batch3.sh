#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b3
S=gpurun_out/b3/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_engine.py -m gpu -x -q > gpurun_out/b3/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -8 gpurun_out/b3/pytest.log | tee -a $S
echo "== schedule A/B old vs new (warmup 5)" | tee -a $S
cat > /tmp/cases.txt <<'EOC'
old c3 s2 | VRT_LIB=build/ab/old_r2.so | --config 3 --schedule 2 --warmup 5
new c3 s2 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 3 --schedule 2 --warmup 5
old c3 s3 | VRT_LIB=build/ab/old_r2.so | --config 3 --schedule 3 --warmup 5
new c3 s3 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 3 --schedule 3 --warmup 5
old c4 s2 | VRT_LIB=build/ab/old_r2.so | --config 4 --schedule 2 --warmup 5
new c4 s2 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 4 --schedule 2 --warmup 5
old c4 s3 | VRT_LIB=build/ab/old_r2.so | --config 4 --schedule 3 --warmup 5
new c4 s3 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 4 --schedule 3 --warmup 5
old c5 s2 | VRT_LIB=build/ab/old_r2.so | --config 5 --schedule 2 --warmup 5
new c5 s2 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 5 --schedule 2 --warmup 5
old c5 s3 | VRT_LIB=build/ab/old_r2.so | --config 5 --schedule 3 --warmup 5
new c5 s3 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 5 --schedule 3 --warmup 5
old c3 auto | VRT_LIB=build/ab/old_r2.so | --config 3 --warmup 5
new c3 auto | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 3 --warmup 5
old c4 auto | VRT_LIB=build/ab/old_r2.so | --config 4 --warmup 5
new c4 auto | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 4 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
echo "== scene build profile 512 tiled" | tee -a $S
tools/profile_scene.sh r3scene_tiled 512 2>&1 | tee -a $S
echo "== scene build profile 512 pervoxel" | tee -a $S
VRT_SCENE_BUILD=pervoxel tools/profile_scene.sh r3scene_pervoxel 512 2>&1 | tee -a $S
for mode in tiled pervoxel; do
  echo "== config 5 scene build (1024^3 uint8 + LUT) $mode" | tee -a $S
  mkdir -p gpurun_out/prof_r3lut_$mode
  ( cd /tmp && export TMPDIR=/tmp && VRT_SCENE_BUILD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r3lut_$mode/err.txt )
  python3 tools/summarize_prof.py gpurun_out/prof_r3lut_$mode/t gpurun_out/prof_r3lut_$mode/summary.json; rm -rf gpurun_out/prof_r3lut_$mode/t
  python3 - <<PY | tee -a $S
import json
j=json.load(open("gpurun_out/prof_r3lut_$mode/summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k or "absmax" in k: print(k[:90], v["launches"], "avg_ms %.3f" % (v["avg_ns"]/1e6))
PY
done
