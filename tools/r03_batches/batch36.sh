#!/bin/bash
# batch 36: scene build with every load unconditional (sc_u1), + loads two planes ahead (sc_u2), quads on/off, vs sc_part
set -o pipefail
mkdir -p gpurun_out/b36
REPO=$GRAFT_REPO_ROOT
S=$REPO/gpurun_out/b36/summary.txt
VRT_LIB=$REPO/build/ab/sc_u2.so timeout -k 10 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee -a $S
VRT_LIB=$REPO/build/ab/sc_u1.so VRT_SCENE_QUADS=0 timeout -k 10 600 python -m pytest tests/test_gpu_scene.py -m gpu -x -q 2>&1 | tail -2 | tee -a $S
for rep in 1 2; do
for v in sc_part sc_u1 sc_u1q0 sc_u2 sc_u2q0; do
  [ $rep = 2 ] && [ $v = sc_part ] && continue
  lib=$REPO/build/ab/${v%q0}.so; q=1; [ $v != ${v%q0} ] && q=0
  echo "== $v (rep $rep)" | tee -a $S
  [ $q = 1 ] && VRT_LIB=$lib VRT_SCENE_QUADS=$q tools/profile_scene.sh b36_$v 512 2>&1 | grep -v "^wrote" | tee -a $S
  O=$REPO/gpurun_out/b36/lut_$v
  ( cd /tmp && export TMPDIR=/tmp && VRT_LIB=$lib VRT_SCENE_QUADS=$q timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $REPO/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $O.json 2> $O.err ) || echo "trace $v failed" | tee -a $S
  python3 tools/summarize_prof.py $O ${O}_summary.json > /dev/null; rm -rf $O
  python3 - <<PY | tee -a $S
import json
j=json.load(open("${O}_summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k: print("$v LUT 1024^3:", k[:70], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
done
done
