#!/bin/bash
# batch 41: scene build tile shapes (rows x z): 8x32 (HEAD), 8x64, 8x128, 4x128, 4x64 -- wider runs share fewer lines across workgroups
set -o pipefail
mkdir -p gpurun_out/b41
REPO=$GRAFT_REPO_ROOT
S=$REPO/gpurun_out/b41/summary.txt
for v in t8x128 t4x64; do
VRT_LIB=$REPO/build/ab/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_scene.py -m gpu -x -q 2>&1 | tail -2 | tee -a $S
done
for rep in 1 2; do
for v in main t8x64 t8x128 t4x128 t4x64; do
  lib=$REPO/volumeraytracer_amd/lib/libvrt_hip.so; [ $v != main ] && lib=$REPO/build/ab/$v.so
  echo "== $v (rep $rep)" | tee -a $S
  VRT_LIB=$lib tools/profile_scene.sh b41_$v 512 2>&1 | grep -v "^wrote" | tee -a $S
  O=$REPO/gpurun_out/b41/lut_$v
  ( cd /tmp && export TMPDIR=/tmp && VRT_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $REPO/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $O.json 2> $O.err ) || echo "trace $v failed" | tee -a $S
  python3 tools/summarize_prof.py $O ${O}_summary.json > /dev/null; rm -rf $O
  python3 - <<PY | tee -a $S
import json
j=json.load(open("${O}_summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k: print("$v LUT 1024^3:", k[:70], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
done
done
