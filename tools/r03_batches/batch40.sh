#!/bin/bash
# batch 40: scene build with the runs of every row shifted to a multiple of four voxels of the field's linear index (sc_zs) (VRT_SCENE_ZSHIFT=1, sc_zs) vs unshifted (=0, main)
set -o pipefail
mkdir -p gpurun_out/b40
REPO=$GRAFT_REPO_ROOT
S=$REPO/gpurun_out/b40/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3 | tee -a $S
for rep in 1 2; do
for v in main sc_zs; do
  lib=$REPO/volumeraytracer_amd/lib/libvrt_hip.so; export VRT_SCENE_ZSHIFT=1; [ $v = main ] && export VRT_SCENE_ZSHIFT=0
  echo "== $v (rep $rep)" | tee -a $S
  VRT_LIB=$lib tools/profile_scene.sh b40_$v 512 2>&1 | grep -v "^wrote" | tee -a $S
  O=$REPO/gpurun_out/b40/lut_$v
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
