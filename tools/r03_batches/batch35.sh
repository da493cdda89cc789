#!/bin/bash
# batch 35: what binds the 1024^3 LUT scene build?  counters of k_scene_build_tiled inside bench.py --config 5 (one step);
# and tiles dealt to the XCDs in quads (sc_quad) vs plain (sc_part)
set -o pipefail
mkdir -p gpurun_out/b35
REPO=$GRAFT_REPO_ROOT
S=$REPO/gpurun_out/b35/summary.txt
VRT_LIB=$REPO/build/ab/sc_quad.so timeout -k 10 600 python -m pytest tests/test_gpu_scene.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee -a $S
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in sc_part sc_quad; do
  O=$REPO/gpurun_out/b35/lut_$v
  VRT_LIB=$REPO/build/ab/$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $REPO/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $O.json 2> $O.err || echo "trace $v failed" | tee -a $S
  python3 $REPO/tools/summarize_prof.py $O ${O}_summary.json > /dev/null; rm -rf $O
  python3 - <<PY | tee -a $S
import json
j=json.load(open("${O}_summary.json"))
for k,v in j["kernels"].items():
    if "scene_build" in k: print("$v LUT 1024^3 (rep $rep):", k[:70], "avg_ms %.4f" % (v["avg_ns"]/1e6))
PY
done
done
pass() { v=$1; name=$2; shift; shift
  O=$REPO/gpurun_out/b35/pmc_${v}_$name
  echo "== $v $name" | tee -a $S
  VRT_LIB=$REPO/build/ab/$v.so timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o pmc -- python3 $REPO/bench.py --config 5 --cpu-seconds 0 --steps 1 --warmup 0 > $O.json 2> $O.err || { echo "pmc $v $name failed" | tee -a $S; return; }
  python3 $REPO/tools/summarize_prof.py $O ${O}_summary.json > /dev/null; rm -rf $O
  python3 - <<PY | tee -a $S
import json
j=json.load(open("${O}_summary.json"))
for k,c in j["counters"].items():
    if "scene_build" in k:
        print("$v $name", k[:40], {a:c[a]["per_dispatch"] for a in c})
PY
}
pass sc_part fetch FETCH_SIZE
pass sc_quad fetch FETCH_SIZE
pass sc_quad write WRITE_SIZE
pass sc_quad sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass sc_quad sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS
pass sc_quad tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
