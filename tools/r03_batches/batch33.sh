#!/bin/bash
# batch 33: padded brick row stride (VRT_SLAB_PAD_Z=1 opportunistic, =2 planned) vs HEAD: kernel ms + LDS conflict counters
set -o pipefail
mkdir -p gpurun_out/b33
S=gpurun_out/b33/summary.txt
MAIN=volumeraytracer_amd/lib/libvrt_hip.so
CONFIGS="3 4 5" REPS=2 tools/ab.sh $MAIN build/ab/padz1.so build/ab/padz2.so 2>&1 | tee -a $S
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in main padz1 padz2; do
  lib=$REPO/$MAIN; [ $v != main ] && lib=$REPO/build/ab/$v.so
  for c in 3 5; do
    O=$REPO/gpurun_out/b33/pmc_${v}_$c
    VRT_LIB=$lib rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O -o pmc -- python3 $REPO/bench.py --config $c --cpu-seconds 0 --steps 2 --warmup 5 > $O.json 2> $O.err || echo "pmc $v $c failed" | tee -a $REPO/$S
    python3 $REPO/tools/summarize_prof.py $O ${O}_summary.json > /dev/null; rm -rf $O
    python3 - <<PY | tee -a $REPO/$S
import json
j=json.load(open("${O}_summary.json"))
for k,c in j["counters"].items():
    if "march_slab" in k:
        d={a:c[a]["per_dispatch"] for a in c}
        print("$v cfg $c", k[:48], d, "conflict/active %.3f" % (d["SQ_LDS_BANK_CONFLICT"]/max(1,d["SQ_LDS_IDX_ACTIVE"])))
PY
  done
done
