#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b16
S=gpurun_out/b16/summary.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/b16/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -3 gpurun_out/b16/pytest.log | tee -a $S
cat > /tmp/cases.txt <<'EOC'
prev c3 | VRT_LIB=build/ab/prev.so | --config 3 --warmup 5
new c3 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 3 --warmup 5
prev c2 | VRT_LIB=build/ab/prev.so | --config 2 --warmup 5
new c2 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 2 --warmup 5
prev c5 | VRT_LIB=build/ab/prev.so | --config 5 --warmup 5
new c5 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 5 --warmup 5
prev c4 | VRT_LIB=build/ab/prev.so | --config 4 --warmup 5
new c4 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 4 --warmup 5
EOC
REPS=3 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
mkdir -p gpurun_out/prof_r03order
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --cpu-seconds 0 --steps 4 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03order/err.txt )
python3 tools/summarize_prof.py gpurun_out/prof_r03order/t gpurun_out/prof_r03order/trace_summary.json
python3 - <<'PY' | tee -a $S
import csv,glob,collections
rows=collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_r03order/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(rows.items(), key=lambda kv:-sum(kv[1])):
    if "at::native" in k or "elementwise" in k: continue
    print("%-72s %5d avg_us %9.1f" % (k, len(v), sum(v)/len(v)/1e3))
PY
rm -rf gpurun_out/prof_r03order/t
