#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b10
S=gpurun_out/b10/summary.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/b10/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -4 gpurun_out/b10/pytest.log | tee -a $S
cat > /tmp/cases.txt <<'EOC'
radix c3 | VRT_RAY_ORDER=radix | --config 3 --warmup 5
tiles c3 | VRT_RAY_ORDER=tiles | --config 3 --warmup 5
radix c5 | VRT_RAY_ORDER=radix | --config 5 --warmup 5
tiles c5 | VRT_RAY_ORDER=tiles | --config 5 --warmup 5
radix c4 | VRT_RAY_ORDER=radix | --config 4 --warmup 5
tiles c4 | VRT_RAY_ORDER=tiles | --config 4 --warmup 5
radix c2 | VRT_RAY_ORDER=radix | --config 2 --warmup 5
tiles c2 | VRT_RAY_ORDER=tiles | --config 2 --warmup 5
EOC
REPS=2 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
mkdir -p gpurun_out/prof_r3order
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3order/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --cpu-seconds 0 --steps 4 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_r3order/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r3order/err.txt )
python3 - <<'PY' | tee -a $S
import csv,glob,collections
rows=collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_r3order/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(rows.items(), key=lambda kv:-sum(kv[1])):
    print("%-72s %5d avg_us %9.1f" % (k, len(v), sum(v)/len(v)/1e3))
PY
rm -rf gpurun_out/prof_r3order/t
