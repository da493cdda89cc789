#!/bin/bash
# tools/profile.sh -- rocprofv3 passes over bench.py on the GPU box (run through gpurun).
#   usage: tools/profile.sh <tag> <bench args...>
# Writes gpurun_out/prof_<tag>/{trace,pmc_*}/...; copy the summaries you want judged to profiles/.
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip traces).
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# warmup 5: the engine settles its schedule choice for the workload during the first four calls (bench.py --warmup)
BENCH="python3 $REPO/bench.py --cpu-seconds 0 --steps 4 --warmup 5 $*"
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.json" 2> "$OUT/trace.err" || echo "trace pass failed"
python3 $REPO/tools/summarize_prof.py "$OUT/trace" "$OUT/trace_summary.json"; rm -rf "$OUT/trace"
pass() { # name counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc $name failed"
  python3 $REPO/tools/summarize_prof.py "$OUT/pmc_$name" "$OUT/pmc_${name}_summary.json"; rm -rf "$OUT/pmc_$name"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE
# (the tcp pass is optional and slow on some boxes: run it by hand when needed)
[ "${VRT_PROFILE_TCP:-0}" = 1 ] && pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE
echo done
