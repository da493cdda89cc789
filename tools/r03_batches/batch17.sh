#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/b17
S=gpurun_out/b17/summary.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/b17/pytest.log 2>&1; echo "pytest rc $?" | tee -a $S
tail -3 gpurun_out/b17/pytest.log | tee -a $S
cat > /tmp/cases.txt <<'EOC'
prev c3 | VRT_LIB=build/ab/prev.so | --config 3 --warmup 5
new c3 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 3 --warmup 5
prev c5 | VRT_LIB=build/ab/prev.so | --config 5 --warmup 5
new c5 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 5 --warmup 5
prev c2 | VRT_LIB=build/ab/prev.so | --config 2 --warmup 5
new c2 | VRT_LIB=volumeraytracer_amd/lib/libvrt_hip.so | --config 2 --warmup 5
EOC
REPS=3 tools/ab_env.sh < /tmp/cases.txt 2>&1 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 120 --seed 41 2>&1 | tail -1 | tee -a $S
timeout -k 10 300 python3 tools/fuzz_gpu.py --seconds 120 --seed 42 --big 2>&1 | tail -1 | tee -a $S
for c in 3; do
  tools/profile.sh r03c$c --config $c > gpurun_out/b17/profile_c$c.log 2>&1
  echo "config $c profiled: $(tail -1 gpurun_out/b17/profile_c$c.log)" | tee -a $S
done
