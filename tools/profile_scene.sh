#!/bin/bash
# tools/profile_scene.sh -- rocprofv3 kernel trace of the scene build (k_scene_build) on the GPU box.
#   usage: tools/profile_scene.sh <tag> [N]   -> gpurun_out/prof_<tag>/scene_summary.json (+ achieved GB/s on stdout)
set -u
TAG=$1; N=${2:-512}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/scene" -o scene -- python3 $REPO/tools/scene_build_time.py $N > "$OUT/scene.log" 2> "$OUT/scene.err" || echo "scene trace failed"
python3 $REPO/tools/summarize_prof.py "$OUT/scene" "$OUT/scene_summary.json"; rm -rf "$OUT/scene"
python3 - "$OUT/scene_summary.json" $N <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); N = int(sys.argv[2])
for k, v in j["kernels"].items():
    if "k_scene_build" not in k:
        continue
    fam = "float" if "float" in k.split("k_scene_build")[1][:40] else "fixed-point"
    vox_out = 16 if fam == "float" else 8
    # compulsory traffic: every log / translucency value read once (4 + 4 B per volume voxel), every field voxel written once
    byts = 8.0 * N ** 3 + vox_out * (N - 2) ** 3
    print("%s: %d launches, avg %.3f ms -> %.2f GB compulsory -> %.0f GB/s (%.0f %% of 8 TB/s)" %
          (k[:70], v["launches"], v["avg_ns"] / 1e6, byts / 1e9, byts / v["avg_ns"], 100 * byts / v["avg_ns"] / 8000))
PY
