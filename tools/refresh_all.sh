#!/bin/bash
# Runs on the GPU box: counter passes, kernel stats and the bench line of every workload for one round tag; everything
# that is to be kept lands in gpurun_out/ (copy it into profiles/ afterwards).  Usage: bash tools/refresh_all.sh <tag>
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
TAG=${1:-r03}
( while true; do sleep 45; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
bash tools/refresh_profiles.sh $TAG c2 0 "random spheres 1920x1080 64spp depth50 (headline configuration)" profiles/${TAG}_isa_mix_c2.json || exit 1
bash tools/refresh_profiles.sh $TAG c1 0 "default scene 400x225 16spp depth50" profiles/${TAG}_isa_mix_c1.json || exit 1
bash tools/refresh_profiles.sh $TAG c3 128 "cornell box 1024x1024, one launch of 128 of the 1024 spp, depth50" profiles/${TAG}_isa_mix_c3.json || exit 1
bash tools/refresh_profiles.sh $TAG c4 8 "100k mixed objects 3840x2160, 8 of the 256 spp, depth50" profiles/${TAG}_isa_mix_c4.json || exit 1
bash tools/refresh_profiles.sh $TAG c5 8 "1M spheres 3840x2160, 8 of the 4096 spp, depth50" profiles/${TAG}_isa_mix_c5.json || exit 1
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
echo "all refreshed"
