#!/bin/bash
# Runs on the GPU box: counter passes and kernel stats of the two large scenes, summaries under gpurun_out/.
# Usage: bash tools/refresh_profiles.sh
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
one() { # workload, spp, description
  rm -rf gpurun_out/pmc
  bash profiles/collect_pmc.sh "$1" "--spp $2" || return 1
  python3 profiles/summarize_pmc.py gpurun_out/pmc "$1" > /dev/null || return 1
  python3 profiles/derive_pmc.py gpurun_out/pmc "gpurun_out/r01_pmc_$1_$2spp.json" "$3" || return 1
  bash profiles/collect_stats.sh --workload "$1" --spp "$2" --steps 3 --warmup 1 || return 1
  find gpurun_out/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} "gpurun_out/r01_kernel_stats_$1.csv"
}
one c4 4 "c4 scene (100k mixed objects) 3840x2160 4 spp depth 50, one k_paths_stream launch" && \
one c5 8 "c5 scene (1M spheres) 3840x2160 8 spp depth 50, one k_paths_stream launch"
