#!/bin/bash
# Runs on the GPU box: counter passes and kernel stats of one workload, summaries under gpurun_out/ (copy what is to be
# kept into profiles/ afterwards).  Usage: bash tools/refresh_profiles.sh <round tag> <workload> <spp or 0> "<description>" [isa mix json]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
TAG=$1; WL=$2; SPP=$3; DESC=$4; ISA=${5:-}
EXTRA=""; SUF=""
if [ "$SPP" != "0" ]; then EXTRA="--spp $SPP"; SUF="_${SPP}spp"; fi
rm -rf gpurun_out/pmc gpurun_out/stats
bash profiles/collect_pmc.sh "$WL" "$EXTRA" || exit 1
python3 profiles/summarize_pmc.py gpurun_out/pmc "$WL" > /dev/null || exit 1
python3 profiles/derive_pmc.py gpurun_out/pmc "profiles/${TAG}_pmc_${WL}${SUF}.json" "$DESC" "$WL" $ISA > /dev/null || exit 1
bash profiles/collect_stats.sh --workload "$WL" $EXTRA --steps 3 --warmup 1 || exit 1
find gpurun_out/stats -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} "profiles/${TAG}_kernel_stats_${WL}.csv"
cp profiles/${TAG}_pmc_${WL}${SUF}.json profiles/${TAG}_kernel_stats_${WL}.csv profiles/pmc_latest.json gpurun_out/ 2>/dev/null
echo "refreshed $WL"
