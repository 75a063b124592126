#!/bin/bash
# Copy the judged summaries of a profile round from gpurun_out/<round>/ (scratch) into profiles/ (tracked).
# usage: collect_profiles.sh r05
cd "$(dirname "$0")/.."
R=${1:-r05}
O=gpurun_out/$R
[ -d "$O" ] || { echo "no $O (run scripts/profile_round.sh $R on the GPU box first)"; exit 1; }
cp $O/bench.json profiles/${R}_bench.json
cp $O/traffic.json profiles/${R}_traffic.json
for wl in ssv2_64x2048 darknet53_64x2048 darknet21_32x1024; do
  cp $O/stats_${wl}_kernel_stats.csv profiles/${R}_${wl}_kernel_stats.csv
  cp $O/${wl}_per_op_counters.txt profiles/${R}_${wl}_per_op_counters.txt
done
ls -la profiles/${R}_*
