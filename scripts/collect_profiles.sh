#!/bin/bash
# Copy the judged summaries of a profile round from gpurun_out/<round>/ (scratch) into profiles/ (tracked).
# usage: collect_profiles.sh r02
cd "$(dirname "$0")/.."
R=${1:-r02}
O=gpurun_out/$R
[ -d "$O" ] || { echo "no $O (run scripts/profile_round.sh $R on the GPU box first)"; exit 1; }
cp $O/bench.json profiles/${R}_bench.json
cp $O/stats_kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $O/stats_darknet53_64x2048_kernel_stats.csv profiles/${R}_dn53_kernel_stats.csv
cp $O/stats_darknet21_32x1024_kernel_stats.csv profiles/${R}_dn21_kernel_stats.csv
cp $O/per_op.txt profiles/${R}_per_op.txt
cp $O/per_op_counters.txt profiles/${R}_per_op_counters.txt
cp $O/traffic.json profiles/${R}_traffic.json
ls -la profiles/${R}_*
