#!/bin/bash
# Round profile on the GPU box (every profiler pass under its own timeout: a hung rocprofv3 must not eat the call).  Everything lands in gpurun_out/<round>/; scripts/collect_profiles.sh copies what is
# to be judged into profiles/ (tracked).     usage: profile_round.sh r05
#   bench.json                       the default `python bench.py` line (driver contract + secondary rows + parity_check)
#   stats_<workload>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command (3 lanes), per workload
#   traffic.json                     HBM-side bytes per scan of every workload: separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   <workload>_per_op_counters.txt   one lane: per-operator time, MFMA rate, traffic, SQ wait/stall/active, LDS conflicts
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=${1:-r05}
O=gpurun_out/$R
rm -rf $O; mkdir -p $O
for wl in ssv2_64x2048 darknet53_64x2048 darknet21_32x1024; do
  st=5; [ $wl = darknet53_64x2048 ] && st=3
  B="python3 bench.py --workload $wl --steps $st --warmup 2 --cpu-seconds 0 --no-secondary"
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- $B > $O/stats_$wl.log 2>&1
  f=$(find $O/stats_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/stats_${wl}_kernel_stats.csv
  timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$wl -- $B > /dev/null 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$wl -- $B > /dev/null 2>&1
done
python3 profiles/make_traffic_json.py $O > $O/traffic.json
# the bench line LAST, against the traffic figure just measured on this very binary (bench.py withholds a
# figure whose csrc sha differs from pclseg_build_sha() of the library it loaded)
PCLSEG_TRAFFIC_JSON=$O/traffic.json timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
for wl in ssv2_64x2048 darknet53_64x2048 darknet21_32x1024; do
  st=4; [ $wl = darknet53_64x2048 ] && st=2
  timeout 900 bash scripts/per_op_profile.sh $wl $R/$wl $st > /dev/null 2>&1
  cp gpurun_out/$R/${wl}_per_op_counters.txt $O/ 2>/dev/null
done
tail -c 1200 $O/bench.json; echo; cat $O/traffic.json | head -30; tail -2 $O/*_per_op_counters.txt
