#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats (headline + both Darknet
# workloads), one-lane per-operator time and traffic tables, PMC traffic passes (separate runs,
# kernel-trace only), SQ counters.  Everything lands in gpurun_out/<round>/; copy what is to be
# judged into profiles/ afterwards (scripts/collect_profiles.sh).   usage: profile_round.sh r02
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=${1:-r02}
O=gpurun_out/$R
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
B="python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
for wl in darknet53_64x2048 darknet21_32x1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary > $O/stats_$wl.log 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
# one lane: per-operator times, traffic and SQ counters
( export PCLSEG_LANES=1
  rocprofv3 --kernel-trace --output-format csv -d $O/kt1 -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1_fetch -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1_write -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1_sq -- $B > /dev/null 2>&1 )
python3 profiles/per_op_breakdown.py $(find $O/kt1 -name '*kernel_trace.csv' | head -1) > $O/per_op.txt
python3 profiles/per_op_counters.py $O/per_op.txt $O/pmc1_fetch $O/pmc1_write $O/pmc1_sq > $O/per_op_counters.txt
SCANS=$((32 * 7))
python3 profiles/make_traffic_json.py $O/pmc_fetch $O/pmc_write $SCANS > $O/traffic.json
for d in stats stats_darknet53_64x2048 stats_darknet21_32x1024; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
tail -c 1500 $O/bench.json; echo; tail -3 $O/per_op.txt; cat $O/traffic.json | head -12
