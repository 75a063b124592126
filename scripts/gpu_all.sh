#!/bin/bash
# ONE gpurun call that collects, in priority order, everything a round needs from an MI355X — written for a pool that may
# open for a single call.  Every pass has its own timeout and writes under gpurun_out/<round>/ as it goes, so a call that
# is cut short still returns what finished.      usage: gpu_all.sh r06 [quick]
#   1  tests_shipped.log        python -m pytest tests -m gpu  (no -x: every failure is wanted) on the shipped library
#   2  bench.json               the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5
#   3  stats_*_kernel_stats.csv rocprofv3 --kernel-trace --stats per workload (no counters in this pass)
#   4  ab_bitwise.log, ab.log   candidates (build/libpclseg_cand.so) vs shipped: bit identity, then throughput alternating
#   5  tests_cand.log           operator + network tests on the candidate library
#   6  traffic.json             separate --pmc FETCH_SIZE / WRITE_SIZE passes -> bytes per scan; bench line against it
#   7  *_per_op_counters.txt    one-lane per-operator counters
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=${1:-r06}; QUICK=$2
O=gpurun_out/$R
mkdir -p $O
make -s all candidates > $O/make.log 2>&1 || tail -5 $O/make.log     # (no-op when the libraries that travelled are current)
CAND=$PWD/build/libpclseg_cand.so
WLS="ssv2_64x2048 darknet53_64x2048 darknet21_32x1024"
rocminfo 2>/dev/null | grep -m1 gfx9 > $O/device.txt; rocm-smi --showclocks 2>/dev/null | head -20 >> $O/device.txt

echo "== 1 tests (shipped)"; timeout 2400 python -m pytest tests -m gpu -q -rxXs -p no:cacheprovider > $O/tests_shipped.log 2>&1; tail -4 $O/tests_shipped.log
echo "== 2 bench"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
[ "$QUICK" = quick ] && exit 0

echo "== 3 kernel stats"
for wl in $WLS; do
  st=5; [ $wl = darknet53_64x2048 ] && st=3
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps $st --warmup 2 --cpu-seconds 0 --no-secondary > $O/stats_$wl.log 2>&1
  f=$(find $O/stats_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/stats_${wl}_kernel_stats.csv && rm -rf $O/stats_$wl
done
ls $O/*kernel_stats.csv

if [ -f $CAND ]; then
  echo "== 4 A/B candidates vs shipped"
  timeout 900 python scripts/ab_bitwise.py $PWD/pclsegmentation_amd/libpclseg.so $CAND > $O/ab_bitwise.log 2>&1; tail -8 $O/ab_bitwise.log
  for wl in $WLS; do
    for lib in "" "$CAND" "" "$CAND"; do
      ( [ -n "$lib" ] && export PCLSEG_DEBUG=1 PCLSEG_LIB=$lib; echo "== $wl [${lib:-shipped}]"
        timeout 240 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 |
          python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('spread'), d['ms_per_step'])" )
    done
  done > $O/ab.log 2>&1
  cat $O/ab.log
  # one lane, per OPERATOR (launch position in the micro-batch), SqueezeSegV2: shipped | candidates | shipped | candidates
  timeout 900 bash scripts/ab_prof.sh "" "PCLSEG_DEBUG=1 PCLSEG_LIB=$CAND" "" "PCLSEG_DEBUG=1 PCLSEG_LIB=$CAND" > $O/ab_per_op_ssv2.txt 2>&1; tail -26 $O/ab_per_op_ssv2.txt
  # one lane, per KERNEL: which candidate pays where (both libraries, same box)
  for lib in "" "$CAND"; do
    tag=shipped; [ -n "$lib" ] && tag=cand
    ( [ -n "$lib" ] && export PCLSEG_DEBUG=1 PCLSEG_LIB=$lib
      for wl in ssv2_64x2048 darknet21_32x1024; do
        PCLSEG_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lane1_${tag}_$wl -- python3 bench.py --workload $wl --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary > /dev/null 2>&1
        f=$(find $O/lane1_${tag}_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/lane1_${tag}_${wl}_kernel_stats.csv && rm -rf $O/lane1_${tag}_$wl
      done )
  done
  echo "== 5 tests (candidates)"
  PCLSEG_DEBUG=1 PCLSEG_LIB=$CAND timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider > $O/tests_cand.log 2>&1; tail -4 $O/tests_cand.log
fi

echo "== 6 traffic"
for wl in $WLS; do
  st=5; [ $wl = darknet53_64x2048 ] && st=3
  B="python3 bench.py --workload $wl --steps $st --warmup 2 --cpu-seconds 0 --no-secondary"
  timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$wl -- $B > /dev/null 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$wl -- $B > /dev/null 2>&1
done
python3 profiles/make_traffic_json.py $O > $O/traffic.json && PCLSEG_TRAFFIC_JSON=$O/traffic.json timeout 900 python bench.py > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err
head -30 $O/traffic.json; rm -rf $O/pmc_fetch_* $O/pmc_write_*

echo "== 7 per-operator counters"
for wl in $WLS; do
  st=4; [ $wl = darknet53_64x2048 ] && st=2
  timeout 900 bash scripts/per_op_profile.sh $wl $R/$wl $st > /dev/null 2>&1
done
tail -2 $O/*_per_op_counters.txt
