# A/B of the shipped kernels against the round-3/4 kernel set (build/libpclseg_r4.so, scripts/build_baseline.sh 9098503 r4):
# bit-identity of logits and predictions on all three networks, then throughput of both on the same box, alternating.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BASE=$GRAFT_REPO_ROOT/build/libpclseg_r4.so
timeout 900 python scripts/ab_bitwise.py $GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg.so $BASE > gpurun_out/r05_s2_bitwise.log 2>&1; tail -12 gpurun_out/r05_s2_bitwise.log
for wl in ssv2_64x2048 darknet53_64x2048 darknet21_32x1024; do
  for lib in "" "$BASE" "" "$BASE"; do
    ( [ -n "$lib" ] && export PCLSEG_DEBUG=1 PCLSEG_LIB=$lib; echo "== $wl [${lib:-shipped}]"; timeout 240 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('spread'), d['ms_per_step'])" )
  done
done > gpurun_out/r05_s2_ab.log 2>&1
cat gpurun_out/r05_s2_ab.log
