# The SHORT GPU check (about 10 minutes): is the r4x build bit-identical to the shipped one, and is it faster?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R4X=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_r4x.so
NOSLAB=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_r4x_noslab.so
timeout 600 python scripts/ab_bitwise.py $GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg.so $R4X > gpurun_out/r04_fast_bitwise.log 2>&1; tail -12 gpurun_out/r04_fast_bitwise.log
PCLSEG_LIB=$R4X timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wide_1x1" 2>&1 | tail -3
for wl in ssv2_64x2048 darknet53_64x2048 darknet21_32x1024; do
  for lib in "" "PCLSEG_LIB=$R4X" "PCLSEG_LIB=$NOSLAB" "" "PCLSEG_LIB=$R4X"; do
    ( [ -n "$lib" ] && export $lib; echo "== $wl [$lib]"; timeout 240 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['spread'], d['ms_per_step'])" )
  done
done > gpurun_out/r04_fast_ab.log 2>&1
cat gpurun_out/r04_fast_ab.log
