# GPU check of the round's pending work: (1) the shipped library after the ABI / distributed changes, (2) the
# -DPCLSEG_R4X kernel variants: parity tests, one-lane per-operator A/B and 3-lane throughput against the shipped
# kernels on the same box, (3) phase stamps of the tail kernel and of Darknet's 1x1 layers.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R4X=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_r4x.so
timeout 1200 python -m pytest tests/test_preproc_golden.py tests/test_gpu_ops.py tests/test_c_abi.py tests/test_gpu_models.py -m gpu -x -q -k "preproc or golden or fused or full_size_kitti or c1_reference or normalize or head or abi or range or host_boundary" > gpurun_out/r04_run2_tests.log 2>&1; tail -4 gpurun_out/r04_run2_tests.log
PCLSEG_LIB=$R4X timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "conv2d or golden or fused or full_size or c1_reference or intermediate or micro_batch or strides" > gpurun_out/r04_run2_tests_r4x.log 2>&1; tail -4 gpurun_out/r04_run2_tests_r4x.log
timeout 900 python scripts/ab_bitwise.py $GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg.so $R4X > gpurun_out/r04_run2_bitwise.log 2>&1; tail -12 gpurun_out/r04_run2_bitwise.log
NOSLAB=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_r4x_noslab.so
SKIP_TESTS=1 timeout 1200 bash scripts/quick.sh "" "PCLSEG_LIB=$R4X" "PCLSEG_LIB=$NOSLAB" "" "PCLSEG_LIB=$R4X" "PCLSEG_LIB=$NOSLAB" > gpurun_out/r04_run2_ab.log 2>&1; cat gpurun_out/r04_run2_ab.log | tail -44
timeout 300 bash scripts/stamps.sh ssv2_64x2048 fire13 > gpurun_out/r04_run2_stamps.log 2>&1
timeout 600 bash scripts/stamps.sh darknet53_64x2048 enc5/residual_0/conv1 enc4/residual_0/conv1 dec5/block/conv1 enc5/residual_0/conv2 >> gpurun_out/r04_run2_stamps.log 2>&1
cat gpurun_out/r04_run2_stamps.log
# Darknet A/B (3 lanes): shipped kernels vs the r4x wide 1x1 kernel, both workloads
for lib in "" "PCLSEG_LIB=$R4X" "" "PCLSEG_LIB=$R4X"; do
  for wl in darknet53_64x2048 darknet21_32x1024; do
    ( [ -n "$lib" ] && export $lib; echo "== $wl [$lib]"; timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 | cut -c1-140 )
  done
done > gpurun_out/r04_run2_dn.log 2>&1
cat gpurun_out/r04_run2_dn.log
# the driver's own command on the shipped library: the whole bench line (five regions, spread, build, c1_gpu, parity_check)
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run2_bench.json 2> gpurun_out/r04_run2_bench.err; tail -c 3000 gpurun_out/r04_run2_bench.json; tail -3 gpurun_out/r04_run2_bench.err
