# GPU check of the round's pending work: (1) the shipped library after the ABI / distributed changes, (2) the
# -DPCLSEG_R4X kernel variants: parity tests, one-lane per-operator A/B and 3-lane throughput against the shipped
# kernels on the same box, (3) phase stamps of the tail kernel and of Darknet's 1x1 layers.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R4X=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_r4x.so
timeout 1200 python -m pytest tests/test_preproc_golden.py tests/test_gpu_ops.py tests/test_c_abi.py tests/test_gpu_models.py -m gpu -x -q -k "preproc or golden or fused or full_size_kitti or c1_reference or normalize or head or abi or range or host_boundary" > gpurun_out/r04_run2_tests.log 2>&1; tail -4 gpurun_out/r04_run2_tests.log
PCLSEG_LIB=$R4X timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "golden or fused or full_size_kitti or c1_reference or intermediate or micro_batch" > gpurun_out/r04_run2_tests_r4x.log 2>&1; tail -4 gpurun_out/r04_run2_tests_r4x.log
SKIP_TESTS=1 timeout 900 bash scripts/quick.sh "" "PCLSEG_LIB=$R4X" "" "PCLSEG_LIB=$R4X" > gpurun_out/r04_run2_ab.log 2>&1; cat gpurun_out/r04_run2_ab.log | tail -40
timeout 300 bash scripts/stamps.sh ssv2_64x2048 fire13 > gpurun_out/r04_run2_stamps.log 2>&1
timeout 600 bash scripts/stamps.sh darknet53_64x2048 enc5/residual_0/conv1 enc4/residual_0/conv1 dec5/block/conv1 enc5/residual_0/conv2 >> gpurun_out/r04_run2_stamps.log 2>&1
cat gpurun_out/r04_run2_stamps.log
