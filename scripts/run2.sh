cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preproc_golden.py tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -x -q -k "preproc or golden or fused or full_size_kitti or c1_reference or intermediate or normalize or head" > gpurun_out/r04_run2_tests.log 2>&1; tail -4 gpurun_out/r04_run2_tests.log
SKIP_TESTS=1 timeout 900 bash scripts/quick.sh "PCLSEG_LIB=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_base.so" "" "PCLSEG_LIB=$GRAFT_REPO_ROOT/pclsegmentation_amd/libpclseg_base.so" "" > gpurun_out/r04_run2_ab.log 2>&1; cat gpurun_out/r04_run2_ab.log | tail -40
timeout 300 bash scripts/stamps.sh ssv2_64x2048 fire13 > gpurun_out/r04_run2_stamps.log 2>&1
timeout 600 bash scripts/stamps.sh darknet53_64x2048 enc5/residual_0/conv1 enc4/residual_0/conv1 dec5/block/conv1 enc5/residual_0/conv2 >> gpurun_out/r04_run2_stamps.log 2>&1
cat gpurun_out/r04_run2_stamps.log
