# First GPU minutes of a round: verify what ships — the whole GPU suite, then the driver's bench command.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rxXs > gpurun_out/r06_s1_tests.log 2>&1; tail -15 gpurun_out/r06_s1_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_s1_bench.json 2> gpurun_out/r06_s1_bench.err; tail -c 6000 gpurun_out/r06_s1_bench.json; tail -5 gpurun_out/r06_s1_bench.err
