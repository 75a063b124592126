cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $1 $2"; env $1 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 $2 2>&1 | grep -E "metric|rror" | cut -c60-100; }
python -m pytest tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
run "A=1" ""
run "A=2" ""
run "PCLSEG_LANES=1" ""
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cam16 -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
