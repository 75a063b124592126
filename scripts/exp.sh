cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $1 $2"; env $1 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 $2 2>&1 | grep -E "metric|rror" | cut -c60-100; }
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
run "PCLSEG_DIRECT1X1=0" ""
run "PCLSEG_DIRECT1X1=1" ""
run "PCLSEG_DIRECT1X1=1 PCLSEG_LANES=1" ""
for w in darknet21_32x1024 darknet53_64x2048; do python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | grep -E "metric|rror" | cut -c30-110; done
PCLSEG_LANES=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t18 -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/t18.log 2>&1
