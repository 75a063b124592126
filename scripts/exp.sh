cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $1 $2"; env $1 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 $2 2>&1 | grep metric | cut -c60-100; }
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
run "A=1" ""
run "PCLSEG_LANES=2" "--micro-batch 8"
run "PCLSEG_LANES=3" "--micro-batch 4"
run "PCLSEG_LANES=2" "--batch 64"
