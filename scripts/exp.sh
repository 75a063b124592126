cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $1 $2"; env $1 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 $2 2>&1 | grep -E "metric|rror" | cut -c60-100; }
run "PCLSEG_CK=32 PCLSEG_LANES=3" "--micro-batch 4"
run "PCLSEG_CK=32 PCLSEG_LANES=4" "--micro-batch 4"
run "PCLSEG_CK=64 PCLSEG_LANES=3" "--micro-batch 4"
run "PCLSEG_CK=64 PCLSEG_LANES=4" "--micro-batch 4"
run "PCLSEG_CK=32 PCLSEG_LANES=4" "--micro-batch 8"
run "PCLSEG_CK=32 PCLSEG_LANES=4" "--micro-batch 2"
run "PCLSEG_CK=32 PCLSEG_LANES=8" "--micro-batch 2"
run "PCLSEG_CK=32 PCLSEG_LANES=4" "--micro-batch 4 --batch 64"
