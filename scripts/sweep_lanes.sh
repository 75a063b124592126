#!/bin/bash
# lanes x micro-batch sweep of the default workload: sweep_lanes.sh "lanes..." "mbs..." [extra env]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for l in $1; do for mb in $2; do
  v=$(env PCLSEG_LANES=$l $3 python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-secondary --micro-batch $mb | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "lanes=$l mb=$mb $3  $v"
done; done
