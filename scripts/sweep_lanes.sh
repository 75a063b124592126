#!/bin/bash
# lanes x micro-batch sweep of the default workload: sweep_lanes.sh [batch] 
cd "$(dirname "$0")/.."
for l in 2 3 4; do for mb in 2 3 4 6 8; do
  v=$(PCLSEG_LANES=$l python bench.py --steps 40 --warmup 4 --cpu-seconds 0 --batch ${1:-32} --micro-batch $mb | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "lanes=$l mb=$mb  $v"
done; done
