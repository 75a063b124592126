#!/bin/bash
# lanes x micro-batch sweep of the default workload (batch 48 so every lane count divides it evenly)
cd "$(dirname "$0")/.."
for l in 1 2 3 4; do for mb in 2 4 8; do
  v=$(PCLSEG_LANES=$l python bench.py --steps 30 --warmup 4 --cpu-seconds 0 --batch ${1:-48} --micro-batch $mb | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "lanes=$l mb=$mb  $v"
done; done
