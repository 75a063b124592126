#!/bin/bash
# Round profile: tests, bench, rocprofv3 kernel stats and PMC traffic passes (separate runs).
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; mkdir -p gpurun_out
R=${1:-r01}
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/${R}_pytest.log
python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
B="python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -- $B > gpurun_out/${R}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- $B > /dev/null 2>&1
cat gpurun_out/${R}_pytest.log; cat gpurun_out/${R}_bench.json | cut -c1-400
