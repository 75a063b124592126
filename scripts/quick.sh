#!/bin/bash
# GPU quick check: parity tests (PYTEST_ARGS to narrow), 3-lane throughput, then one-lane per-operator
# tables for each config string given ("K=V K=V" ...; "" = defaults).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS} 2>&1 | tail -40
fi
for cfg in "$@"; do
  echo "== throughput [$cfg]"
  ( [ -n "$cfg" ] && export $cfg; python bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 | cut -c1-160 )
done
bash scripts/ab_prof.sh "$@"
