#!/bin/bash
# A/B of the wide-layer kernel variants (tuning build): usage dn_variants.sh "ENV=V ENV=V" ...
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export PCLSEG_DEBUG=1 PCLSEG_LIB=$PWD/build/libpclseg_tuning.so
for cfg in "$@"; do
  for wl in darknet53_64x2048 darknet21_32x1024; do
    ( [ -n "$cfg" ] && export $cfg; python3 bench.py --workload $wl --steps 8 --warmup 3 --cpu-seconds 0 --no-secondary 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-40s %-20s %8.1f scans/s  %.3f ms/step' % ('$cfg', '$wl', d['value'], d['ms_per_step']))" )
  done
done
