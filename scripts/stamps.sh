#!/bin/bash
# Phase split (s_memtime stamps, `make stamps` build) of the launches whose name contains one of the given
# strings, one lane.   usage: stamps.sh <workload> <layer substring> [<layer substring> ...]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
WL=$1; shift
for name in "$@"; do
  PCLSEG_DEBUG=1 PCLSEG_LIB=$PWD/build/libpclseg_stamps.so PCLSEG_STAMP="$name" PCLSEG_LANES=1 \
    python bench.py --workload $WL --steps 8 --warmup 2 --cpu-seconds 0 --no-secondary 2>&1 | grep -A8 "^STAMPS"
done
