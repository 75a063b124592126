#!/bin/bash
# The THREE-lane view of the headline workload: what saturates when 2.7 kernels are in flight.
#   kt1 / kt3      kernel traces at 1 and 3 lanes: per-kernel inflation, co-residency matrix, kernels in flight
#   pmc3_*         one rocprofv3 --pmc pass per counter group at the default 3 lanes (kernel-trace only beside it)
# usage: lanes3_counters.sh <out tag>     -> gpurun_out/<tag>/...   and   gpurun_out/<tag>_3lane_counters.txt
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
TAG=${1:-r04_3lane}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-secondary"
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
PCLSEG_LANES=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt1 -- $B > $O/kt1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt3 -- $B > $O/kt3.log 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  # (every pass under its own timeout: a group that needs more TCC slots than the hardware has made rocprofv3 abort
  # and then hang until the box's limit in round 4)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc3_$i -- $B > $O/pmc3_$i.log 2>&1
  PCLSEG_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc1_$i -- $B > $O/pmc1_$i.log 2>&1
done <<'GROUPS'
SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
TCC_HIT_sum TCC_MISS_sum
SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_SGPR_SIMD_FULL_CSN
SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
FETCH_SIZE
WRITE_SIZE
GROUPS
python3 profiles/lanes3_table.py $O > gpurun_out/${TAG}_3lane_counters.txt 2> $O/table.err
tail -60 gpurun_out/${TAG}_3lane_counters.txt
