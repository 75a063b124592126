#!/bin/bash
# One-lane per-operator profile of a workload: kernel-trace times + separate PMC passes (kernel-trace only):
# FETCH_SIZE, WRITE_SIZE, two SQ sets (wait / stall / active split, MFMA busy, LDS conflicts, instruction mix).
# usage: per_op_profile.sh <workload> <out tag> [steps]   ->  gpurun_out/<tag>_per_op.txt, <tag>_per_op_counters.txt
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
WL=${1:-ssv2_64x2048}; TAG=${2:-prof}; STEPS=${3:-4}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export PCLSEG_LANES=1
B="python3 bench.py --workload $WL --steps $STEPS --warmup 2 --cpu-seconds 0 --no-secondary"
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $B > $O/bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/sq1 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- $B > /dev/null 2>&1
python3 profiles/per_op_table.py $WL $O > gpurun_out/${TAG}_per_op_counters.txt
cat gpurun_out/${TAG}_per_op_counters.txt
