#!/bin/bash
# matrix-core / LDS counters of the Darknet-53 workload, one lane; separate passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PCLSEG_LANES=1
for set in ${PMC_SETS:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" "LDSBankConflict"}; do
  d=gpurun_out/pmc_dn_$(echo $set | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 bench.py --workload ${1:-darknet53_64x2048} --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
  print(sys.argv[1], "no output"); sys.exit()
agg = collections.defaultdict(float)
big = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
  if "pclseg" not in r["Kernel_Name"]: continue
  agg[r["Counter_Name"]] += float(r["Counter_Value"])
  if int(r["Grid_Size"]) >= 4096 * 256 and "conv_kernel<4, 2, 2" in r["Kernel_Name"]:
    big[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: "%.4g" % v for k, v in agg.items()}, "| deep 3x3 only:", {k: "%.4g" % v for k, v in big.items()})
PY
done
