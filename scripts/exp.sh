cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in darknet21_32x1024 darknet53_64x2048 ssv2_32x240; do timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | grep -E "metric|rror" | cut -c1-330; done
PCLSEG_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dn53 -- python3 bench.py --workload darknet53_64x2048 --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/dn53.log 2>&1
