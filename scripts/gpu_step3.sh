# Third GPU call of a round: re-measure (kernel stats, PMC traffic, per-operator counters of the three workloads) on the
# library that ships, then `scripts/collect_profiles.sh r05` here copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 bash scripts/profile_round.sh r05 > gpurun_out/r05_s3_profile.log 2>&1; tail -40 gpurun_out/r05_s3_profile.log
