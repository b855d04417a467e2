#!/bin/bash
# per-kernel durations of the stage calls (rocprofv3 kernel trace over scripts/stage_times.py)
TAG=${1:-kstats}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st -- python scripts/stage_times.py > $OUT/run.log 2>&1
python scripts/rocpd_stats.py $OUT/st_results.db $OUT/kernel_stats.csv
rm -f $OUT/*_results.db
head -24 $OUT/kernel_stats.csv | cut -c1-90,112-170
