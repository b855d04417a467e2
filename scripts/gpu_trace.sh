#!/bin/bash
# kernel trace (per-kernel average durations) of scripts/stage_times.py for a list of NR_STAGE_FLAGS values
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-trace}; mkdir -p $OUT
for fl in ${FLAGS:-0}; do
  NR_STAGE_FLAGS=$fl ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st_$fl -- python scripts/stage_times.py > $OUT/st_$fl.log 2>&1
  python scripts/rocpd_stats.py $OUT/st_${fl}_results.db $OUT/kernel_stats_$fl.csv > /dev/null 2>&1
  echo "== flags $fl"; head -${HEAD:-14} $OUT/kernel_stats_$fl.csv | cut -c1-75,100-175; tail -1 $OUT/st_$fl.log | cut -c1-400
  rm -f $OUT/*_results.db
done
