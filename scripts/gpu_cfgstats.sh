#!/bin/bash
# per-kernel durations of selected BASELINE configurations (ONLY=C4,C5 ...), default build or NR_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-cfgstats}; mkdir -p $OUT
for c in ${ONLY//,/ }; do
  ONLY=$c timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o st_$c -- python scripts/bench_configs.py > $OUT/run_$c.log 2>&1
  python scripts/rocpd_stats.py $OUT/st_${c}_results.db $OUT/kernel_stats_$c.csv > /dev/null
  echo "== $c"; head -14 $OUT/kernel_stats_$c.csv | cut -c1-70,112-170
done
rm -f $OUT/*_results.db
