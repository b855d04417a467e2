#!/bin/bash
# per-kernel durations of BASELINE configs 4 and 5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 4 5; do
  OUT=gpurun_out/cfg${c}_k; mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o st -- python scripts/profile_config$c.py > $OUT/run.log 2>&1
  python scripts/rocpd_stats.py $OUT/st_results.db $OUT/kernel_stats.csv; rm -f $OUT/*_results.db
  echo "== config $c: $(tail -1 $OUT/run.log)"; head -14 $OUT/kernel_stats.csv | cut -c1-80,112-160
done
