#!/bin/bash
# round 5: where k_bpm_px's time goes -- switch-off builds (wrong results by construction), kernel trace per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/px5; mkdir -p $OUT
for v in v0 v1 v2 v3 v4 v5; do
  export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so
  for B in 64 8; do
    B=$B ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st_$v -- python scripts/stage_times.py > $OUT/st_$v.log 2>&1
    python scripts/rocpd_stats.py $OUT/st_${v}_results.db $OUT/ks_${v}_$B.csv > /dev/null 2>&1
    echo "$v B=$B $(grep k_bpm_px $OUT/ks_${v}_$B.csv | cut -d, -f 2- | tr -d '\"' | awk -F, '{print $(NF-4), $(NF-3)}')"
    rm -f $OUT/*_results.db
  done
done
