#!/bin/bash
# SQ counters of some kernels on one configuration of scripts/bench_configs.py (CONFIG=C4; KERNELS="k_face_raster k_large_raster")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-pmc_cfg}; mkdir -p $OUT
CONFIG=${CONFIG:-C4}
ONLY=$CONFIG timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o stats_$CONFIG -- python scripts/bench_configs.py > $OUT/stats_$CONFIG.log 2>&1
python scripts/rocpd_stats.py $OUT/stats_${CONFIG}_results.db $OUT/kernel_stats_$CONFIG.csv > /dev/null 2>&1
head -12 $OUT/kernel_stats_$CONFIG.csv | cut -c1-70,100-170
n=0
: > $OUT/pmc_$CONFIG.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  n=$((n+1))
  ONLY=$CONFIG timeout 600 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$n -- python scripts/bench_configs.py > $OUT/p$n.log 2>&1
  for k in ${KERNELS:-k_face_raster k_large_raster}; do
    python scripts/rocpd_pmc.py $OUT/p${n}_results.db $k 2>&1 | cut -c1-30,60-200 >> $OUT/pmc_$CONFIG.txt
  done
done
cat $OUT/pmc_$CONFIG.txt
rm -f $OUT/*_results.db
