#!/bin/bash
# round 5, session 2: where k_bpm_px spends its time -- kernel trace + SQ counters on the headline K6 stage
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/px2; mkdir -p $OUT
for v in "" pxnp; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  tag=${v:-product}
  ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st_$tag -- python scripts/stage_times.py > $OUT/st_$tag.log 2>&1
  python scripts/rocpd_stats.py $OUT/st_${tag}_results.db $OUT/kernel_stats_$tag.csv > /dev/null 2>&1
  echo "== $tag"; head -12 $OUT/kernel_stats_$tag.csv | cut -c1-60,100-170
  n=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
             "SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES"; do
    n=$((n+1))
    ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p${n}_$tag -- python scripts/stage_times.py > $OUT/p${n}_$tag.log 2>&1
    python scripts/rocpd_pmc.py $OUT/p${n}_${tag}_results.db k_bpm_px 2>&1 | cut -c1-30,60-200 | tee -a $OUT/pmc_$tag.txt
  done
done
rm -f $OUT/*_results.db
