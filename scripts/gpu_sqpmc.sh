#!/bin/bash
# SQ counters of one kernel (KERNEL=substring, default k_face_raster; several passes), default build or NR_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fwdpmc; mkdir -p $OUT
[ -n "$AVAIL" ] && rocprofv3 -L > $OUT/avail.txt 2>&1
n=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES"; do
  n=$((n+1))
  B=${B:-64} ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$n -- python scripts/stage_times.py > $OUT/p$n.log 2>&1
  python scripts/rocpd_pmc.py $OUT/p${n}_results.db ${KERNEL:-k_face_raster} 2>&1 | cut -c1-30,60-200
done
rm -f $OUT/*_results.db
