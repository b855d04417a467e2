#!/bin/bash
# forward kernels under the microscope: kernel trace + SQ counters of the forward kernels, default build and VARIANTS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  OUT=gpurun_out/fwdprof_${v:-base}
  mkdir -p $OUT
  for b in ${BATCHES:-64}; do
    B=$b ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st$b -- python scripts/stage_times.py > $OUT/run$b.log 2>&1
    python scripts/rocpd_stats.py $OUT/st${b}_results.db $OUT/kernel_stats_$b.csv > /dev/null
    echo "== ${v:-base} B=$b"; grep -E "k_face_raster|k_large_raster|k_resolve|k_shade" $OUT/kernel_stats_$b.csv | cut -c1-50,112-170
  done
  B=64 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU -d $OUT -o sq -- python scripts/stage_times.py > $OUT/sq.log 2>&1
  python scripts/rocpd_pmc.py $OUT/sq_results.db k_face_raster | cut -c1-30,60-200
  B=64 ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT -o sq2 -- python scripts/stage_times.py > $OUT/sq2.log 2>&1
  python scripts/rocpd_pmc.py $OUT/sq2_results.db k_face_raster | cut -c1-30,60-200
  rm -f $OUT/*_results.db
done
