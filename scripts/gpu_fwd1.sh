#!/bin/bash
# round 5: the forward with the constants of undrawn pixels stored inside the raster kernel's launch (A/B by flag), tests, trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-fwd1}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
SH="1x256 8x256 16x256 32x256 64x256 64x512 16x512"
for fl in 48 131120 48 131120; do  # (131072: NR_FLAG_FILL_IN_RESOLVE)
  echo "fwd_flags=$fl (48: epochs + sparse weights; +131072: NR_FLAG_NO_RASTER_FILL)"
  FWD_FLAGS=$fl SHAPES="$SH" ITERS=50 timeout 300 python scripts/fwd_variants.py 2>> $OUT/err.log | tee -a $OUT/fwd_$fl.jsonl
done
for fl in 48 131120; do
  echo "C4 fwd_flags=$fl"; SCENE=C4 FWD_FLAGS=$fl SHAPES="64x256" ITERS=30 timeout 300 python scripts/fwd_variants.py 2>> $OUT/err.log | tee -a $OUT/fwd_c4_$fl.jsonl
done
for f in tests/test_hip_parity.py tests/test_frontend_gpu.py tests/test_fuzz_gpu.py; do
  timeout 900 python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4
done
ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st -- python scripts/stage_times.py > $OUT/st.log 2>&1
python scripts/rocpd_stats.py $OUT/st_results.db $OUT/kernel_stats.csv > /dev/null 2>&1; rm -f $OUT/*_results.db
grep -E "k_face_raster|k_resolve|k_large" $OUT/kernel_stats.csv | cut -c1-50,100-180
tail -1 $OUT/st.log
