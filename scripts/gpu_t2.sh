#!/bin/bash
OUT=gpurun_out/${TAG:-t2}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/pytest.log
for f in tests/test_multi_rank_gpu.py tests/test_bench_contract.py; do
  echo "=== $f" >> $OUT/pytest.log
  timeout 1500 python -m pytest $f -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -40 >> $OUT/pytest.log
done
grep -E "===|passed|failed|rror" $OUT/pytest.log | head -30; tail -25 $OUT/pytest.log
cp gpurun_out/*ranks_one_gpu*.log $OUT/ 2>/dev/null
GPUS="1 2" ONE_GPU=1 STEPS=5 WARMUP=2 bash scripts/scale_sweep.sh $OUT/scale 2>&1 | tail -12
