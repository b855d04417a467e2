#!/bin/bash
OUT=gpurun_out/${TAG:-t1}; mkdir -p $OUT; rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/pytest.log
for f in ${FILES:-tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_abi.py}; do
  echo "=== $f" >> $OUT/pytest.log
  timeout 900 python -m pytest $f -q --tb=short -p no:cacheprovider 2>&1 | tail -40 >> $OUT/pytest.log
done
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
grep -E "===|passed|failed|rror" $OUT/pytest.log | head -30
