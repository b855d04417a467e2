#!/bin/bash
# The test part of scripts/gpu_round.sh alone: every GPU test file, file by file, into gpurun_out/<tag>/pytest.log
TAG=${1:-tests}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/pytest.log
for f in tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_gradient_pins_gpu.py \
         tests/test_sharding_gpu.py tests/test_frontend_gpu.py tests/test_face_light_gpu.py tests/test_texture_io.py tests/test_optimizer.py \
         tests/test_examples_gpu.py tests/test_bench_contract.py tests/test_multi_rank_gpu.py tests/test_rccl_gpu.py; do
  echo "=== $f" >> $OUT/pytest.log
  timeout ${TEST_TIMEOUT:-600} python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -${TAIL:-200} >> $OUT/pytest.log
done
echo "=== tests/test_abi.py (no GPU marker: the ABI / symbol checks)" >> $OUT/pytest.log
timeout 300 python -m pytest tests/test_abi.py -q --tb=short -p no:cacheprovider 2>&1 | tail -5 >> $OUT/pytest.log
cp gpurun_out/parity_errors.jsonl gpurun_out/*ranks_one_gpu*.log $OUT/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
grep -E "===|passed|failed|error" $OUT/pytest.log | head -40
tail -2 $OUT/smoke.log
