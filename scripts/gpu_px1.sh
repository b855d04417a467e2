#!/bin/bash
# round 5, session 1: the lane-parallel band kernel (k_bpm_px) -- parity tests, then error levels and stage times against the
# legacy kernel and the variant builds
OUT=gpurun_out/px1
mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in tests/test_hip_parity.py tests/test_fuzz_gpu.py; do
  echo "=== $f" >> $OUT/pytest.log
  timeout 600 python -m pytest $f -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -40 >> $OUT/pytest.log
done
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
VARIANTS="pxnp pxns" SCENES="H C4" K6_FLAGS="0 128 2" ITERS=30 timeout 600 python scripts/k6_numerics.py > $OUT/numerics.jsonl 2> $OUT/numerics.err
VARIANTS="pxnp" SHAPES="8x256 16x256 64x256 64x512 16x512" timeout 600 python scripts/k6_variants.py > $OUT/variants.jsonl 2> $OUT/variants.err
grep -E "===|passed|failed|error|Error" $OUT/pytest.log | head; tail -5 $OUT/pytest.log
cut -c1-330 $OUT/numerics.jsonl; tail -3 $OUT/numerics.err
cat $OUT/variants.jsonl; tail -3 $OUT/variants.err
