#!/bin/bash
# A/B session: parity tests with the default build, then stage times (headline scene) and selected BASELINE configurations for
# the default build and every variant library in VARIANTS (neural_renderer_amd/libnr_hip_<v>.so, picked up through NR_HIP_LIB).
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in ${TESTS:-tests/test_hip_parity.py tests/test_fuzz_gpu.py}; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider > $OUT/$n.log 2>&1
  echo "=== $f: $(tail -1 $OUT/$n.log)"
done
for v in "" ${VARIANTS}; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  for b in ${BATCHES:-64}; do
    B=$b TAG="lib=${v:-default} B=$b" ITERS=20 timeout 200 python scripts/stage_times.py 2>/dev/null | tail -1 | tee -a $OUT/stages.log
  done
  if [ -n "$ONLY" ]; then
    echo "lib=${v:-default}" >> $OUT/configs.log
    timeout 900 python scripts/bench_configs.py 2>$OUT/configs.err | cut -c1-400 | tee -a $OUT/configs.log
  fi
done
for v in ${VARIANT_TESTS}; do  # parity of a variant build: the forward / K6 suites against the oracle
  export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so
  timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/variant_$v.log 2>&1
  echo "=== variant $v: $(tail -1 $OUT/variant_$v.log)"
done
unset NR_HIP_LIB
grep -E "^FAILED|^ERROR" $OUT/*.log | cut -c1-200 | head -40
