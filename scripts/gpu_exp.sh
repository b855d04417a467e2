#!/bin/bash
# Development GPU session: selected tests with full logs, one fuzz scene under the microscope, stage timings of variant
# builds (NR_HIP_LIB), the per-phase profile of k_bpm_fast, the other BASELINE configurations.
TAG=${1:-exp}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in ${TESTS:-tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_multi_rank_gpu.py tests/test_bench_contract.py}; do
  n=$(basename $f .py)
  timeout 600 python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider > $OUT/$n.log 2>&1
  echo "=== $f: $(tail -1 $OUT/$n.log)"
done
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 120 python scripts/debug_fuzz_case.py 3 6 > $OUT/fuzz_3_6.log 2>&1
for v in "" ${VARIANTS}; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  TAG=base$v ITERS=20 timeout 200 python scripts/stage_times.py 2>/dev/null | tail -1 >> $OUT/variants.log
done
for fl in ${STAGE_FLAGS}; do
  unset NR_HIP_LIB
  NR_STAGE_FLAGS=$fl TAG=flags$fl ITERS=20 timeout 200 python scripts/stage_times.py 2>/dev/null | tail -1 >> $OUT/variants.log
done
for v in ${VARIANT_TESTS}; do  # parity of a variant build: the forward / K6 suites against the oracle
  export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so
  timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/variant_$v.log 2>&1
  echo "=== variant $v: $(tail -1 $OUT/variant_$v.log)"
done
export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_phases.so
timeout 200 python scripts/k6_phases.py > $OUT/phases.log 2>&1
unset NR_HIP_LIB
if [ -n "$BENCH" ]; then timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['stages_us'].items()})
    print(json.dumps(d['grad_check'])[:700]); print(d['extra_rows'], d['renderer_end_to_end'])
except Exception as e:
    print('bench parse failed', e); print(open('$OUT/bench.err').read()[-2000:])
PY
fi
if [ -n "$CONFIGS" ]; then timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; fi
cat $OUT/variants.log; cat $OUT/phases.log | tail -12; cat $OUT/fuzz_3_6.log | tail -8
grep -E "^FAILED|^ERROR" $OUT/*.log | cut -c1-200 | head -40
[ -n "$CONFIGS" ] && (cat $OUT/configs.jsonl | cut -c1-300; tail -3 $OUT/configs.err)
